R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06cy
rm -rf $O; mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for s in room doorway; do
( timeout 900 python tools/soak_episodes.py --episodes 25 --scene $s --out $O/soak_$s.json ) > $O/soak_$s.log 2>&1
python -c "
import json; d=json.load(open('$O/soak_$s.json')); print('$s', d['params_sha256_16'], d['psnr_min_max'], d['seconds_min_max'], d['skipped_for_overflow_total'])"
done
