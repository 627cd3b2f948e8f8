R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06bz
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do
( timeout 900 python bench.py --no-config5 --no-render-block --no-config4 ) > $O/bench_$i.log 2> $O/bench_$i.err
python - <<PY
import json
t=open('$O/bench_$i.log').read()
d=json.loads([l for l in t.splitlines() if l.startswith('{')][-1])
print(d['ms_per_step'], d['faithful'], d['summary'].get('episode_train_seconds'), d['summary'].get('episode_psnr_db'), d['train_app']['ms_per_step'])
PY
done
