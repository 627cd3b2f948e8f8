R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06cn
rm -rf $O; mkdir -p $O
cd $R
for s in room doorway pillars; do
( timeout 900 python tools/soak_episodes.py --episodes 25 --scene $s --out $O/soak_$s.json ) > $O/soak_$s.log 2>&1
python - <<PY
import json
d=json.load(open('$O/soak_$s.json'))
print('$s', {k: d[k] for k in d if k in ('psnr_min_max','seconds_min_max','skipped_for_overflow_total','skipped_for_truncation_total','params_sha256_16')}, [k for k in d if 'flagged' in k], [d[k] for k in d if 'flagged' in k])
PY
done
( PERF_BOOK_IN_REPAIR_LAUNCH=0 timeout 900 python tools/soak_episodes.py --episodes 25 --scene room --out $O/soak_room_own_launch.json ) > $O/soak_room_own.log 2>&1
python -c "
import json; d=json.load(open('$O/soak_room_own_launch.json')); print('room, bookkeeping as its own launch', d['params_sha256_16'], d['seconds_min_max'])"
