R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06ay
rm -rf $O; mkdir -p $O
cd $R
for K in 1 2 3 4 6; do
  timeout 600 python tools/render_dense.py --poses 300 --batch 524288 --head $K > $O/rd_$K.log 2>&1
  python - <<PY
import json
t=open('$O/rd_$K.log').read()
try:
    d=json.loads(t[t.rindex('\n{'):] if '\n{' in t else t[t.index('{'):])
    print('head $K', {k: d[k] for k in d if 'frames_per_s' in k or 'checksum' in k or k in ('ms_per_frame',)})
except Exception as e: print('head $K parse failed', e, t[-600:])
PY
done
