R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06as
rm -rf $O; mkdir -p $O
cd $R
for M in 200 256 400 512 700 64; do
  for T in 28 30; do
  timeout 600 python tools/config5.py --pano-log2 $T --layout line_local --local-min-res $M > $O/c5_${M}_$T.log 2>&1
  python - <<PY
import json
t=open('$O/c5_${M}_$T.log').read()
try:
    d=json.loads(t[t.index('{'):])['T$T']
    print('aligned: T$T local_min_res $M', d['seconds_per_panorama'], d['roofline']['ms_per_launch'], d['roofline']['frac'])
except Exception as e: print('$M failed', e, t[-500:])
PY
  done
done
