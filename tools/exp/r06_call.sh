R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06bm
rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "deep_grid or overlapping" ) > $O/pytest_ops.log 2>&1; tail -3 $O/pytest_ops.log
for cfg in "0 0" "2 4" "2 8" "4 2" "4 8" "8 2" "8 4" "4 1" "0 0"; do
set -- $cfg
if [ $1 = 0 ]; then unset PERF_EXP_STEPS PERF_EXP_TURN; else export PERF_EXP_STEPS=$1 PERF_EXP_TURN=$2; fi
timeout 600 python tools/config5.py --pano-log2 28 --layout line_overlap 2>&1 | grep -v amdgpu.ids > $O/c5.log
python - <<PY
import json,re
t=open('$O/c5.log').read()
m=re.findall(r'"(seconds_per_panorama|ms_per_launch|frac)": ([0-9.e+]+)', t)
print('steps,turn=$cfg', m[:6])
PY
done
