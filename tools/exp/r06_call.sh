R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06bk
rm -rf $O; mkdir -p $O
cd $R
( timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config5.py tests/test_gpu_dist.py -m gpu -x -q -k "deep_grid or overlapping or beyond_32 or config5 or 20_level or more_than_16" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for lay in tcnn line_local line_overlap; do
for T in 28 30; do
timeout 600 python tools/config5.py --pano-log2 $T --layout $lay 2>&1 | grep -v amdgpu.ids > $O/c5_${T}_$lay.log
python - <<PY
import json,re
t=open('$O/c5_${T}_$lay.log').read()
m=re.findall(r'"(seconds_per_panorama|ray_samples_per_s|ms_per_launch|frac)": ([0-9.e+]+)', t)
print('T$T $lay', m[:8])
PY
done
done
cd /tmp && export TMPDIR=/tmp
for LAY in line_overlap line_local; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$LAY -o kt -- python $R/tools/config5.py --pano-log2 28 --layout $LAY > $O/kt_$LAY.log 2>&1
grep big $O/kt_$LAY/kt_kernel_stats.csv | sed 's/(perf::GridParams.*)",/ /' | cut -c1-160
done
cd $R
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
