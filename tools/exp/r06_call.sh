R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06bw
rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config4.py tests/test_gpu_counts.py tests/test_gpu_scene.py -m gpu -x -q -k "team or visib or compos or compact or config4 or frame or two_phase or render or distloss or accumulate" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
timeout 600 python tools/render_dense.py --poses 300 --batch 524288 > $O/rd_$i.log 2>&1
python - <<PY
import json
t=open('$O/rd_$i.log').read()
d=json.loads(t[t.rindex('\n{'):] if '\n{' in t else t[t.index('{'):])
print({k: d[k] for k in d if 'frames_per_s' in k or 'checksum' in k or 'rgb_sum' in k})
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/render_dense.py --poses 100 --batch 524288 > $O/kt.log 2>&1
grep "visibility\|compact\|composite\|finish" $O/kt/kt_kernel_stats.csv | sed 's/(.*)",/ /' | cut -c1-150
cd $R
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
