R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06ch
rm -rf $O; mkdir -p $O
cd $R
cp perf_amd/libperf_hip.so /tmp/lib_keep.so
for v in a d a d; do
cp tools/exp/_variants/lib_$v.so perf_amd/libperf_hip.so
timeout 600 python tools/exp/team_shape_ab.py 2>&1 | grep -v amdgpu.ids > $O/ab_$v.log
python -c "
import json; d=json.load(open('$O/ab_$v.log')); print('$v', {k.replace(' samples/ray, ','x').replace(' rays',''):(v['perf_visibility_count'],v['perf_composite_fwd']) for k,v in d.items()})"
done
cp /tmp/lib_keep.so perf_amd/libperf_hip.so
( timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config4.py tests/test_gpu_scene.py -m gpu -x -q -k "team or visib or compos or compact or config4 or frame or head or render" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
timeout 600 python tools/render_dense.py --poses 300 --batch 524288 > $O/rd_$i.log 2>&1
python - <<PY
import json
t=open('$O/rd_$i.log').read()
d=json.loads(t[t.rindex('\n{'):] if '\n{' in t else t[t.index('{'):])
print({k: d[k] for k in d if 'frames_per_s' in k or 'rgb_sum' in k})
PY
done
