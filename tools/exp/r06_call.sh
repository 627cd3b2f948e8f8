R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06ce
rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config4.py tests/test_gpu_counts.py tests/test_gpu_scene.py -m gpu -x -q -k "march or lattice or config4 or frame or two_phase or head or render" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/exp/march_shape_ab.py 2>&1 | grep -v amdgpu.ids > $O/ab.log; python -c "
import json; d=json.load(open('$O/ab.log')); print({k:(v['samples_per_ray'],v['perf_occ_march_count'],v['perf_occ_march_write_points']) for k,v in d.items()})"
for i in 1 2; do
timeout 600 python tools/render_dense.py --poses 300 --batch 524288 > $O/rd_$i.log 2>&1
python - <<PY
import json
t=open('$O/rd_$i.log').read()
d=json.loads(t[t.rindex('\n{'):] if '\n{' in t else t[t.index('{'):])
print({k: d[k] for k in d if 'frames_per_s' in k or 'rgb_sum' in k})
PY
done
