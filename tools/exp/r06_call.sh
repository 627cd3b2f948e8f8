R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06ci
rm -rf $O; mkdir -p $O
cd $R
( timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_scene.py tests/test_gpu_counts.py -m gpu -x -q -k "head or step or scene or train or graph or geo or app" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
( timeout 900 python bench.py --no-config5 --no-config4 ) > $O/bench_$i.log 2> $O/bench_$i.err
python - <<PY
import json
t=open('$O/bench_$i.log').read()
d=json.loads([l for l in t.splitlines() if l.startswith('{')][-1])
k=d['kernels']
print(d['ms_per_step'], d['faithful']['geo_ms_per_step'], d['faithful']['app_ms_per_step'], d['summary'].get('episode_psnr_db'), d['train_app']['ms_per_step'], {n:k[n]['ms_per_launch'] for n in ('perf_train_head_geo','perf_visibility_count')}, d['train_app']['kernels'].get('perf_train_head_app',{}).get('ms_per_launch'), d['render']['kernel_ms_per_panorama']['perf_visibility_count'], d['summary']['render_ray_samples_per_s'])
PY
done
