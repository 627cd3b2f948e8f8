R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06az
rm -rf $O; mkdir -p $O
cd $R
OFF="--no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 --no-render-block --no-config4 --no-config5 --no-train-app"
for i in 1 2 3; do
  timeout 600 python bench.py --steps 30 --warmup 5 $OFF > $O/bench_$i.log 2>&1
  python - <<PY
import json
t=open('$O/bench_$i.log').read().strip().splitlines()[-1]
d=json.loads(t)
k=d['kernels']
print('run $i', round(d['ms_per_step'],4), {n:k[n]['ms_per_launch'] for n in ('perf_hashgrid_fwd','perf_hashgrid_bwd')})
PY
done
( timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
