#!/bin/bash
mkdir -p gpurun_out/bwd
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_counts.py tests/test_gpu_dist.py -q -x -m gpu 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr > gpurun_out/bwd/bench_runs.json 2> gpurun_out/bwd/bench_runs.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bwd/bench_runs.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d.get('strict_two_evaluations'), d.get('sustained'))
for k in d.get('kernels', []): print(k)
PY
