#!/bin/bash
# the data-parallel step on a world of ONE rank over the real RCCL backend vs the plain single-process step (same box)
mkdir -p gpurun_out/dp1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr > gpurun_out/dp1/plain.json 2> gpurun_out/dp1/plain.err
PERF_DP_SINGLE_RANK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr > gpurun_out/dp1/rccl1.json 2> gpurun_out/dp1/rccl1.err
PERF_DP_SINGLE_RANK=1 PERF_DP_GRAPH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr > gpurun_out/dp1/rccl1_eager.json 2> gpurun_out/dp1/rccl1_eager.err
python - <<'PY'
import json
for f in ('plain','rccl1','rccl1_eager'):
    try:
        d=[json.loads(l) for l in open(f"gpurun_out/dp1/{f}.json") if l.startswith("{")][-1]
        print(f, d['ms_per_step'], d['value'], d.get('launch'), (d.get('strict_two_evaluations') or {}).get('ms_per_step'), (d.get('sustained') or {}).get('ms_per_step'))
    except Exception as e:
        print(f, 'FAILED', e); print(open(f'gpurun_out/dp1/{f}.err').read()[-1500:])
PY
