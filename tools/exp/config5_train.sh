#!/bin/bash
# config-5 training step with and without the per-tile bitmaps of the large levels (tools/config5.py --train-log2)
mkdir -p gpurun_out/bwd
for b in 1 0; do PERF_BWD_BITMAP=$b timeout 900 python tools/config5.py --log2 --train-log2 19 22 24 > /dev/null 2>&1; cp gpurun_out/config5.json gpurun_out/bwd/config5_train_bitmap$b.json; python - <<PY
import json
d=json.load(open('gpurun_out/bwd/config5_train_bitmap$b.json'))
for k,v in d.items():
    if k.startswith('train'): print('bitmap=$b', k, v['ms_per_step'], v['kernel_ms_per_step'].get('perf_hashgrid_bwd'), v['samples_per_s'])
PY
done
