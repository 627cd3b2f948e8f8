#!/bin/bash
mkdir -p gpurun_out/bwd
timeout 600 python tools/exp/bwd_sort.py gpurun_out/bwd/bwd_bitmap.json > gpurun_out/bwd/bwd_bitmap.log 2>&1; echo "harness exit $?"
grep -v "amdgpu.ids\|    level" gpurun_out/bwd/bwd_bitmap.log
grep "rays bitmap=1 runs=1" -A16 gpurun_out/bwd/bwd_bitmap.log
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "hashgrid_bwd" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bs -- python $GRAFT_REPO_ROOT/tools/exp/bwd_sort.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/bwd/bitmap_kernel_stats.csv; head -6 $f | cut -c1-60,330-420
