#!/bin/bash
mkdir -p gpurun_out/bwd
timeout 600 python tools/exp/bwd_sort_exp.py > gpurun_out/bwd/bwd_sort_exp.log 2>&1; echo "exp exit $?"
grep -v amdgpu.ids gpurun_out/bwd/bwd_sort_exp.log
cd /tmp && export TMPDIR=/tmp
PERF_BWD_SORT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bs -- python $GRAFT_REPO_ROOT/tools/exp/bwd_sort_exp.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/bwd/exp_kernel_stats.csv; head -12 $f | cut -c1-200
