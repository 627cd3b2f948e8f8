cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ad
mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config4.py tests/test_gpu_counts.py -m gpu -q -x > $O/t1.log 2>&1; tail -3 $O/t1.log; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o c4 -- python $R/tools/render_dense.py --poses 300 --batch 524288 > $O/c4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ep -o ep -- python $R/tools/train_episode.py --geo 1000 --app 500 > $O/kt_ep.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
grep -h "composite_fwd\|visibility_count\|train_head\|compact_prefix2" $O/kt_c4/c4_kernel_stats.csv | cut -c1-50,150-330
echo; grep -h "composite_fwd\|visibility_count\|train_head\|compact_prefix2" $O/kt_ep/ep_kernel_stats.csv | cut -c1-50,150-330
