cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05af
mkdir -p $O
cd $R; timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config4.py tests/test_gpu_counts.py tests/test_gpu_scene.py tests/test_gpu_fullsize.py tests/test_gpu_config5.py -m gpu -q --maxfail=5 > $O/t1.log 2>&1; tail -3 $O/t1.log; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o c4 -- python $R/tools/render_dense.py --poses 300 --batch 524288 > $O/c4.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
grep -h "composite_fwd\|march_count_shared" $O/kt_c4/c4_kernel_stats.csv | cut -c1-50,150-330
cd $R; for i in 1 2 3; do timeout 300 python tools/render_dense.py --poses 600 --batch 524288 2>/dev/null | tail -1 | cut -c1-300; done
