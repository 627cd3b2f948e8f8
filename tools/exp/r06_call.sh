R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06am
rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config5.py -m gpu -x -q -k "deep_grid or beyond_32 or config5" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/config5.py --pano-log2 28 --layout line_local > $O/kt.log 2>&1
grep big $O/kt/kt_kernel_stats.csv | sed 's/(perf::GridParams.*)",/ /' | cut -c1-160
cd $R
timeout 600 python tools/config5.py --pano-log2 28 30 --layout line_local > $O/c5_ll.log 2>&1; grep -E "seconds_per_panorama|ms_per_launch|\"frac\"" $O/c5_ll.log
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
