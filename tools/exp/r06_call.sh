R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06ae
rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "deep_grid or beyond_32" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/config5.py --pano-log2 28 --layout line_local > $O/kt.log 2>&1
grep -E "big" $O/kt/*kernel_stats.csv | cut -c1-60,200-330
cd $R
timeout 600 python tools/config5.py --pano-log2 28 30 --layout line_local > $O/c5_line_local.log 2>&1
grep -E "seconds_per_panorama|ms_per_launch|\"frac\"|ray_samples_per_s" $O/c5_line_local.log
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
