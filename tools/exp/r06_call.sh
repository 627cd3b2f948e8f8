R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06x
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/exp/march_live_stats.py > $O/live.log 2>&1; tail -12 $O/live.log
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "scan_and_empty or field_backward" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
