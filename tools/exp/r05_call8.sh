R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python tools/exp/march_probe.py 2>&1 | tail -8
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "large_levels or bitmap or run_merging or lagged" 2>&1 | tail -3
