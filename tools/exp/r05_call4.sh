R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python tools/exp/dp_margin_sweep.py 1 2 3 4 6 2>&1 | cut -c1-1500
timeout 300 python -m pytest tests/test_gpu_config5.py tests/test_gpu_ops.py -m gpu -q -k "config5 or mlp" 2>&1 | tail -5
