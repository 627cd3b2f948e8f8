R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
PERF_DP_DIAG=1 timeout 600 python tools/exp/dp_margin_sweep.py 1 2>&1 | cut -c1-400 | head -150
bash tools/exp/r05_call5.sh
timeout 900 python -m pytest tests -m gpu -q -x -k "hashgrid or grid or encod or field or geo_step or app_step or large or deep or bitmap" 2>&1 | tail -8
