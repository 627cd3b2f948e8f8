R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python tools/exp/march_probe.py 2>&1 | grep -v amdgpu | cut -c1-200
timeout 600 python -m pytest tests -m gpu -q -x -k "march or sampling or lattice or head or hashgrid_bwd or geo_step or counts or graph" 2>&1 | tail -4
