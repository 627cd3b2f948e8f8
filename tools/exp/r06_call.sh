R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06bu
rm -rf $O; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/exp/gather_probe2.hip -o /tmp/gather_probe2 2>/dev/null
/tmp/gather_probe2 18 20 22 24 | tee $O/probe2.log
