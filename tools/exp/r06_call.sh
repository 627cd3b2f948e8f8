R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06ao
rm -rf $O; mkdir -p $O
cd $R
echo "two workgroups per CU (shipped):"; timeout 300 python tools/exp/mlp_bwd_scratch_ab.py 2>&1 | tail -7
sed -i 's/#define PERF_MLP_BWD_SPILL_FREE 0/#define PERF_MLP_BWD_SPILL_FREE 1/' perf_amd/csrc/mlp_device.hpp
python -m perf_amd.build > $O/build.log 2>&1; tail -1 $O/build.log
echo "one workgroup per CU for the variants with scratch:"; timeout 300 python tools/exp/mlp_bwd_scratch_ab.py 2>&1 | tail -7
