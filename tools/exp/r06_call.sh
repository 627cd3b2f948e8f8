R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06ap
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python tools/exp/c5_batch_size.py 2>&1 | grep -v amdgpu.ids | tee $O/batch.log
