cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05aa
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o c4 -- python $R/tools/render_dense.py --poses 300 --batch 524288 > $O/c4.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R; timeout 1200 python -m pytest tests/test_gpu_psnr.py -m gpu -q -s > $O/psnr.log 2>&1; tail -3 $O/psnr.log; grep -c "HIP - oracle" $O/psnr.log
