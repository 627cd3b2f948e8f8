R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c
rm -rf $O; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_config5.py tests/test_gpu_psnr.py -m gpu -q -k "config5 or data_parallel" ) > $O/pytest.log 2>&1
tail -30 $O/pytest.log | cut -c1-400
