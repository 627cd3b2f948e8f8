R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06cm
rm -rf $O; mkdir -p $O
cd $R
( timeout 1800 python -m pytest tests/test_gpu_counts.py tests/test_gpu_scene.py tests/test_gpu_ops.py tests/test_gpu_runner_state.py -m gpu -x -q -k "book or overflow or repair or gate or step or scene or train or graph or adam or field_back or episode or runner" ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
