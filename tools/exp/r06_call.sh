# Round 6 GPU call (rewritten per call; the log of calls is profiles/r06_gpurun_calls.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06e
rm -rf $O; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config5.py -m gpu -x -q --durations=10 ) > $O/pytest_new.log 2>&1
echo "pytest rc=$?" >> $O/pytest_new.log
for LAY in tcnn line_local; do
  ( time timeout 600 python tools/config5.py --pano-log2 28 30 --layout $LAY ) > $O/c5_$LAY.log 2>&1
  cp gpurun_out/config5_pano.json $O/config5_pano_$LAY.json
done
cd /tmp && export TMPDIR=/tmp
for LAY in tcnn line_local; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/c5_${LAY}_$C -o c -- python $R/tools/config5.py --pano-log2 28 30 --pano-batches 8 --layout $LAY > $O/c5_${LAY}_$C.log 2>&1
  done
done
cd $R
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; tail -15 $O/pytest_new.log; grep -h "ray_samples_per_s\|\"frac\"\|ms_per_launch\|seconds_per_panorama" $O/c5_tcnn.log $O/c5_line_local.log
