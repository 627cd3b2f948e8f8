R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06ai
rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "deep_grid or beyond_32" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/config5.py --pano-log2 28 30 --layout tcnn > $O/c5_tcnn.log 2>&1; grep -E "seconds_per_panorama|ms_per_launch|\"frac\"" $O/c5_tcnn.log
timeout 600 python tools/config5.py --pano-log2 28 30 --layout line_local > $O/c5_ll.log 2>&1; grep -E "seconds_per_panorama|ms_per_launch|\"frac\"" $O/c5_ll.log
cd /tmp && export TMPDIR=/tmp
for LAY in tcnn; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/c5_${LAY}_$C -o c -- python $R/tools/config5.py --pano-log2 28 30 --pano-batches 8 --layout $LAY --tile 128 128 > $O/c5_${LAY}_$C.log 2>&1
  done
done
cd $R
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
