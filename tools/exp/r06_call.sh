R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06bl
rm -rf $O; mkdir -p $O
cd $R
( time timeout 3000 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err; tail -c 600 $O/bench.log; tail -4 $O/bench.err
cd /tmp && export TMPDIR=/tmp
for LAY in line_overlap line_local; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/c5_${LAY}_$C -o c -- python $R/tools/config5.py --pano-log2 28 30 --pano-batches 8 --layout $LAY --tile 128 128 > $O/c5_${LAY}_$C.log 2>&1
  done
done
for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$N -o c -- python $R/tools/config5.py --pano-log2 28 --pano-batches 8 --layout line_overlap --tile 128 128 > $O/pmc_$N.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/config5.py --pano-log2 28 --layout line_overlap > $O/kt.log 2>&1
grep big $O/kt/kt_kernel_stats.csv | sed 's/(perf::GridParams.*)",/ /' | cut -c1-160
cd $R
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
du -sh $O
