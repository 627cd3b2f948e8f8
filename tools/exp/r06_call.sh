# Round 6 GPU call (rewritten per call; the log of calls is profiles/r06_gpurun_calls.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06j
rm -rf $O; mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log > $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $O/c5_line_local_$C -o c -- python $R/tools/config5.py --pano-log2 28 30 --pano-batches 8 --layout line_local > $O/c5_line_local_$C.log 2>&1
done
for SET in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_$N -o c -- python $R/tools/config5.py --pano-log2 28 --pano-batches 8 --layout line_local > $O/pmc_$N.log 2>&1
done
cd $R
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; tail -6 $O/pytest.log; tail -c 1500 $O/bench_line.json
