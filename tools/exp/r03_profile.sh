# Round-3 profiling passes on the GPU box (one gpurun call): kernel-trace statistics of the default bench command, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes), MFMA and SQ counters of the training step, L1 counters of the forward encode
# before / after run de-duplication.  Counters are collected WITHOUT --kernel-trace / --stats, each group in its own run.
# Raw CSVs land in gpurun_out/r03/; tools/exp/r03_fold.py folds them into profiles/r03_*.json (+ copies the CSVs).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
BENCH="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 > $O/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -o c -- $BENCH --no-graph > $O/pmc_$C.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o c -- $BENCH --no-graph > $O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o c -- $BENCH --no-graph > $O/pmc_sq.log 2>&1
for V in static dedup; do
  if [ $V = static ]; then export PERF_FWD_V2=0; else unset PERF_FWD_V2; fi
  timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/l1_$V/a -o c -- python $R/tools/exp/fwd_one.py train > $O/l1_$V.log 2>&1
  timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/l1_$V/b -o c -- python $R/tools/exp/fwd_one.py train >> $O/l1_$V.log 2>&1
done
unset PERF_FWD_V2
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; find $O -name "*.csv" | head -30
