cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_SECTORS_sum TCC_TAG_STALL_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE" \
         "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  for V in 0 1; do
    if [ $V = 1 ]; then export PERF_FWD_PAIR=1; else unset PERF_FWD_PAIR; fi
    timeout 120 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmcfwd/v$V/p$i -- python $R/tools/exp/fwd_pair.py train128 > /dev/null 2>$R/gpurun_out/pmcfwd_err_$V_$i.txt
  done
done
find $R/gpurun_out/pmcfwd -name "*.db" -delete
echo V0; python $R/tools/exp/pmc_fold.py $R/gpurun_out/pmcfwd/v0
echo V1; python $R/tools/exp/pmc_fold.py $R/gpurun_out/pmcfwd/v1
