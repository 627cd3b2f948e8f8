# Round-6 profiling passes on the GPU box (one gpurun call): kernel-trace statistics of the default bench command (training
# step only: the render / config-4 / PSNR blocks are switched off), HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), MFMA
# and SQ counters of the training step.  Counters are collected WITHOUT --kernel-trace / --stats, each group in its own run.
# Raw CSVs land in gpurun_out/r06/; tools/exp/r06_fold.py folds them into profiles/r06_*.json (+ copies the CSVs).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
OFF="--no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 --no-render-block --no-config4 --no-config5 --no-train-app"
BENCH="python $R/bench.py --steps 10 --warmup 3 $OFF"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 5 $OFF > $O/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -o c -- $BENCH --no-graph > $O/pmc_$C.log 2>&1
done
# the colour phase's step (train_one_step_app at bench scale): its own kernel trace and MFMA counters
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_app -o kt -- python $R/bench.py --steps 20 --warmup 5 --mode train_app $OFF > $O/kt_app.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_app -o c -- $BENCH --mode train_app --no-graph > $O/pmc_mfma_app.log 2>&1
# the reference-faithful episode (1000 + 500 iterations): per-kernel time per step
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ep -o ep -- python $R/tools/train_episode.py --geo 1000 --app 500 > $O/kt_ep.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o c -- $BENCH --no-graph > $O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o c4 -- python $R/tools/render_dense.py --poses 300 --batch 524288 > $O/kt_c4.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; find $O -name "*.csv" | head -30
