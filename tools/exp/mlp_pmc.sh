# SQ counters of the MLP kernels alone (tools/exp/mlp_pmc_run.py), three passes of eight; folded by tools/exp/mlp_pmc_fold.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/mlp_pmc
rm -rf $O; mkdir -p $O
RUN="python $R/tools/exp/mlp_pmc_run.py 5"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/exp/mlp_pmc_run.py 20 > $O/kt.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/a -o c -- $RUN > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $O/b -o c -- $RUN > $O/b.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/c -o c -- $RUN > $O/c.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python $R/tools/exp/mlp_pmc_fold.py $O > $O/fold.json; cat $O/fold.json | head -80
