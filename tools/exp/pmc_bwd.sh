cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d gpurun_out/pmc_a -o a --output-format csv -- python tools/exp/bwd_block_times.py > gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_b -o b --output-format csv -- python tools/exp/bwd_block_times.py > gpurun_out/pmc_b.log 2>&1
ls gpurun_out/pmc_a gpurun_out/pmc_b; tail -3 gpurun_out/pmc_b.log | cut -c1-300
