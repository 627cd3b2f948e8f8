cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05ae
mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_c4 -o c -- python $R/tools/render_dense.py --poses 24 --batch 524288 > $O/c4.log 2>&1
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
ls -la $O/pmc_c4/*/ 2>/dev/null | head; du -sh $O
