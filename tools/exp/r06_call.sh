# Round 6 GPU call (rewritten per call; the log of calls is profiles/r06_gpurun_calls.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06m
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c4 -o c4 -- python $R/tools/render_dense.py --poses 300 --batch 524288 > $O/c4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ep -o ep -- python $R/tools/train_episode.py --geo 1000 --app 500 > $O/ep.log 2>&1
cd $R
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; head -30 $O/kt_c4/*kernel_stats.csv | cut -c1-160
