cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05w
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ep -o ep -- python $R/tools/train_episode.py --geo 1000 --app 500 > $O/kt_ep.log 2>&1
tail -2 $O/kt_ep.log | cut -c1-600
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for i in 1 2 3; do timeout 300 python $R/tools/train_episode.py --geo 3000 --app 1500 2>/dev/null | tail -1 | cut -c1-400; done
