# rocprofv3 kernel trace of a short reference-faithful episode (graph-replayed steps): per-kernel GPU time per step
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/ep_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ep_prof -o ep -- python $R/tools/train_episode.py --geo 1000 --app 500 > $R/gpurun_out/ep_prof.log 2>&1
find $R/gpurun_out/ep_prof -name "*.db" -delete
find $R/gpurun_out/ep_prof -name "*kernel_trace.csv" -delete
ls $R/gpurun_out/ep_prof/
