cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ep_prof -- python $R/tools/train_episode.py --geo 400 --app 200 > $R/gpurun_out/ep_prof.log 2>&1
find $R/gpurun_out/ep_prof -name "*.db" -delete
find $R/gpurun_out/ep_prof -name "*kernel_trace.csv" -delete
ls $R/gpurun_out/ep_prof/*/
