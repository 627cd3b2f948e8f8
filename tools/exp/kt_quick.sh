# kernel-trace statistics of the default training step (20 steps) for a build / a set of environment switches:
#   bash tools/exp/kt_quick.sh <tag> [VAR=value ...]   ->  gpurun_out/kt_<tag>/..._kernel_stats.csv
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$tag -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 --no-render-block --no-config4 > $R/gpurun_out/kt_$tag.log 2>&1
find $R/gpurun_out/kt_$tag -name "*.db" -delete; find $R/gpurun_out/kt_$tag -name "*kernel_trace.csv" -delete; find $R/gpurun_out/kt_$tag -name "*agent_info.csv" -delete
