R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06df
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OFF="--no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 --no-render-block --no-config4 --no-config5 --no-train-app"
cp $R/perf_amd/libperf_hip.so /tmp/keep.so
for v in new r6 new r6 new r6; do
  cp $R/tools/exp/_variants/lib_$v.so $R/perf_amd/libperf_hip.so
  rm -rf $O/kt_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o kt -- python $R/bench.py --steps 40 --warmup 5 $OFF > $O/kt_$v.log 2>&1
  echo $v $(grep "hashgrid_bwd_kernel<true>" $O/kt_$v/kt_kernel_stats.csv | sed 's/(.*)"//' | cut -d, -f2,4,6 )
done
cp /tmp/keep.so $R/perf_amd/libperf_hip.so
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
