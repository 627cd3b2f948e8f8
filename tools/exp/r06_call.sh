R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06dh
rm -rf $O; mkdir -p $O
cd $R
( timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_counts.py -m gpu -x -q -k "bwd or backward or grad or field or overflow or book" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
OFF="--no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 --no-render-block --no-config4 --no-config5 --no-train-app"
cp $R/perf_amd/libperf_hip.so /tmp/keep.so
for v in head pad head pad head pad; do
  cp $R/tools/exp/_variants/lib_$v.so $R/perf_amd/libperf_hip.so
  rm -rf $O/kt_$v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o kt -- python $R/bench.py --steps 40 --warmup 5 $OFF > $O/kt_$v.log 2>&1
  echo $v $(grep "hashgrid_bwd_kernel<true>" $O/kt_$v/kt_kernel_stats.csv | sed 's/(.*)"//' | cut -d, -f2,4,6 ) $(grep "tile_codes4" $O/kt_$v/kt_kernel_stats.csv | sed 's/(.*)"//' | cut -d, -f4 )
done
cp /tmp/keep.so $R/perf_amd/libperf_hip.so
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
