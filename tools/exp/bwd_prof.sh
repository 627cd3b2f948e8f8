cd /tmp && export TMPDIR=/tmp
R=/root/repo
for V in lists codes; do
  if [ $V = codes ]; then export PERF_BWD_NO_LISTS=1; else unset PERF_BWD_NO_LISTS; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bwdprof_$V -- python $R/tools/exp/bwd_ab.py > /dev/null 2>&1
  find $R/gpurun_out/bwdprof_$V -name "*.db" -delete; find $R/gpurun_out/bwdprof_$V -name "*kernel_trace.csv" -delete
  echo $V; python - <<P
import csv,glob
f=glob.glob('$R/gpurun_out/bwdprof_$V/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'perf::' in r['Name']: print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
P
done
