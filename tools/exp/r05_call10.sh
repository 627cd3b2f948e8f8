R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05j
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MARCH_PROBE_REPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o m -- python $R/tools/exp/march_probe.py > $O/probe.log 2>&1
grep -v amdgpu $O/probe.log | grep "jittered" | cut -c1-260
python - <<PY
import csv,glob
f=glob.glob('$O/kt/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'march' in r['Name'] or 'scan' in r['Name']: print(r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['MinNs'])/1e3,1), round(float(r['MaxNs'])/1e3,1), r['Name'][:60])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
