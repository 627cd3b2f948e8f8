R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05k
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "march or sampling or lattice or head or counts or graph or renderer or scene or fullsize" > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
MARCH_PROBE_REPS=20 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o m -- python $R/tools/exp/march_probe.py > $O/probe.log 2>&1
grep -v amdgpu $O/probe.log | grep "jittered\|lattice_runs" | cut -c1-260
python - <<PY
import csv,glob
f=glob.glob('$O/kt/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'march' in r['Kernel_Name'] or 'lattice_runs' in r['Kernel_Name']]
# the probe runs its configurations in a fixed order: print the mean duration of consecutive groups of equal kernels
import itertools
out=[]
for name, grp in itertools.groupby(rows, key=lambda r: r['Kernel_Name'][:50]):
    g=list(grp); d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in g]
    out.append((name, len(d), round(sum(d)/len(d),1)))
for o in out:
    if o[1] >= 15: print(o)
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
