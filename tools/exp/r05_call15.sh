R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05o
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o m -- python $R/tools/exp/bwd_live_sweep.py > $O/log 2>&1
python - <<PY
import csv,glob,itertools
f=glob.glob('$O/kt/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted([r for r in csv.DictReader(open(f)) if 'perf::' in r['Kernel_Name']], key=lambda r:int(r['Start_Timestamp']))
for name, grp in itertools.groupby(rows, key=lambda r: r['Kernel_Name'][:44]):
    pass
# consecutive triples (codes, owners, reduce) x 12 per live count
by={}
seq=[(r['Kernel_Name'][6:40], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
import collections
i=0; live=['0','1024','16384','65536','262144','1M']
per=len(seq)//6
for k,l in enumerate(live):
    chunk=seq[k*per:(k+1)*per]
    agg=collections.defaultdict(list)
    for n_,d_ in chunk: agg[n_].append(d_)
    print(l, {n_: round(sum(v[2:])/max(len(v)-2,1),1) for n_,v in agg.items()})
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
