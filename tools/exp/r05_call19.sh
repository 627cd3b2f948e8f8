R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc -o c -- python $R/tools/render_dense.py --poses 40 --batch 524288 > $O/log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('$O/pmc/**/*counter_collection.csv', recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'march_count_kernel<true>' in n and int(r['Grid_Size'])>10000000: agg['count_head_frame'][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'launches', len(next(iter(v.values()))))
PY
find $O -name "*.db" -delete
