R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06aw
rm -rf $O; mkdir -p $O
cd $R
( timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "deep_grid or beyond_32" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do for T in 28 30; do
  timeout 600 python tools/config5.py --pano-log2 $T --layout line_local > $O/c5_${T}_$i.log 2>&1
  python - <<PY
import json
t=open('$O/c5_${T}_$i.log').read()
try:
    d=json.loads(t[t.index('{'):])['T$T']
    print('4x2x2 sectors: T$T', d['seconds_per_panorama'], d['roofline']['ms_per_launch'], d['roofline']['frac'])
except Exception as e: print('failed', e, t[-500:])
PY
done; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_TCP -o c -- python $R/tools/config5.py --pano-log2 28 --pano-batches 8 --layout line_local --tile 128 128 > $O/pmc_TCP.log 2>&1
python - <<PY
import csv
s={}
for r in csv.DictReader(open('$O/pmc_TCP/c_counter_collection.csv')):
    if 'hashgrid_fwd_big' in r['Kernel_Name']:
        s.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
n=len(s['TCP_TOTAL_CACHE_ACCESSES_sum'])//2
print({k: round(sum(v)/n/4194304,2) for k,v in s.items()}, 'per sample,', n, 'encodes')
PY
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
