R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05t
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "lane_per_ray or fullsize or config4 or march or renderer or scene" > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o c4 -- python $R/tools/render_dense.py --poses 300 --batch 524288 > $O/log 2>&1
tail -2 $O/log | cut -c1-400
python - <<PY
import csv,glob
f=glob.glob('$O/kt/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:8]:
    print(r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['TotalDurationNs'])/1e6,2), r['Name'][:90])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
