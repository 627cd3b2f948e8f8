cd /root/repo
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror|^FAILED" | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_final.json
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_final.json').read())
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['cpu_baseline']['value'], d['sustained']['value'], d['with_feature_reuse']['value'], d['psnr']['curve'])
P
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_v3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_v3 -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-reuse-line --sustain-seconds 0 > /dev/null 2>&1
find /root/repo/gpurun_out/prof_v3 -name "*.db" -delete; find /root/repo/gpurun_out/prof_v3 -name "*kernel_trace.csv" -delete
ls /root/repo/gpurun_out/prof_v3/*/
