"""Per-step kernel time table from a rocprofv3 --kernel-trace --stats run of bench.py:
   python tools/kernel_stats.py gpurun_out/prof/x_kernel_stats.csv [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
calls = [int(r['Calls']) for r in rows if 'hashgrid_bwd_kernel' in r['Name'] or 'hashgrid_fwd_kernel' in r['Name']]
calls = min(calls) if calls else 1
tot = 0.0
for r in rows:
    tot += float(r['TotalDurationNs']) / calls / 1e3
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / calls / 1e3:8.1f} us/step  x{int(r['Calls']) / calls:5.2f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:80]}")
print(f'{tot:8.1f} us/step all kernels')
