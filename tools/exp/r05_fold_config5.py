#!/usr/bin/env python
"""Fold the config-5 PMC passes (profiles/r05_gpurun_calls.md, call 1: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs, of
`python tools/config5.py --pano-log2 28 30 --pano-batches 8`) into profiles/r05_config5_pmc.json -- HBM bytes per launch of the
L = 20 encode kernel at each table size -- and copy the raw CSVs next to it (profiles/r05_raw/).  bench.py's `config5` block
reads the fold for `roofline.traffic` / `moved_frac`.

  python tools/exp/r05_fold_config5.py [gpurun_out/r05a]"""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r05a')
DST = os.path.join(ROOT, 'profiles')


def rows(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and 'perf::hashgrid_fwd' in r['Kernel_Name']:
            out.append((int(r['Dispatch_Id']), float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    return [(v, ns) for _, v, ns in sorted(out)]


tables = {}
f = rows(os.path.join(SRC, 'c5_FETCH_SIZE', 'c_counter_collection.csv'), 'FETCH_SIZE')
w = rows(os.path.join(SRC, 'c5_WRITE_SIZE', 'c_counter_collection.csv'), 'WRITE_SIZE')
sizes = [28, 30]
per = len(f) // len(sizes)
assert len(f) == len(w) == per * len(sizes)
for k, T in enumerate(sizes):
    ff, ww = f[k * per:(k + 1) * per], w[k * per:(k + 1) * per]
    fetch_kib = sum(v for v, _ in ff) / per
    write_kib = sum(v for v, _ in ww) / per
    ns = sum(t for _, t in ff) / per
    tables[f'T{T}'] = {'launches': per, 'samples_per_launch': 4 * 4096 * 256, 'fetch_size_kib_raw': round(fetch_kib, 1), 'write_size_kib': round(write_kib, 1),
                       'hbm_bytes_per_launch': int((2 * fetch_kib + write_kib) * 1024),
                       'hbm_bytes_per_launch_if_requests_are_64B': int((fetch_kib + write_kib) * 1024),
                       'algorithmic_bytes_per_launch': 640 * 4 * 4096 * 256,
                       'mean_ns_under_the_counter_pass': round(ns),
                       'per_launch_fetch_kib_raw': [round(v) for v, _ in ff]}
out = {'source': 'profiles/r05_config5_pmc.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python tools/config5.py --pano-log2 28 30 '
                 '--pano-batches 8` (8 batches of 4 panorama rows spread from pole to pole, 2 encodes each); FETCH_SIZE doubled as '
                 'MI355X_MICROARCH.md prescribes (gfx950 tallies 128-byte requests at 64 B) -- should random sector requests be 64 bytes '
                 'wide the moved bytes are the `_if_requests_are_64B` figure; raw CSVs in profiles/r05_raw/',
       'kernel': 'perf::hashgrid_fwd_kernel<FP16> (generic L-level encode), L = 20, finest resolution 8192',
       'note': 'rows at the poles (first and last two launches of a table size) are ray-coherent and fetch a quarter of what equatorial rows do',
       'tables': tables}
json.dump(out, open(os.path.join(DST, 'r05_config5_pmc.json'), 'w'), indent=1)
os.makedirs(os.path.join(DST, 'r05_raw'), exist_ok=True)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    shutil.copy(os.path.join(SRC, f'c5_{c}', 'c_counter_collection.csv'), os.path.join(DST, 'r05_raw', f'config5_pmc_{c}.csv'))
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != 'per_launch_fetch_kib_raw'} for k, v in tables.items()}, indent=1))
