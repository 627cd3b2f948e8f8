#!/usr/bin/env python
"""Fold round 6's config-5 counter passes into profiles/:
  r06_fetch_size_calibration.json  rocprofv3 --pmc FETCH_SIZE / TCC_EA0_RDREQ / TCC_{HIT,MISS,REQ} of tools/exp/gather_probe.hip (a KNOWN number of
                                   random 4-, 8- and 16-byte loads over 16 GiB): what one random gather tallies
  r06_config5_pmc.json             FETCH_SIZE / WRITE_SIZE (separate passes) of `python tools/config5.py --pano-log2 28 30 --pano-batches 8 --layout L`
                                   for both table layouts: HBM bytes per launch of hashgrid_fwd_big_kernel (bench.py's config5 block reads it)
  r06_config5_counters.json        SQ / TA / TCP / TCC counters of the line-local encode at T = 2^28, one lane per sample (r06f) and four lanes per
                                   sample (r06j)
  python tools/exp/r06_fold_config5.py      (reads gpurun_out/r06a, r06f, r06j, r06v, r06ai, r06at, r06bi, r06bl; copies the raw CSVs to profiles/r06_raw/)"""
import collections, csv, glob, json, os, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, 'gpurun_out')
DST = os.path.join(ROOT, 'profiles')
RAW = os.path.join(DST, 'r06_raw')
os.makedirs(RAW, exist_ok=True)
PER_LAUNCH = 4 * 4096 * 256


def keep(src, name):
    """Raw rows of the kernels the folds are about (the passes also record every torch / set-up kernel of the process)."""
    with open(src) as f, open(os.path.join(RAW, name), 'w') as g:
        for i, line in enumerate(f):
            if i == 0 or 'hashgrid_fwd_big' in line or 'probe<' in line:
                g.write(line)


# ---- calibration ---------------------------------------------------------------------------------------------------------------
cal = {'command': 'hipcc --offload-arch=gfx950 -O3 tools/exp/gather_probe.hip -o /tmp/gather_probe; rocprofv3 --pmc <counters> -- /tmp/gather_probe 34',
       'what': '16,777,216 lanes x K x rounds independent loads at uniformly random aligned addresses of a 16 GiB table (no cache holds it); '
               'variant <mode, K>: mode 0 dword, 1 dword nontemporal, 2 dword sc0 sc1, 3 dwordx2, 4 dwordx4', 'per_dispatch': {}}
lanes = 65536 * 256
rounds = {('0', '8'): 2, ('4', '8'): 2}
for tag, counters in (('FETCH', ['FETCH_SIZE']), ('RDREQ', ['TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum']), ('TCC', ['TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_REQ_sum'])):
    path = os.path.join(G, 'r06a', f'probe_{tag}', 'p_counter_collection.csv')
    keep(path, f'gather_probe_{tag}.csv')
    for r in csv.DictReader(open(path)):
        if 'probe<' not in r['Kernel_Name']:
            continue
        mode, k = r['Kernel_Name'].split('probe<')[1].split('>')[0].replace(' ', '').split(',')
        key = f'probe<{mode},{k}>'
        d = cal['per_dispatch'].setdefault(key, {'loads': lanes * int(k) * rounds.get((mode, k), 1)})
        d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for key, d in cal['per_dispatch'].items():
    for c in list(d):
        if isinstance(d[c], list):
            d[c] = round(sum(d[c]) / len(d[c]), 1)
    d['FETCH_SIZE_bytes_per_load'] = round(d['FETCH_SIZE'] * 1024 / d['loads'], 2)
    d['fabric_read_requests_per_load'] = round(d['TCC_EA0_RDREQ_sum'] / d['loads'], 4)
cal['reading'] = ('every random gather -- 4, 8 or 16 bytes wide, any cache policy -- is ONE fabric read request (TCC_EA0_RDREQ, none of them 32-byte), and FETCH_SIZE '
                  'tallies it at 64 bytes exactly like the requests of a streaming read, which MI355X_MICROARCH.md calibrates at 128 bytes moved per request '
                  '(FETCH_SIZE = half the bytes of a wide coalesced read).  A random gather therefore moves one 128-byte line through the fabric: the doubling '
                  'applies, and the 4.8e10 requests/s this probe reaches at every width (profiles/r02_gather_probe.json) are 6.2 TB/s -- the achievable HBM rate, '
                  'not a separate request-rate limit.  `moved` figures of the config-5 encode are 2 x FETCH_SIZE + WRITE_SIZE.')
json.dump(cal, open(os.path.join(DST, 'r06_fetch_size_calibration.json'), 'w'), indent=1)


# ---- FETCH / WRITE of the encode, both layouts -----------------------------------------------------------------------------------
def encodes(path, counters):
    """Per ENCODE (one perf_hashgrid_fwd call = the line-local launch + the gather launch of the other levels, or the gather launch alone
    with tcnn's layout): {counter: value summed over its launches, 'ns': summed duration}, in dispatch order."""
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] in counters and 'hashgrid_fwd_big' in r['Kernel_Name']:
            d = per[int(r['Dispatch_Id'])]
            d['name'] = r['Kernel_Name']; d[r['Counter_Name']] = float(r['Counter_Value']); d['ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    out = []
    for _, d in sorted(per.items()):
        follows_local = bool(out) and out[-1]['_open'] and 'big_gather' in d['name']
        if follows_local:
            g = out[-1]; g['_open'] = False
        else:
            g = {'ns': 0, '_open': 'big_local' in d['name']}
            out.append(g)
        g['ns'] += d['ns']
        for c in counters:
            g[c] = g.get(c, 0.0) + d.get(c, 0.0)
    return out


def rows(path, counter):
    return [(g[counter], g['ns']) for g in encodes(path, [counter])]


tables = {}
for layout, call in (('tcnn', 'r06ai'), ('line_local', 'r06bl'), ('line_overlap', 'r06bl')):
    f = rows(os.path.join(G, call, f'c5_{layout}_FETCH_SIZE', 'c_counter_collection.csv'), 'FETCH_SIZE')
    w = rows(os.path.join(G, call, f'c5_{layout}_WRITE_SIZE', 'c_counter_collection.csv'), 'WRITE_SIZE')
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        keep(os.path.join(G, call, f'c5_{layout}_{c}', 'c_counter_collection.csv'), f'config5_{layout}_{c}.csv')
    per = len(f) // 2
    assert len(f) == len(w) == 2 * per
    for k, T in enumerate((28, 30)):
        ff, ww = f[k * per:(k + 1) * per], w[k * per:(k + 1) * per]
        fk = sum(v for v, _ in ff) / per; wk = sum(v for v, _ in ww) / per; ns = sum(t for _, t in ff) / per
        moved = int((2 * fk + wk) * 1024)
        tables[f'T{T}' + ('' if layout == 'tcnn' else '_' + layout)] = {
            'layout': layout, 'launches': per, 'samples_per_launch': PER_LAUNCH, 'fetch_size_kib_raw': round(fk, 1), 'write_size_kib': round(wk, 1),
            'hbm_bytes_per_launch': moved, 'algorithmic_bytes_per_launch': 640 * PER_LAUNCH, 'moved_over_algorithmic': round(moved / (640 * PER_LAUNCH), 3),
            'mean_ns_under_the_counter_pass': round(ns), 'moved_TBps': round(moved / ns / 1e3, 3), 'moved_frac_of_8TBps': round(moved / ns / 8e3, 4),
            'algorithmic_frac_of_8TBps': round(640 * PER_LAUNCH / ns / 8e3, 4), 'per_launch_fetch_kib_raw': [round(v) for v, _ in ff]}
json.dump({'source': 'profiles/r06_config5_pmc.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python tools/config5.py --pano-log2 28 30 '
                     '--pano-batches 8 --layout <tcnn|line_local|line_overlap> --tile 128 128` (8 batches of 128 x 128 pixels spread from pole to pole, 2 encodes each); moved = 2 x FETCH_SIZE + '
                     'WRITE_SIZE (profiles/r06_fetch_size_calibration.json: a random gather is one 128-byte request tallied at 64); raw CSVs in profiles/r06_raw/',
           'kernel': 'perf::hashgrid_fwd_big_mixed_kernel<FP16, 4, OVL> (line_local / line_overlap: the 15 line-local levels and the 5 coarse ones in one launch) / '
                     'perf::hashgrid_fwd_big_gather_kernel<FP16, 1> (tcnn), summed per encode; L = 20, finest resolution 8192; line_local = 4x4x2-vertex lines in '
                     '32x64x256-vertex super-blocks, levels of resolution >= 64; line_overlap = the same lines with x runs that overlap by one vertex (8-byte requests), 128x32x128 super-blocks',
           'tables': tables}, open(os.path.join(DST, 'r06_config5_pmc.json'), 'w'), indent=1)

# ---- SQ / TA / TCP / TCC counters of the line-local encode at T = 2^28 ------------------------------------------------------------
cnt = {'command': 'rocprofv3 --pmc <set> -- python tools/config5.py --pano-log2 28 --pano-batches 8 --layout line_local (one pass per set), means per launch of '
                  'hashgrid_fwd_big_kernel (4,194,304 samples x 20 levels)', 'variants': {}}
for name, call in (('one lane per sample: four consecutive 16-byte x-run loads (64x64x128 super-blocks)', 'r06f'),
                   ('four lanes per sample: the four x-runs of a sample in one instruction (32x64x256 super-blocks), 4-row strip batches', 'r06j'),
                   ('four lanes per sample, 128 x 128-pixel tile batches', 'r06v'),
                   ('the same with long-lived waves: four steps of 64 samples per wave, coordinates requested one step ahead (shipped; line-local launch + '
                    'coarse-level gather launch summed; levels aligned to super-blocks)', 'r06at'),
                   ('overlapping x runs (layout line_overlap, 128x32x128 super-blocks): one 8-byte request per (y, z) corner pair, 56 registers; two launches', 'r06bi'),
                   ('overlapping x runs, line-local and coarse levels in ONE launch (shipped: hashgrid_fwd_big_mixed_kernel)', 'r06bl')):
    agg = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(G, call, 'pmc_*', ''))):
        fcsv = glob.glob(d + '*counter_collection.csv')
        if not fcsv:
            continue
        keep(fcsv[0], f'config5_counters_{call}_{os.path.basename(os.path.dirname(d))}.csv')
        names = sorted({r['Counter_Name'] for r in csv.DictReader(open(fcsv[0]))})
        for g in encodes(fcsv[0], names):
            for c in names:
                agg[c].append(g[c])
                agg['launch_ns'].append(g['ns'])
    m = {k: round(sum(v) / len(v)) for k, v in agg.items()}
    if 'SQ_WAVE_CYCLES' in m:
        m['wave_cycles_waiting_to_issue_frac'] = round(m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], 3)
        m['wave_cycles_waiting_for_data_frac'] = round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 3)
        m['wave_cycles_issuing_frac'] = round(m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'], 3)
    if 'TCP_TOTAL_CACHE_ACCESSES_sum' in m:
        m['l1_accesses_per_sample'] = round(m['TCP_TOTAL_CACHE_ACCESSES_sum'] / PER_LAUNCH, 1)
        m['l1_to_l2_reads_per_sample'] = round(m['TCP_TCC_READ_REQ_sum'] / PER_LAUNCH, 1)
    if 'TCC_MISS_sum' in m:
        m['l2_misses_per_sample'] = round(m['TCC_MISS_sum'] / PER_LAUNCH, 2)
        m['l2_hit_rate'] = round(m['TCC_HIT_sum'] / max(m['TCC_REQ_sum'], 1), 3)
    cnt['variants'][name] = m
json.dump(cnt, open(os.path.join(DST, 'r06_config5_counters.json'), 'w'), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != 'per_launch_fetch_kib_raw'} for k, v in tables.items()}, indent=1))
print(json.dumps(cnt['variants'], indent=1))
print(json.dumps({k: {kk: v[kk] for kk in ('loads', 'FETCH_SIZE_bytes_per_load', 'fabric_read_requests_per_load')} for k, v in cal['per_dispatch'].items()}, indent=1))
