#!/usr/bin/env python
"""Fold the raw rocprofv3 CSVs of tools/exp/r06_profile.sh (gpurun_out/r06/) into profiles/r06_*.json and copy the CSVs next
to them (profiles/r06_raw/), so that every number of the folds can be recomputed from committed data."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, 'gpurun_out', 'r06')
DST = os.path.join(ROOT, 'profiles')
RAW = os.path.join(DST, 'r06_raw')
KERNELS = ['lattice_runs_kernel', 'march_write_kernel', 'hashgrid_fwd_v2_kernel', 'hashgrid_fwd_kernel', 'hashgrid_bwd_kernel', 'tile_codes4_kernel', 'tile_codes_kernel', 'hashgrid_bwd_reduce_kernel',
           'mlp_fwd_kernel', 'mlp_bwd_kernel', 'mlp_reduce_kernel', 'adam4_kernel', 'adam_kernel', 'march_count_kernel', 'compact_prefix_kernel',
           'composite_distloss_fwd_kernel', 'composite_distloss_bwd_kernel', 'train_head_kernel', 'march_count_shared_kernel']
MLP_ENTRY = {'mlp_fwd_kernel': 'perf_mlp_fwd', 'mlp_bwd_kernel': 'perf_mlp_bwd'}
ENTRY = {'hashgrid_bwd_kernel': 'perf_hashgrid_bwd', 'tile_codes_kernel': 'perf_hashgrid_bwd', 'tile_codes4_kernel': 'perf_hashgrid_bwd', 'hashgrid_bwd_reduce_kernel': 'perf_hashgrid_bwd',
         'hashgrid_fwd_v2_kernel': 'perf_hashgrid_fwd', 'hashgrid_fwd_kernel': 'perf_hashgrid_fwd', 'mlp_bwd_kernel': 'perf_mlp_bwd',
         'mlp_reduce_kernel': 'perf_mlp_bwd', 'mlp_fwd_kernel': 'perf_mlp_fwd', 'adam_kernel': 'perf_adam_step_dev', 'adam4_kernel': 'perf_adam_step_dev'}


SAMPLES = 8192 * 128          # ray-samples per launch of the bench workload


def algorithmic_flop(k):
    """SURVEY.md 8(d)'s UNPADDED FLOP per sample of an MLP kernel (the MFMA counters count the padded work: 16 output rows, zero-padded
    inputs): forward geo 2 (32 x 64 + 64 x 1) = 4,224, app 2 (32 x 64 + 64 x 64 + 64 x 3) = 12,672; backward = 3 x forward (recomputed
    forward + dW + dX)."""
    if not k.startswith('mlp_') or '<' not in k:
        return None
    nh = int(k.split('<')[1].split(',')[1])
    fwd = 4224 if nh == 1 else 12672
    return fwd if k.startswith('mlp_fwd') else 3 * fwd


def kname(full):
    if 'perf::mlp_fwd_kernel<' in full or 'perf::mlp_bwd_kernel<' in full:        # keep the template arguments apart (density / colour net)
        base = 'mlp_fwd_kernel' if 'mlp_fwd_kernel' in full else 'mlp_bwd_kernel'
        args = full.split(base + '<')[1].split('>')[0].replace('perf::', '').replace(' ', '')
        return f'{base}<{args}>'
    if 'perf::hashgrid_bwd_kernel<false>' in full:          # the predicated fp32 repair launch (a no-op dispatch in these runs)
        return 'hashgrid_bwd_kernel<false> (redo, no-op)'
    for k in KERNELS:
        if 'perf::' + k + '<' in full or 'perf::' + k + '(' in full or full.strip().endswith(k) or ('perf::' + k) in full:
            return k
    return None


def fold_counters(pattern):
    """{kernel: {counter: mean per launch, 'launches': n}} over all counter_collection CSVs matching the pattern."""
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    files = glob.glob(os.path.join(SRC, pattern), recursive=True)
    for path in files:
        for r in csv.DictReader(open(path)):
            k = kname(r['Kernel_Name'])
            if k is None:
                continue
            tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
    out = {}
    for k in tot:
        out[k] = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
        out[k]['launches'] = max(cnt[k].values())
    return out, files


def copy_raw(files, sub):
    os.makedirs(os.path.join(RAW, sub), exist_ok=True)
    for f in files:           # (rows of this library's kernels only: the passes also record every torch / set-up kernel)
        with open(f) as src, open(os.path.join(RAW, sub, os.path.basename(os.path.dirname(f)) + '_' + os.path.basename(f)), 'w') as dst:
            for i, line in enumerate(src):
                if i == 0 or 'perf::' in line:
                    dst.write(line)


def main():
    os.makedirs(RAW, exist_ok=True)
    # ---- kernel trace stats
    stats = glob.glob(os.path.join(SRC, 'kt', '**', '*kernel_stats.csv'), recursive=True)
    dur = {}
    if stats:
        shutil.copy(stats[0], os.path.join(DST, 'r06_train_geo_kernel_stats.csv'))
        for r in csv.DictReader(open(stats[0])):
            k = kname(r['Name'])
            if k:
                dur.setdefault(k, []).append((int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:90]))
    # ---- HBM traffic
    f, ff = fold_counters('pmc_FETCH_SIZE/**/*counter_collection.csv')
    w, wf = fold_counters('pmc_WRITE_SIZE/**/*counter_collection.csv')
    copy_raw(ff, 'pmc_fetch'); copy_raw(wf, 'pmc_write')
    kernels = defaultdict(lambda: {'fetch_size_kb_raw': 0.0, 'write_size_kb': 0.0, 'parts': {}})
    main_of = {}
    entry = dict(ENTRY)
    for k in list(f) + list(w):                      # templated MLP kernels: one C-ABI entry per template instance
        if '<' in k and k.split('<')[0] in MLP_ENTRY:
            entry[k] = MLP_ENTRY[k.split('<')[0]] + '<' + k.split('<')[1]
    for k, e in entry.items():
        if k in f or k in w:
            main_of.setdefault(e, k)
    for k, e in entry.items():
        if k in f and e in main_of:
            n = f[k]['launches'] / f[main_of[e]]['launches']
            kernels[e]['fetch_size_kb_raw'] += f[k]['FETCH_SIZE'] * n
            kernels[e]['parts'].setdefault(k, {})['fetch_kb_raw'] = round(f[k]['FETCH_SIZE'] * n, 1)
        if k in w and e in main_of:
            n = w[k]['launches'] / w[main_of[e]]['launches']
            kernels[e]['write_size_kb'] += w[k]['WRITE_SIZE'] * n
            kernels[e]['parts'].setdefault(k, {})['write_kb'] = round(w[k]['WRITE_SIZE'] * n, 1)
    for e, d in kernels.items():
        d['fetch_size_kb_raw'] = round(d['fetch_size_kb_raw'], 1); d['write_size_kb'] = round(d['write_size_kb'], 1)
        d['hbm_bytes_per_launch'] = int((2 * d['fetch_size_kb_raw'] + d['write_size_kb']) * 1024)
    json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no tracing), bench.py default workload (8192 rays x 128 marched spp, '
                       'bf16, geometry step with the sampling-pass features reused; eager steps), mean per C-ABI call over the run (all kernels the '
                       'call launches); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B).  Raw CSVs: profiles/r06_raw/',
               'workload': 'train_geo:8192x128:prepass=1', 'kernels': kernels}, open(os.path.join(DST, 'r06_pmc_traffic.json'), 'w'), indent=1)
    # ---- MFMA
    m, mf = fold_counters('pmc_mfma/**/*counter_collection.csv')
    out = {'command': 'rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py '
                      '--steps 10 --warmup 3 --no-graph ... (tools/exp/r06_profile.sh; counters of THIS round\'s build, per launch); durations: '
                      'rocprofv3 --kernel-trace --stats of the default bench command, no counters (profiles/r06_train_geo_kernel_stats.csv)',
           'definition': 'mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); achieved = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 FLOP / '
                         'duration against the 2.5 PFLOP/s dense bf16 peak (one v_mfma_f32_32x32x16_bf16 = 64 MOPS = 32,768 FLOP) -- PADDED work; '
                         'algorithmic_frac_of_mfma_peak = SURVEY.md 8(d) unpadded FLOP per sample x 1,048,576 samples / duration / 2.5 PFLOP/s', 'kernels': {}}
    for k in sorted(m):
        if not k.startswith('mlp_'):
            continue
        row = {c: round(v, 1) for c, v in m[k].items()}
        if k in dur:
            calls, avg_us, _ = max(dur[k])
            row['duration_us'] = round(avg_us, 2)
            row['mfma_utilisation'] = round(m[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg_us * 1e-6 * 2.4e9 * 1024), 4)
            row['achieved_TFLOPs'] = round(m[k]['SQ_INSTS_VALU_MFMA_MOPS_BF16'] * 512 / (avg_us * 1e-6) / 1e12, 1)
            row['frac_of_mfma_peak'] = round(row['achieved_TFLOPs'] / 2500.0, 4)
            algo = algorithmic_flop(k)
            if algo:
                row['algorithmic_flop_per_sample'] = algo
                row['algorithmic_frac_of_mfma_peak'] = round(algo * SAMPLES / (avg_us * 1e-6) / 2.5e15, 4)
        out['kernels'][k] = row
    json.dump(out, open(os.path.join(DST, 'r06_mfma_util.json'), 'w'), indent=1)
    # ---- the colour phase's step: kernel stats + MFMA utilisation of its MLP kernels
    stats_app = glob.glob(os.path.join(SRC, 'kt_app', '**', '*kernel_stats.csv'), recursive=True)
    dur_app = {}
    if stats_app:
        shutil.copy(stats_app[0], os.path.join(DST, 'r06_train_app_kernel_stats.csv'))
        for r in csv.DictReader(open(stats_app[0])):
            k = kname(r['Name'])
            if k:
                dur_app.setdefault(k, []).append((int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:90]))
    ma, maf = fold_counters('pmc_mfma_app/**/*counter_collection.csv')
    out_app = {'command': 'as r06_mfma_util.json, with --mode train_app (the colour phase\'s step at bench scale: 8192 rays x 128 samples)', 'kernels': {}}
    for k in sorted(ma):
        if not k.startswith('mlp_'):
            continue
        row = {c: round(v, 1) for c, v in ma[k].items()}
        if k in dur_app:
            calls, avg_us, _ = max(dur_app[k])
            row['duration_us'] = round(avg_us, 2)
            row['mfma_utilisation'] = round(ma[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg_us * 1e-6 * 2.4e9 * 1024), 4)
            row['achieved_TFLOPs'] = round(ma[k]['SQ_INSTS_VALU_MFMA_MOPS_BF16'] * 512 / (avg_us * 1e-6) / 1e12, 1)
            row['frac_of_mfma_peak'] = round(row['achieved_TFLOPs'] / 2500.0, 4)
            algo = algorithmic_flop(k)
            if algo:
                row['algorithmic_flop_per_sample'] = algo
                row['algorithmic_frac_of_mfma_peak'] = round(algo * SAMPLES / (avg_us * 1e-6) / 2.5e15, 4)
        out_app['kernels'][k] = row
    json.dump(out_app, open(os.path.join(DST, 'r06_mfma_util_train_app.json'), 'w'), indent=1)
    stats_ep = glob.glob(os.path.join(SRC, 'kt_ep', '**', '*kernel_stats.csv'), recursive=True)
    if stats_ep:
        shutil.copy(stats_ep[0], os.path.join(DST, 'r06_episode_kernel_stats.csv'))
    stats_c4 = glob.glob(os.path.join(SRC, 'kt_c4', '**', '*kernel_stats.csv'), recursive=True)
    if stats_c4:
        shutil.copy(stats_c4[0], os.path.join(DST, 'r06_config4_kernel_stats.csv'))
    print(json.dumps(out, indent=1)[:3000])


if __name__ == '__main__':
    main()
