"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as /opt/skills/guides/MI355X_MICROARCH.md asks)
into profiles/<name>.json: HBM bytes per launch of every perf:: kernel, grouped by the C-ABI entry point.

  python tools/pmc_traffic.py gpurun_out/prof4_fetch/f_counter_collection.csv gpurun_out/prof4_write/w_counter_collection.csv \
         profiles/r01_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies 128-byte fetches as 64 bytes, hence the x2 on FETCH_SIZE
(calibrated on adam_kernel: 4 x 26.6 MB read -> 51.9 MB raw)."""
import csv, json, sys
from collections import defaultdict

ENTRY = {'hashgrid_bwd_kernel': 'perf_hashgrid_bwd', 'tile_codes_kernel': 'perf_hashgrid_bwd', 'hashgrid_bwd_reduce_kernel': 'perf_hashgrid_bwd',
         'hashgrid_fwd_kernel': 'perf_hashgrid_fwd', 'mlp_bwd_kernel': 'perf_mlp_bwd', 'mlp_reduce_kernel': 'perf_mlp_bwd',
         'mlp_fwd_kernel': 'perf_mlp_fwd', 'adam_kernel': 'perf_adam_step_dev'}


def fold(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name']
        for k, e in ENTRY.items():
            if 'perf::' + k in name:
                tot[(e, k)] += float(r['Counter_Value']); cnt[(e, k)] += 1
    return tot, cnt


def main(fetch_csv, write_csv, out, workload='train_geo:8192x128:prepass=1'):
    f, fc = fold(fetch_csv, 'FETCH_SIZE')
    w, wc = fold(write_csv, 'WRITE_SIZE')
    kernels = defaultdict(lambda: {'fetch_size_kb_raw': 0.0, 'write_size_kb': 0.0, 'parts': {}})
    # launches of an entry point = launches of its main kernel (first listed in ENTRY)
    main_of = {}
    for k, e in ENTRY.items():
        main_of.setdefault(e, k)
    for (e, k), v in f.items():
        n = fc[(e, main_of[e])]
        kernels[e]['fetch_size_kb_raw'] += v / n
        kernels[e]['parts'].setdefault(k, {})['fetch_kb_raw'] = round(v / n, 1)
    for (e, k), v in w.items():
        n = wc[(e, main_of[e])]
        kernels[e]['write_size_kb'] += v / n
        kernels[e]['parts'].setdefault(k, {})['write_kb'] = round(v / n, 1)
    for e, d in kernels.items():
        d['fetch_size_kb_raw'] = round(d['fetch_size_kb_raw'], 1); d['write_size_kb'] = round(d['write_size_kb'], 1)
        d['hbm_bytes_per_launch'] = int((2 * d['fetch_size_kb_raw'] + d['write_size_kb']) * 1024)
    note = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py default workload (8192 rays x 128 marched spp, bf16, reference step incl. the sampling-pass density evaluation; eager steps), per '
            'C-ABI call (all kernels the call launches); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)')
    json.dump({'note': note, 'workload': workload, 'kernels': kernels}, open(out, 'w'), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:5])
