"""PSNR@iter parity as a dev tool: the HIP path (bf16 / fp16, fixed-point or fp32 gradient accumulation) against the fp32
CPU oracle's committed curve (tests/golden/psnr_curve.json; the harness is tests/psnr_parity_lib.py, the asserted version
tests/test_gpu_psnr.py).   python tools/psnr_parity.py [room doorway pillars]  ->  gpurun_out/psnr_parity.json

Splits what moves a seed away from the oracle: `fixed` vs `fp32` accumulation of the grid gradient isolates the fixed-point fields,
bf16 vs fp16 the storage type (profiles/r06_psnr_split.json is this tool's output folded per family)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests import psnr_parity_lib as P


def ensemble(fam, dtype='bf16'):
    """Per seed: the HIP run from the golden initialisation and from that initialisation moved by ONE fp32 ULP up / down (what
    make_psnr_curve.py spread does to the oracle) -> how far a 16-bit path moves under a perturbation that small."""
    name = 'psnr_curve.json' if fam == 'room' else f'psnr_curve_{fam}.json'
    golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', name)))
    cfg = golden['config']
    scene = P.make_scene(*cfg['pano'], fam)
    rows = []
    for row in golden['seeds']:
        geo0, app0 = P.init_params(row['seed'])
        draws = P.make_draws(scene[0].shape[0], cfg['batch'], cfg['geo_iters'] + cfg['app_iters'], row['seed'])
        runs = []
        for towards in (None, float('inf'), -float('inf')):
            g, a = geo0, app0
            if towards is not None:
                g = torch.nextafter(geo0, torch.full_like(geo0, towards)); a = torch.nextafter(app0, torch.full_like(app0, towards))
            got = P.run_hip(scene, g, a, draws, cfg['geo_iters'], cfg['app_iters'], tuple(cfg['marks']), dtype, 'fixed')
            runs.append({k: round(got[k] - row['oracle'][k], 4) for k in got if k.startswith('psnr')})
        rows.append({'seed': row['seed'], 'hip_minus_oracle': {'nominal': runs[0], 'one_ulp_up': runs[1], 'one_ulp_down': runs[2]}})
        print(fam, dtype, row['seed'], runs, flush=True)
    return rows


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'ensemble':
        out = {fam: {dt: ensemble(fam, dt) for dt in ('bf16', 'fp16')} for fam in (sys.argv[2:] or ['room'])}
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'psnr_ensemble.json'), 'w'), indent=1)
        return
    families = sys.argv[1:] or ['room']
    out = {}
    for fam in families:
        out[fam] = one_family(fam)
    from perf_amd import tcnn
    tcnn.GRID_GRAD_ACCUM = 'fixed'
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out if len(families) > 1 else out[families[0]], open(os.path.join(ROOT, 'gpurun_out', 'psnr_parity.json'), 'w'), indent=1)


def one_family(fam):
    name = 'psnr_curve.json' if fam == 'room' else f'psnr_curve_{fam}.json'
    golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', name)))
    cfg = golden['config']
    scene = P.make_scene(*cfg['pano'], fam)
    res = {'config': cfg, 'runs': []}
    for row in golden['seeds']:
        geo0, app0 = P.init_params(row['seed'])
        draws = P.make_draws(scene[0].shape[0], cfg['batch'], cfg['geo_iters'] + cfg['app_iters'], row['seed'])
        for dtype, accum in (('bf16', 'fixed'), ('fp16', 'fixed'), ('bf16', 'fp32'), ('fp16', 'fp32')):
            t = time.time()
            got = P.run_hip(scene, geo0, app0, draws, cfg['geo_iters'], cfg['app_iters'], tuple(cfg['marks']), dtype, accum)
            delta = {k: round(got[k] - row['oracle'][k], 4) for k in got if k.startswith('psnr')}
            res['runs'].append({'seed': row['seed'], 'dtype': dtype, 'accum': accum, 'hip': got, 'hip_minus_oracle': delta,
                                'seconds': round(time.time() - t, 1)})
            print(fam, row['seed'], dtype, accum, delta, flush=True)
    return res


if __name__ == '__main__':
    main()
