"""Generates tests/golden/psnr_curve.json: the fp32 CPU oracle's PSNR@iter curve on the synthetic room (256x512 panorama,
1024-ray batches, 300 geometry + 300 colour iterations = 600 >= 500, reference-faithful sampling) for several seeds.
Runs in the build container (CPU, minutes per seed); tests/test_gpu_psnr.py replays the same schedule on the HIP path on
the GPU box and asserts |mean delta PSNR| <= 0.1 dB for the default dtype (north_star).

  python tests/golden/make_psnr_curve.py [n_seeds] [out_path] [scene]      (scene: room | doorway | pillars -> psnr_curve[_<scene>].json)
  python tests/golden/make_psnr_curve.py spread - [scene]                  (adds the oracle's OWN sensitivity to the same file)
  python tests/golden/make_psnr_curve.py quant16 - [scene] [bf16|fp16]     (adds the oracle's 16-bit-emulated curves to the same file)

`spread`: the same fp32 CPU schedule of seed 0 run twice more from an initialisation moved by ONE ULP (every parameter of both
networks to its fp32 neighbour above / below): what a perturbation far below any 16-bit effect does to PSNR@iter of this chaotic
optimisation.  Stored as `oracle_spread` = {runs: [curve, curve], per mark the largest pairwise |difference| among (seed 0,
above, below)}; tests/test_gpu_psnr.py asserts single seeds against max(0.1 dB, that spread)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests import psnr_parity_lib as P

H, W, BATCH, N_GEO, N_APP = 256, 512, 1024, 300, 300
MARKS = (150, 300)


def spread_main():
    torch.set_num_threads(int(os.environ.get('PERF_ORACLE_THREADS', min(os.cpu_count() or 1, 32))))
    scene_name = sys.argv[3] if len(sys.argv) > 3 else 'room'
    default = 'psnr_curve.json' if scene_name == 'room' else f'psnr_curve_{scene_name}.json'
    out_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != '-' else os.path.join(ROOT, 'tests', 'golden', default)
    res = json.load(open(out_path))
    cfg = res['config']
    assert cfg.get('scene', 'room') == scene_name and cfg['lattice'] == P.O.DEFAULT_LATTICE
    scene = P.make_scene(*cfg['pano'], scene_name)
    base = next(r for r in res['seeds'] if r['seed'] == 0)
    draws = P.make_draws(scene[0].shape[0], cfg['batch'], cfg['geo_iters'] + cfg['app_iters'], 0)
    assert P.draws_digest(draws) == base['draws_digest']
    sp = res.get('oracle_spread') or {'seed': 0, 'perturbation': 'every parameter of both networks moved to its fp32 neighbour (one ulp) above / below', 'runs': []}
    for k, towards in enumerate((float('inf'), -float('inf'))):
        if k < len(sp['runs']):
            continue
        geo0, app0 = P.init_params(0)
        geo0 = torch.nextafter(geo0, torch.full_like(geo0, towards)); app0 = torch.nextafter(app0, torch.full_like(app0, towards))
        t = time.time()
        curve = P.run_oracle(scene, geo0, app0, draws, cfg['geo_iters'], cfg['app_iters'], tuple(cfg['marks']),
                             log=lambda m: print(f'{scene_name} spread run {k}: {m}', flush=True))
        curve['seconds'] = round(time.time() - t, 1)
        sp['runs'].append(curve)
        curves = [base['oracle']] + sp['runs']
        sp['max_abs_delta_db'] = {f'psnr@app{m}': max(abs(a[f'psnr@app{m}'] - b[f'psnr@app{m}']) for a in curves for b in curves) for m in cfg['marks']}
        res['oracle_spread'] = sp
        json.dump(res, open(out_path, 'w'), indent=1)
        print(scene_name, 'spread so far', sp['max_abs_delta_db'], flush=True)


def quant_main():
    """`oracle_16bit` = {dtype: {seed: curve}}: the same schedule with the oracle's 16-bit emulation (psnr_parity_lib.run_oracle
    quant=...), and per mark the largest |16-bit-emulated - fp32| over the seeds: what rounding parameters and features to the
    storage type ALONE does to PSNR@iter -- the reference's own tcnn path stores fp16.  tests/test_gpu_psnr.py bounds single seeds
    of (HIP - fp32 oracle) by max(0.1 dB, that figure)."""
    torch.set_num_threads(int(os.environ.get('PERF_ORACLE_THREADS', min(os.cpu_count() or 1, 32))))
    scene_name = sys.argv[3] if len(sys.argv) > 3 else 'room'
    dtype = sys.argv[4] if len(sys.argv) > 4 else 'bf16'
    default = 'psnr_curve.json' if scene_name == 'room' else f'psnr_curve_{scene_name}.json'
    out_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != '-' else os.path.join(ROOT, 'tests', 'golden', default)
    res = json.load(open(out_path))
    cfg = res['config']
    assert cfg.get('scene', 'room') == scene_name and cfg['lattice'] == P.O.DEFAULT_LATTICE
    scene = P.make_scene(*cfg['pano'], scene_name)
    for row in res['seeds']:
        sd = row['seed']
        res = json.load(open(out_path))                     # (the spread job may be writing the same file: merge, do not clobber)
        block = res.setdefault('oracle_16bit', {}).setdefault(dtype, {'curves': {}})
        if str(sd) in block['curves']:
            continue
        geo0, app0 = P.init_params(sd)
        draws = P.make_draws(scene[0].shape[0], cfg['batch'], cfg['geo_iters'] + cfg['app_iters'], sd)
        assert P.draws_digest(draws) == row['draws_digest']
        t = time.time()
        curve = P.run_oracle(scene, geo0, app0, draws, cfg['geo_iters'], cfg['app_iters'], tuple(cfg['marks']), quant=dtype,
                             log=lambda m: print(f'{scene_name} {dtype} seed {sd}: {m}', flush=True))
        curve['seconds'] = round(time.time() - t, 1)
        res = json.load(open(out_path))
        block = res.setdefault('oracle_16bit', {}).setdefault(dtype, {'curves': {}})
        block['curves'][str(sd)] = curve
        rows = {str(r['seed']): r['oracle'] for r in res['seeds']}
        block['max_abs_delta_db'] = {f'psnr@app{m}': max(abs(c[f'psnr@app{m}'] - rows[s][f'psnr@app{m}']) for s, c in block['curves'].items())
                                     for m in cfg['marks']}
        json.dump(res, open(out_path, 'w'), indent=1)
        print(scene_name, dtype, 'seed', sd, {k: round(curve[k] - rows[str(sd)][k], 4) for k in curve if k.startswith('psnr')}, flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'spread':
        return spread_main()
    if len(sys.argv) > 1 and sys.argv[1] == 'quant16':
        return quant_main()
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.set_num_threads(int(os.environ.get('PERF_ORACLE_THREADS', min(os.cpu_count() or 1, 32))))
    scene_name = sys.argv[3] if len(sys.argv) > 3 else 'room'
    scene = P.make_scene(H, W, scene_name)
    default = 'psnr_curve.json' if scene_name == 'room' else f'psnr_curve_{scene_name}.json'
    out_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != '-' else os.path.join(ROOT, 'tests', 'golden', default)
    res = {'config': {'pano': [H, W], 'batch': BATCH, 'geo_iters': N_GEO, 'app_iters': N_APP, 'marks': list(MARKS),
                      'geo_marks': list(P.GEO_MARKS), 'torch': torch.__version__, 'lattice': P.O.DEFAULT_LATTICE}, 'seeds': []}
    if scene_name != 'room':
        res['config']['scene'] = scene_name

    if os.path.exists(out_path):
        old = json.load(open(out_path))
        if old.get('config') == res['config']:
            res = old
    done = {r['seed'] for r in res['seeds']}
    for sd in range(n_seeds):
        if sd in done:
            continue
        geo0, app0 = P.init_params(sd)
        draws = P.make_draws(scene[0].shape[0], BATCH, N_GEO + N_APP, sd)
        t = time.time()
        curve = P.run_oracle(scene, geo0, app0, draws, N_GEO, N_APP, MARKS, log=lambda m: print(f'seed {sd}: {m}', flush=True))
        res['seeds'].append({'seed': sd, 'draws_digest': P.draws_digest(draws), 'oracle': curve, 'seconds': round(time.time() - t, 1)})
        json.dump(res, open(out_path, 'w'), indent=1)
        print('seed', sd, curve, flush=True)


if __name__ == '__main__':
    main()
