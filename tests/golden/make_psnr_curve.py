"""Generates tests/golden/psnr_curve.json: the fp32 CPU oracle's PSNR@iter curve on the synthetic room (256x512 panorama,
1024-ray batches, 300 geometry + 300 colour iterations = 600 >= 500, reference-faithful sampling) for several seeds.
Runs in the build container (CPU, minutes per seed); tests/test_gpu_psnr.py replays the same schedule on the HIP path on
the GPU box and asserts |mean delta PSNR| <= 0.1 dB for the default dtype (north_star).

  python tests/golden/make_psnr_curve.py [n_seeds] [out_path] [scene]      (scene: room | doorway | pillars -> psnr_curve[_<scene>].json)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests import psnr_parity_lib as P

H, W, BATCH, N_GEO, N_APP = 256, 512, 1024, 300, 300
MARKS = (150, 300)


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
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
