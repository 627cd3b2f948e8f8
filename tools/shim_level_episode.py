"""What an UNMODIFIED PeRF NeRFScene costs over the operator shims (tinycudann / nerfacc / torch_efficient_distloss module API,
torch autograd, torch.optim.Adam, the reference's 256-call occupancy warm-up of modules/scene/nerf.py:147-168), measured on
the mirror configured to take exactly those paths -- the reference class itself cannot run on the GPU box (its tree and its
other dependencies are absent).  Beside it: the mirror's own episode (explicit kernel chains, hipGraph replays, one-launch
occupancy build), which install_shims(scene=True) puts behind the reference's import names.

  python tools/shim_level_episode.py [--geo 3000] [--app 1500]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perf_amd import synthetic
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays, psnr

ap = argparse.ArgumentParser()
ap.add_argument('--geo', type=int, default=3000)
ap.add_argument('--app', type=int, default=1500)
ap.add_argument('--dtype', default='bf16')
args = ap.parse_args()
H, W = 1024, 2048
rays = gen_pano_rays(torch.eye(4), H, W)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
out = {'config': f'{args.geo} + {args.app} iterations of 8192 rays, {W}x{H} synthetic room, {args.dtype}'}
for tag, shim_level in (('mirror (explicit chains, graphs, one-launch occupancy)', False), ('shim level (autograd formulation + 256-call warm-up)', True)):
    torch.manual_seed(0)
    scene = NeRFScene(dtype=args.dtype)
    if shim_level:
        scene.fused_steps = False; scene.fused_adam = False
    for rep in range(2):                                        # second episode: warm allocator / kernels
        torch.cuda.synchronize(); t0 = time.perf_counter()
        scene.set_train()
        scene.prepare_occupancy(pool, 'reference' if shim_level else 'direct')
        torch.cuda.synchronize(); t1 = time.perf_counter()
        scene.train_one_episode(pool, args.geo, args.app, warmup='reference' if shim_level else 'direct')
        torch.cuda.synchronize(); t2 = time.perf_counter()
    pre_grid, _ = pool.gen_occ_grid(256)
    diff = int((scene.estimator.binaries.reshape(-1) != pre_grid.bool()).sum())
    scene.set_eval()
    res = scene.render(rays, ['rgb'])
    out[tag] = {'occupancy_build_s': round(t1 - t0, 4), 'episode_s_incl_occupancy': round(t2 - t1, 4),
                'ms_per_step_incl_occupancy': round((t2 - t1) / (args.geo + args.app) * 1e3, 4), 'psnr_dB': round(psnr(res['rgb'], rgb), 2),
                'cells_differing_from_pre_grid': diff}
print(json.dumps(out, indent=1))
