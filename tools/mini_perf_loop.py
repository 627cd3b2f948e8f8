"""A miniature of PeRF's progressive loop (core_exp_runner.py:170-221, modules/scene/nerf.py:321-358,
modules/dataset/sup_info.py:99-120,261-302) on the synthetic room with an occluding box: train an episode on the registered
panoramas, move to a new position, render the distance there, ask which pixels the registered panoramas have already seen
(get_pano_visibility_mask), "inpaint" the rest with the scene's ground truth, drop what contradicts registered geometry
(geo_check), register the new panorama's valid pixels (PanoSupInfo's rules), rebuild the occupancy, train again.  Exercises
what the single-panorama benchmarks do not: supervision pools of several origins, rays that do not start at the centre,
the visibility / reprojection kernels on rendered distances, the occupancy splat of several panoramas.

  python tools/mini_perf_loop.py [--views 5] [--geo 1000] [--app 500] [--height 256]"""
import argparse, json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perf_amd import synthetic
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays, psnr

ap = argparse.ArgumentParser()
ap.add_argument('--views', type=int, default=5)
ap.add_argument('--geo', type=int, default=1000)
ap.add_argument('--app', type=int, default=500)
ap.add_argument('--height', type=int, default=256)
ap.add_argument('--dtype', default='bf16')
args = ap.parse_args()
torch.manual_seed(0)
H, W = args.height, 2 * args.height
dev = 'cuda'
scene = NeRFScene(dtype=args.dtype)
pool = SupInfoPool()


def pose_at(x, y, z):
    p = torch.eye(4, device=dev); p[:3, 3] = torch.tensor([x, y, z], device=dev)
    return p


def truth(pose):
    rays = gen_pano_rays(pose, H, W, device=dev)
    dist, rgb = synthetic.room_with_box(rays.o, rays.d)
    return rays, dist, rgb


ring = [pose_at(0.22 * math.cos(a), 0.18 * math.sin(a), 0.04 * math.sin(2 * a)) for a in [2 * math.pi * k / max(args.views - 1, 1) for k in range(args.views - 1)]]
held_out = pose_at(0.12, -0.10, 0.05)
rays0, dist0, rgb0 = truth(pose_at(0, 0, 0))
pool.register_sup_info(pose_at(0, 0, 0), torch.ones(H, W, 1, device=dev), rgb0, dist0)
rows = []
for k in range(args.views):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    scene.train_one_episode(pool, args.geo, args.app)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    scene.set_eval()
    ev = {}
    for name, pose in (('first_pano', pose_at(0, 0, 0)), ('held_out', held_out)):
        r, d_gt, c_gt = truth(pose)
        out = scene.render(r, ['rgb', 'distance'])
        ev[name] = {'psnr_dB': round(psnr(out['rgb'], c_gt), 2), 'mean_abs_distance_err': round(float((out['distance'] - d_gt).abs().mean()), 5)}
    c = scene.sample_counters.tolist()
    row = {'episode': k, 'panoramas': pool.n_panos, 'supervision_rays': len(pool), 'train_s': round(t1 - t0, 3), **ev,
           'skipped_steps': int(c[4] + c[5])}
    if k < len(ring):
        pose = ring[k]
        rays, d_gt, c_gt = truth(pose)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        seen = scene.get_pano_visibility_mask(pool, rays)                       # [H, W] 1 = some registered panorama saw the point
        torch.cuda.synchronize(); t3 = time.perf_counter()
        seen = seen.reshape(H, W, 1).float()
        ok = pool.geo_check(rays, d_gt).reshape(H, W, 1).float()                # 1 = the "inpainted" geometry contradicts nobody
        new = (1.0 - seen) * ok
        before = len(pool)
        pool.register_sup_info(pose, new, c_gt, d_gt)
        row.update({'new_view_seen_fraction': round(float(seen.mean()), 4), 'geo_check_ok_fraction': round(float(ok.mean()), 4),
                    'new_rays_registered': len(pool) - before, 'visibility_mask_s': round(t3 - t2, 4)})
    rows.append(row)
    print(json.dumps(row), flush=True)
print(json.dumps({'config': f'{args.views} episodes of {args.geo} + {args.app} iterations, {W}x{H} panoramas, room with an occluding box, {args.dtype}',
                  'episodes': rows}))
