"""One PeRF training episode at the reference's settings (modules/scene/nerf.py:137-184, configs/nerf.yaml): occupancy
from the supervision, 3000 geometry + 1500 colour iterations of 8,192 rays drawn from a 1024x2048 panorama, reference-
faithful variable-count sampling (step 5e-4, early stop 1e-4), then a full-panorama evaluation.

  python tools/train_episode.py [--geo 3000] [--app 1500] [--dtype bf16]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perf_amd import synthetic, tcnn
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays, psnr

ap = argparse.ArgumentParser()
ap.add_argument('--geo', type=int, default=3000)
ap.add_argument('--app', type=int, default=1500)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--head', type=int, default=-1, help='renderer.head_samples for the run (0 = one-phase sampler; default: the renderer default)')
args = ap.parse_args()
torch.manual_seed(0)
scene = NeRFScene(dtype=args.dtype)
if args.head >= 0:
    scene.renderer.head_samples = args.head or None
H, W = 1024, 2048
rays = gen_pano_rays(torch.eye(4), H, W)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
marks = {}
def cb(phase, i):
    if i == 0:
        torch.cuda.synchronize(); marks[phase] = time.perf_counter()
torch.cuda.synchronize(); t0 = time.perf_counter()
scene.train_one_episode(pool, args.geo, args.app, callback=cb)
torch.cuda.synchronize(); t1 = time.perf_counter()
res = scene.render_once(pool.rand_ray_color_data(8192)[0], ['ray_indices'], app_inference=True)
spp = res['ray_indices'].numel() / 8192
scene.set_eval()
torch.cuda.synchronize(); t2 = time.perf_counter()
out = scene.render(rays, ['rgb', 'distance'])
torch.cuda.synchronize(); t3 = time.perf_counter()
# per-kernel times of the reference-faithful steps on the trained scene: 10 eager steps of each kind with HIP events around
# every C-ABI launch (the torch glue ops between them are not seen)
from perf_amd import ops
kern = {}
scene.set_train()
scene.renderer.sample_capacity = 8192 * scene.TRAIN_SAMPLES_PER_RAY
for kind in ('geo', 'app'):
    net = scene.nerf.geo_mlp if kind == 'geo' else scene.nerf.app_mlp
    opt = scene.make_optimizer(net, 0.0)
    fn = scene.train_one_step_geo if kind == 'geo' else scene.train_one_step_app
    for _ in range(3):
        fn(opt, pool, progress=0.9)
    ops.start_kernel_timing()
    for _ in range(10):
        fn(opt, pool, progress=0.9)
    kern[kind] = {k: (round(n / 10, 1), round(ms * 1e3, 1)) for k, (n, ms) in sorted(ops.stop_kernel_timing().items(), key=lambda kv: -kv[1][0] * kv[1][1])}
    kern[kind + '_sum_us_per_step'] = round(sum(n * us for n, us in kern[kind].values()), 1)
    kern[kind + '_launches_per_step'] = round(sum(n for n, us in kern[kind].values()), 1)
print(json.dumps({'config': f'{args.geo} geometry + {args.app} colour iterations, 8192-ray batches, {W}x{H} supervision panorama, {args.dtype}',
                  'episode_s': t1 - t0, 'occupancy_s': marks['geo'] - t0, 'geo_phase_s': marks['app'] - marks['geo'], 'app_phase_s': t1 - marks['app'],
                  'ms_per_geo_step': (marks['app'] - marks['geo']) / args.geo * 1e3, 'ms_per_app_step': (t1 - marks['app']) / args.app * 1e3,
                  'train_batch_mean_samples_per_ray': spp, 'grid_gradient_mode_at_end': tcnn.GRID_GRAD_ACCUM,
                  'eval_full_pano_s': t3 - t2, 'psnr_dB': psnr(out['rgb'], rgb), 'mean_abs_distance_err': float((out['distance'] - dist).abs().mean()),
                  'kernels_per_step (launches, us per launch)': kern}, indent=1))
