"""BASELINE config 4 (render_dense, core_exp_runner.py:223-247): a dense camera trajectory through a trained scene,
512x1024 panoramic frames, fp16 inference, reference-faithful variable-count sampling.

  python tools/render_dense.py [--poses 600] [--geo-steps 300] [--app-steps 150]

Frames are rendered back to back on the device (the reference writes PNGs and a video in between: host IO, out of
scope); prints frames/s, rays/s, ray-samples/s actually evaluated, and a checksum of the frames."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from perf_amd import ops, synthetic
from perf_amd.pose_sampler import CirclePoseSampler, DenseTravelPoseSampler
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays

ap = argparse.ArgumentParser()
ap.add_argument('--poses', type=int, default=600)
ap.add_argument('--geo-steps', type=int, default=300)
ap.add_argument('--app-steps', type=int, default=150)
ap.add_argument('--dtype', default='fp16')
ap.add_argument('--head', type=int, default=2, help='two-phase sampler: density first on that many samples per ray (0 = one phase)')
ap.add_argument('--batch', type=int, default=32768, help='rays per graph-captured eval batch (the reference hard-codes 32768, nerf.py:86)')
args = ap.parse_args()

torch.manual_seed(0); np.random.seed(0)
scene = NeRFScene(dtype=args.dtype)
H, W = 1024, 2048
rays = gen_pano_rays(torch.eye(4), H, W)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
# The anchors only need the panorama's distance map: the dense trajectory (10,000-step tour annealing on the host, 0.2 s) is
# started in a worker process NOW and runs beside the training below (off config 4's critical path)
sparse = CirclePoseSampler(dist.reshape(H, W).cpu(), traverse_ratios=[.2, .4, .6], n_anchors_per_ratio=[8, 8, 8])
rng0 = np.random.get_state()
t0 = time.perf_counter()
dense_future = DenseTravelPoseSampler.start(sparse, n_dense_poses=args.poses)
t_start_call = time.perf_counter() - t0
scene.set_train(); scene.prepare_occupancy(pool)
tc = scene.train_conf
scene.nerf.reset_geo()
opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
for i in range(args.geo_steps):
    scene.update_lr(opt, tc.geo_optimizer, i / args.geo_steps)
    scene.train_one_step_geo(opt, pool, progress=i / args.geo_steps)
opt = scene.make_optimizer(scene.nerf.app_mlp, 0.0)
for i in range(args.app_steps):
    scene.update_lr(opt, tc.app_optimizer, i / args.app_steps)
    scene.train_one_step_app(opt, pool, progress=i / args.app_steps)
torch.cuda.synchronize()

# pose samplers run on the host (utils of the reference, restated in perf_amd/pose_sampler.py)
t0 = time.perf_counter()
dense = dense_future.result()                                       # (finished long ago: it ran beside the training)
t_wait = time.perf_counter() - t0
from perf_amd import pose_sampler as _ps
_ps._DENSE_CACHE.clear(); rng1 = np.random.get_state(); np.random.set_state(rng0)
t0 = time.perf_counter()
dense_seq = DenseTravelPoseSampler(sparse, n_dense_poses=args.poses)   # the same trajectory computed in line, for the record
t_sampler = time.perf_counter() - t0
assert torch.equal(dense_seq.sample_poses, dense.sample_poses) and np.array_equal(np.random.get_state()[1], rng1[1])
poses = []
for i in range(dense.n_poses):
    p = dense.sample_pose(i).clone().float()
    p[:3, :3] = torch.eye(3)                                        # core_exp_runner.py:232
    poses.append(p)

fh, fw = 512, 1024
def frame_eager(p):
    r = gen_pano_rays(p, fh, fw)
    return scene.render(r, ['rgb', 'distance'], batch_size=args.batch)

# ONE hipGraph per frame: ray generation from a device-resident pose + 16 eval batches of 32,768 rays (nerf.py:86)
scene.renderer.head_samples = args.head or None
frame = scene.make_graphed_render(fh, fw, ('rgb', 'distance'), batch_size=args.batch)
for p in poses[:3]:
    frame(p)
torch.cuda.synchronize()
ref = frame_eager(poses[1]); got = frame(poses[1])
same = bool(torch.equal(ref['rgb'], got['rgb']) and torch.equal(ref['distance'], got['distance']))
ops.start_kernel_timing()
frame_eager(poses[0]); torch.cuda.synchronize()
kern = ops.stop_kernel_timing()
t0 = time.perf_counter()
for p in poses:
    last = frame(p)
torch.cuda.synchronize()
t = time.perf_counter() - t0
checksum = float(last['rgb'].double().sum())
t0 = time.perf_counter()
for p in poses[:60]:
    frame_eager(p)
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / 60
from perf_amd.scene import Rays as _Rays
_r0 = gen_pano_rays(poses[0], fh, fw)
scene.set_eval(); scene.renderer.sample_capacity = fh * fw * 64
with torch.no_grad():
    _c = scene.render_once(_Rays(_r0.o.reshape(-1, 3), _r0.d.reshape(-1, 3)), ['n_marched_dev', 'n_samples_dev'])
scene.renderer.sample_capacity = None
print(json.dumps({'config': 'render_dense: %d poses, %dx%d panoramic frames in %d hipGraph-captured %d-ray batches, %s, variable-count sampling' % (len(poses), fw, fh, (fh * fw + args.batch - 1) // args.batch, args.batch, args.dtype),
                  'frame0_marched_samples': int(_c['n_marched_dev'].item()), 'frame0_kept_samples': int(_c['n_samples_dev'].item()),
                  'frames_per_s': len(poses) / t, 'rays_per_s': len(poses) * fh * fw / t, 'seconds': t,
                  'eager_sync_free_frames_per_s': 1.0 / t_eager, 'graphed_frame_equals_eager_frame': same,
                  'per_ray_sample_capacity': frame.state['per_ray'], 'head_samples': args.head,
                  'pose_sampler_host_s': t_sampler, 'pose_sampler_start_call_s': t_start_call, 'pose_sampler_wait_s': t_wait,
                  'wall_s_including_sampler': {'overlapped (started before training, as this tool does)': t + t_start_call + t_wait,
                                               'in line (round 2)': t + t_sampler},
                  'last_frame_rgb_sum': checksum,
                  'kernel_ms_one_frame': {k: round(n * ms, 3) for k, (n, ms) in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])}}, indent=1))
