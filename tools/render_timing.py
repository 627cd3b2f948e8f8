"""Reference-faithful (variable-count) full-panorama render: time, mean samples per ray, per-kernel breakdown."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perf_amd import ops, synthetic
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays

torch.manual_seed(0)
scene = NeRFScene(dtype='bf16')
H, W = 1024, 2048
rays = gen_pano_rays(torch.eye(4), H, W)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
scene.set_train(); scene.prepare_occupancy(pool)
# short training so that density is meaningful (early termination behaves like a trained scene)
tc = scene.train_conf
scene.nerf.reset_geo()
opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
t0 = time.perf_counter()
n_geo = 300
for i in range(n_geo):
    scene.update_lr(opt, tc.geo_optimizer, i / n_geo)
    scene.train_one_step_geo(opt, pool, progress=i / n_geo)
torch.cuda.synchronize()
t_train = time.perf_counter() - t0
# count samples of one train batch
res = scene.render_once(pool.rand_ray_color_data(8192)[0], ['ray_indices'], app_inference=True)
spp_train = res['ray_indices'].numel() / 8192
for _ in range(2):
    scene.render(rays, ['rgb', 'distance'])
torch.cuda.synchronize()
ops.start_kernel_timing()
t0 = time.perf_counter()
out = scene.render(rays, ['rgb', 'distance'])
torch.cuda.synchronize()
t_render = time.perf_counter() - t0
kern = ops.stop_kernel_timing()
t0 = time.perf_counter()
out = scene.render(rays, ['rgb', 'distance'])
torch.cuda.synchronize()
t_render_plain = time.perf_counter() - t0
derr = float((out['distance'] - dist).abs().mean())
batch_times = {}
for bsz in (32768, 262144, 1048576):
    scene.render(rays, ['rgb', 'distance'], batch_size=bsz)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o2 = scene.render(rays, ['rgb', 'distance'], batch_size=bsz)
    torch.cuda.synchronize(); batch_times[bsz] = time.perf_counter() - t0
    assert torch.equal(o2['rgb'], out['rgb']) and torch.equal(o2['distance'], out['distance'])
tot = {k: n * ms for k, (n, ms) in kern.items()}
print(json.dumps({'train_300_geo_steps_s': t_train, 'faithful_ms_per_geo_step': t_train / n_geo * 1e3, 'train_batch_mean_spp': spp_train,
                  'full_pano_render_s': t_render_plain, 'rays_per_s': H * W / t_render_plain, 'mean_abs_distance_err': derr, 'full_pano_render_s_by_batch_size': batch_times,
                  'kernel_ms_total': {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}}, indent=1))
