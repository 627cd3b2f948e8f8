"""Soak: 3000 training steps of the bench workload (8192 rays x 128 fixed samples) with a non-zero learning rate;
the fixed-point grid gradient must never raise its overflow flag, the depth loss must fall."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops, synthetic, tcnn
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
torch.manual_seed(0)
scene = NeRFScene(dtype='bf16')
rays = gen_pano_rays(torch.eye(4), 1024, 2048)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
scene.set_train()
scene.estimator.set_binaries(torch.ones(256 ** 3, dtype=torch.uint8, device='cuda'))
r = scene.renderer
r.render_step_size = 0.99 / 128; r.far_plane = 10.0; r.early_stop_eps = 0.0; r.max_steps = 128; r.sample_capacity = 8192 * 128
scene.nerf.reset_geo()
tc = scene.train_conf
opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
N = 3000
for i in range(N):
    scene.update_lr(opt, tc.geo_optimizer, i / N)
    scene.train_one_step_geo(opt, pool, progress=i / 1500)
    if i in (10, 100, 1000, N - 1):
        print(i, 'depth loss', float(scene.last_losses['depth_loss']), 'dist loss', float(scene.last_losses['dist_loss']),
              'mode', tcnn.GRID_GRAD_ACCUM, 'flag', int(ops.overflow_flag('cuda').item()), flush=True)
opt = scene.make_optimizer(scene.nerf.app_mlp, 0.0)
for i in range(1000):
    scene.update_lr(opt, tc.app_optimizer, i / 1000)
    scene.train_one_step_app(opt, pool, progress=i / 1000)
print('app', 'color loss', float(scene.last_losses['color_loss']), 'mode', tcnn.GRID_GRAD_ACCUM, 'flag', int(ops.overflow_flag('cuda').item()))
