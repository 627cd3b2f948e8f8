"""How much head-room do the fixed-point grid-gradient fields need?  Per level: max |entry sum| / max |dfeat| and the
average fan-in, along a short faithful-mode training run."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops, synthetic, tcnn
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
torch.manual_seed(0)
scene = NeRFScene(dtype='bf16')
rays = gen_pano_rays(torch.eye(4), 128, 256)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
tc = scene.train_conf; tc.pixel_loss_batch_size = 2048
scene.set_train(); scene.prepare_occupancy(pool); scene.nerf.reset_geo()
stats = {}
orig = ops.hashgrid_bwd_into
def probe(grid, x01, dfeat, out, level_absmax=None):
    if level_absmax is not None:                                 # what does the fixed-point path say?
        ops.overflow_flag(x01.device).zero_()
        tmp = torch.empty_like(out)
        orig(grid, x01, dfeat, tmp, level_absmax=level_absmax)
        if int(ops.overflow_flag(x01.device).item()):
            true = dfeat.abs().amax(dim=(1, 2))
            print('FLAG at n =', x01.shape[0], 'given amax', [round(float(v), 6) for v in level_absmax[:16]], 'true', [round(float(v), 6) for v in true])
            ops.overflow_flag(x01.device).zero_()
    r = orig(grid, x01, dfeat, out, level_absmax=None)          # fp32 accumulation: the true sums
    n = x01.shape[0]
    for l in range(grid.n_levels):
        lo, hi = 2 * int(grid.offset[l]), 2 * (int(grid.offset[l]) + int(grid.size[l]))
        a = float(dfeat[l].abs().max()); m = float(out[lo:hi].abs().max())
        if a > 0:
            need = math.log2(max(m / a, 1e-9)); fan = 8.0 * n / int(grid.size[l])
            k = stats.setdefault(l, [0.0, 0.0, n]); k[0] = max(k[0], need); k[1] = max(k[1], need - math.log2(max(fan, 1.0)))
    return r
ops.hashgrid_bwd_into = probe
import perf_amd.scene as S
S.ops.hashgrid_bwd_into = probe
opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
for i in range(150):
    scene.update_lr(opt, tc.geo_optimizer, i / 150)
    scene.train_one_step_geo(opt, pool, progress=i / 100)
print('geo phase: level, log2(max|sum|/amax), same minus log2(avg fan-in), n')
for l, v in sorted(stats.items()): print(l, round(v[0], 2), round(v[1], 2), v[2])
stats.clear()
opt = scene.make_optimizer(scene.nerf.app_mlp, 0.0)
for i in range(100):
    scene.update_lr(opt, tc.app_optimizer, i / 100)
    scene.train_one_step_app(opt, pool, progress=i / 100)
print('app phase')
for l, v in sorted(stats.items()): print(l, round(v[0], 2), round(v[1], 2), v[2])
