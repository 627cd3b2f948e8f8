"""Dev tool: how many 64-interval chunks / 16-interval sub-chunks of a config-4 ray survive the skip grids (perf_amd/csrc/march.hip:
dilated 4^3-block grid at the chunk midpoint, dilated 2^3-block grid at two / four lattice points)?  Decides whether packing
sub-chunks four to a pass would pay.   python tools/exp/march_live_stats.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops, synthetic
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays

torch.manual_seed(0)
scene = NeRFScene(dtype='fp16')
rays = gen_pano_rays(torch.eye(4), 512, 1024)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
scene.set_train(); scene.prepare_occupancy(pool)
est = scene.estimator
res = 256
occ = est.binaries.reshape(res, res, res)
def dilated(block):
    n = res // block
    b = occ.reshape(n, block, n, block, n, block).any(5).any(3).any(1).float()[None, None]
    return (torch.nn.functional.max_pool3d(b, 3, 1, 1)[0, 0] > 0)
coarse, fine = dilated(4), dilated(2)
pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.15, -0.1, 0.05])
r = gen_pano_rays(pose, 512, 1024)
o, d = r.o.reshape(-1, 3), r.d.reshape(-1, 3)
step, n_steps = 5e-4, 3001
k = torch.arange(0, 47 * 64 + 65, device='cuda', dtype=torch.float32)
t = k * step                                                        # (statistics only: the single-rounding lattice)
def block_at(tt, grid, block):                                     # tt [C] -> alive [R, C]
    p = o[:, None, :] + d[:, None, :] * tt[None, :, None]
    u = ((p + 1.0) * 0.5 * res).floor().clamp(0, res - 1).long() // block
    return grid[u[..., 0], u[..., 1], u[..., 2]]
q = torch.arange(47, device='cuda')
out = {}
with torch.no_grad():
    live_c = torch.zeros(o.shape[0], 47, dtype=torch.bool, device='cuda')
    live_f2 = torch.zeros_like(live_c)
    sub = torch.zeros(o.shape[0], 47, 4, dtype=torch.bool, device='cuda')
    for lo in range(0, o.shape[0], 65536):
        sl = slice(lo, lo + 65536)
        oo, dd = o, d
        o_, d_ = o[sl], d[sl]
        def blk(tt, grid, block):
            p = o_[:, None, :] + d_[:, None, :] * tt[None, :, None]
            u = ((p + 1.0) * 0.5 * res).floor().clamp(0, res - 1).long() // block
            return grid[u[..., 0], u[..., 1], u[..., 2]]
        c = blk(t[q * 64 + 32], coarse, 4)
        f2 = blk(t[q * 64 + 16], fine, 2) | blk(t[q * 64 + 48], fine, 2)
        live_c[sl] = c; live_f2[sl] = c & f2
        for j in range(4):
            sub[sl, :, j] = c & blk(t[q * 64 + 8 + 16 * j], fine, 2)
    n_c = live_c.sum(1).float(); n_f2 = live_f2.sum(1).float(); n_sub = sub.sum((1, 2)).float()
    passes_sub = torch.ceil(n_sub / 4)
    out = {'rays': int(o.shape[0]), 'chunks_alive_coarse_only': float(n_c.mean()), 'chunks_alive_coarse_and_fine_2pt (shipped)': float(n_f2.mean()),
           'sub_chunks_alive_4pt': float(n_sub.mean()), 'passes_if_packed_4_sub_chunks': float(passes_sub.mean()),
           'ratio_packed_over_shipped': float(passes_sub.mean() / n_f2.mean())}
print(json.dumps(out, indent=1))
