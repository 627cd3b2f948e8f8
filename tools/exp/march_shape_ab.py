#!/usr/bin/env python
"""The marching kernels of large launches on one lattice (march_count_shared_kernel / march_write_shared_kernel from 262,144 rays on) against
the wave-per-ray / quarter-wave kernels just below that size, on HEAVY rays (every interval occupied: ~128 samples per ray, the eval
render's batches) and on LIGHT ones (a thin shell: a few samples per ray).  ms per launch, HIP events."""
import sys, json
sys.path.insert(0, '.')
import torch
from perf_amd import ops

res, step, far = 128, 0.0117, 1.5
aabb = [-1., -1, -1, 1, 1, 1]
out = {}
for name, fill in (('heavy (all cells occupied)', 'all'), ('light (a thin shell)', 'shell')):
    occ = torch.zeros(res, res, res, dtype=torch.bool, device='cuda')
    if fill == 'all':
        occ[:] = True
    else:
        c = (torch.stack(torch.meshgrid(*[torch.arange(res, device='cuda')] * 3, indexing='ij'), -1).float() + 0.5) / res * 2 - 1
        rr = c.norm(dim=-1)
        occ = (rr > 0.80) & (rr < 0.83)
    bits = ops.occ_pack_bits(occ.reshape(-1).to(torch.uint8))
    coarse = ops.occ_build_coarse(bits, res)
    max_steps = 128
    for R in (262143, 262144):
        g = torch.Generator(device='cuda').manual_seed(3)
        d = torch.nn.functional.normalize(torch.randn(R, 3, device='cuda', generator=g), dim=-1)
        o = torch.zeros(R, 3, device='cuda')
        t0 = (None, 0.0, 0.0, ops.lattice_table(0.0, step, max_steps, 'repeated'))
        for _ in range(2):
            res_ = ops.occ_march(o, d, t0, bits, res, aabb, far, step, max_steps, capacity=R * max_steps, occ_coarse=coarse, points_aabb=aabb, lattice='repeated')
        torch.cuda.synchronize()
        ops.start_kernel_timing()
        for _ in range(10):
            res_ = ops.occ_march(o, d, t0, bits, res, aabb, far, step, max_steps, capacity=R * max_steps, occ_coarse=coarse, points_aabb=aabb, lattice='repeated')
        k = ops.stop_kernel_timing()
        out[f'{name}, {R} rays'] = {'samples_per_ray': round(float(res_[4].item()) / R, 1), **{n: round(ms, 4) for n, (c, ms) in k.items()}}
print(json.dumps(out, indent=1))
