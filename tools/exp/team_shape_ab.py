#!/usr/bin/env python
"""Ray-team kernels (composite.hip:for_rays_of_wave / team_grid) on rays of 128 and of 2 samples at 32,768 rays (the eval render's batches), just
below / at 262,144 rays and at 524,288 (sixteen rays per wave from there on).  ms per launch, HIP events."""
import sys, json
sys.path.insert(0, '.')
import torch
from perf_amd import ops

def case(R, spp, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    counts = torch.full((R,), spp, dtype=torch.int64, device='cuda')
    starts = torch.cumsum(counts, 0) - counts
    packed = torch.stack([starts, counts], -1).to(torch.int32)
    S = R * spp
    ts = torch.rand(S, device='cuda', generator=g) * 1.5; te = ts + 5e-3
    sig = torch.exp(torch.randn(S, device='cuda', generator=g) * 2 - 1)
    rgb = torch.rand(S, 3, device='cuda', generator=g)
    return packed, ts, te, sig, rgb

out = {}
for spp in (128, 2):
    for R in (32768, 262143, 262144, 524288):
        packed, ts, te, sig, rgb = case(R, spp)
        for _ in range(3):
            nc = ops.visibility_count(sig, ts, te, packed, 1e-4); ops.composite_fwd(sig, rgb, ts, te, packed); ops.compact_prefix(packed, nc, ts, te, sig, capacity=R * spp)
        torch.cuda.synchronize()
        ops.start_kernel_timing()
        for _ in range(20):
            nc = ops.visibility_count(sig, ts, te, packed, 1e-4); ops.composite_fwd(sig, rgb, ts, te, packed); ops.compact_prefix(packed, nc, ts, te, sig, capacity=R * spp)
        k = ops.stop_kernel_timing()
        out[f'{spp} samples/ray, {R} rays'] = {n: round(ms, 4) for n, (c, ms) in k.items()}
        del packed, ts, te, sig, rgb
print(json.dumps(out, indent=1))
