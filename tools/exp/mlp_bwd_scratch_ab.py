#!/usr/bin/env python
"""Times perf_mlp_bwd for the template variants that carry scratch at two workgroups per CU (NH = 2 with KS = 1, and NH = 2, KS = 2
without FAST: grids of 8, < 8 or 9-15 levels -- none of them PeRF's L = 16): 1 M samples, HIP events, per variant."""
import json, sys, torch
sys.path.insert(0, '.')
from perf_amd import ops
from perf_amd.grid import MlpConfig
n = 1 << 20
out = {}
for name, L in (('NH2 KS2 FAST (L=16, the shipped hot path)', 16), ('NH2 KS2 plain (L=12)', 12), ('NH2 KS1 FAST (L=8)', 8), ('NH2 KS1 plain (L=6)', 6)):
    cfg = MlpConfig(n_levels=L, n_hidden_layers=2, n_output_dims=3, output_activation='Sigmoid')
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(cfg.n_params, generator=g) * 0.1).bfloat16().cuda()
    f = torch.randn(L, n, 2, generator=g).bfloat16().cuda()
    dout = torch.randn(n, 3, generator=g).cuda()
    for _ in range(3):
        ops.mlp_bwd(cfg, w, f, dout)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(20):
        ops.mlp_bwd(cfg, w, f, dout)
    ev[1].record(); torch.cuda.synchronize()
    out[name] = round(ev[0].elapsed_time(ev[1]) / 20 * 1e3, 1)
print(json.dumps(out, indent=1))
