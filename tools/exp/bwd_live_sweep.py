"""perf_hashgrid_bwd (fixed-point owners, capacity 1 M rows) against the LIVE sample count: what the reference-faithful step (16-35 k live
samples) pays in fixed cost.  Run under rocprofv3 --kernel-trace (profiles/r05_gpurun_calls.md, call 15, folds the trace by group)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig
cfg = GridConfig()
g = torch.Generator(device='cuda').manual_seed(1)
R, S = 8192, 128
d = torch.nn.functional.normalize(torch.randn(R, 3, device='cuda', generator=g), dim=-1)
t = (torch.arange(S, device='cuda') + torch.rand(R, 1, device='cuda', generator=g)) * (0.99 / S)
x = ((d[:, None, :] * t[:, :, None]).reshape(-1, 3) * 0.5 + 0.5).contiguous()
n = x.shape[0]
G = torch.randn(cfg.n_levels, n, 2, device='cuda', generator=g) * 1e-3
amax = torch.zeros(24, device='cuda'); amax[:cfg.n_levels] = G.abs().amax(dim=(1, 2))
hr = ops.headroom_state('cuda')
out = torch.empty(cfg.n_params, device='cuda')
for live in (0, 1024, 16384, 65536, 262144, n):
    n_dev = torch.tensor([live], dtype=torch.int64, device='cuda')
    for _ in range(12):
        ops.hashgrid_bwd(cfg, x, G, out=out, level_absmax=amax, n_dev=n_dev, hr_state=hr)
    torch.cuda.synchronize()
print('done')
