"""Dev tool: forward encode with the levels dealt fractionally over the XCDs.  [PERF_FWD_DEAL=<first level>] python tools/exp/fwd_deal.py
PERF_FWD_DEAL=0: all 16 levels dealt (bit-identical to the shipped kernel); =3: levels 3..15 only (what is left for the L1
when levels 0-2 are served from LDS)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig

cfg = GridConfig()
torch.manual_seed(0)
table = (torch.rand(cfg.n_params, device='cuda') * 2e-4 - 1e-4).to(torch.bfloat16)


def rays_points(n, spp=128):
    R = n // spp + 1
    d = torch.nn.functional.normalize(torch.randn(R, 3, device='cuda'), dim=-1)
    t = (torch.arange(spp, device='cuda') + 0.5) / spp * 0.99
    return ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5)[:n].contiguous()


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


first = int(os.environ.get('PERF_FWD_DEAL', '-1'))
res = {'PERF_FWD_DEAL': first}
for name, x in (('train128', rays_points(1 << 20)), ('random', torch.rand(1 << 20, 3, device='cuda')), ('train128_4M', rays_points(1 << 22))):
    f = ops.hashgrid_fwd(cfg, x, table)
    res[name + '_ms'] = round(timeit(lambda: ops.hashgrid_fwd(cfg, x, table)), 4)
    res[name + '_sum'] = int(f[max(first, 0):].view(torch.int16).to(torch.int64).sum().item())
print(json.dumps(res))
