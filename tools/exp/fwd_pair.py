"""Dev tool: A/B of the forward encode's lane mapping.  [PERF_FWD_PAIR=1] python tools/exp/fwd_pair.py
Prints the time per launch on training-like (128 samples along random rays), random and eval-like point sets, and a
checksum of the features (the two variants must agree bit for bit)."""
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


res = {'variant': 'pair' if os.environ.get('PERF_FWD_PAIR') else 'wide' if os.environ.get('PERF_FWD_WIDE') else 'base'}
n = 1 << 20
only = sys.argv[1] if len(sys.argv) > 1 else None
sets = {'train128': rays_points(n), 'random': torch.rand(n, 3, device='cuda'), 'head4': rays_points(n, 4)}
for name, x in sets.items():
    if only and name != only:
        continue
    f = ops.hashgrid_fwd(cfg, x, table)
    res[name + '_ms'] = round(timeit(lambda: ops.hashgrid_fwd(cfg, x, table)), 4)
    res[name + '_sum'] = int(f.view(torch.int16).to(torch.int64).sum().item())
if only:
    print(json.dumps(res)); sys.exit(0)
nd = torch.tensor([600000], dtype=torch.int64, device='cuda')
res['cap1M_live600k_ms'] = round(timeit(lambda: ops.hashgrid_fwd(cfg, sets['train128'], table, n_dev=nd)), 4)
x = sets['train128'][:1000003].contiguous()
res['odd_n_sum'] = int(ops.hashgrid_fwd(cfg, x, table).view(torch.int16).to(torch.int64).sum().item())
print(json.dumps(res))
