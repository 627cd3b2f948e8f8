"""Dev tool: hashgrid_fwd launch shapes.  PERF_FWD_MAX_CHUNKS=<k> python tools/exp/fwd_chunks.py
Times the encode for (capacity, live) pairs: exact launches and capacity-sized launches with a device-side count."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig

cfg = GridConfig()
torch.manual_seed(0)
table = (torch.rand(cfg.n_params, device='cuda') * 2e-4 - 1e-4).to(torch.bfloat16)


def rays_points(n):
    R = n // 128 + 1
    d = torch.nn.functional.normalize(torch.randn(R, 3, device='cuda'), dim=-1)
    t = (torch.arange(128, device='cuda') + 0.5) / 128 * 0.99
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


res = {'PERF_FWD_MAX_CHUNKS': os.environ.get('PERF_FWD_MAX_CHUNKS', 'default')}
for cap, live in ((1 << 20, 1 << 20), (1 << 20, 600000), (1 << 21, 500000), (1 << 21, 100000), (1 << 21, 0), (32768 * 64, 32768 * 20)):
    x = rays_points(cap)
    nd = torch.tensor([live], dtype=torch.int64, device='cuda')
    res[f'cap{cap}_live{live}_ms'] = round(timeit(lambda: ops.hashgrid_fwd(cfg, x, table, n_dev=nd)), 4)
    if live:
        xe = x[:live].contiguous()
        res[f'exact{live}_ms'] = round(timeit(lambda: ops.hashgrid_fwd(cfg, xe, table)), 4)
counts = torch.randint(0, 60, (32768,), dtype=torch.int32, device='cuda')
res['scan_32768_ms'] = round(timeit(lambda: ops.exclusive_scan_i32(counts)), 4)
c2 = torch.randint(0, 60, (8192,), dtype=torch.int32, device='cuda')
res['scan_8192_ms'] = round(timeit(lambda: ops.exclusive_scan_i32(c2)), 4)
c3 = torch.randint(0, 60, (262144,), dtype=torch.int32, device='cuda')
res['scan_262144_ms'] = round(timeit(lambda: ops.exclusive_scan_i32(c3)), 4)
print(json.dumps(res))
