"""Config-5 style stress: the encode with tables far larger than the 256 MiB Infinity Cache (HBM-bound gathers)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig
from tools.microbench import timeit

dev = 'cuda'
n = 1 << 22
R = n // 128
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
t = (torch.arange(128, device=dev) + 0.5) / 128
xr = ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5).contiguous()
xu = torch.rand(n, 3, device=dev)
out = {}
for T in (18, 20, 22, 24, 26):
    cfg = GridConfig(log2_hashmap_size=T, per_level_scale=1.4472692012786865)
    table = (torch.rand(cfg.n_params, device=dev) * 2 - 1).to(torch.bfloat16)
    mb = cfg.n_params * 2 / 2 ** 20
    for name, x in (('ray', xr), ('uniform', xu)):
        tf = timeit(lambda: ops.hashgrid_fwd(cfg, x, table), iters=5, warm=2)
        sps = n / tf
        out[f'T{T}/{name}'] = {'table_MiB': round(mb, 1), 'Gsamples_per_s': round(sps / 1e9, 3),
                               'algorithmic_GBs': round(sps * 512 / 1e9, 1), 'frac_of_8TBs': round(sps * 512 / 8e12, 4)}
        print(f'T={T} table {mb:8.1f} MiB {name:8s} {sps / 1e9:7.3f} Gsamples/s  algorithmic {sps * 512 / 1e9:8.1f} GB/s')
    del table
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/bigtable.json', 'w'), indent=1)
