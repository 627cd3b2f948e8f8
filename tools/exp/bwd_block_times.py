"""Dev tool: per-workgroup duration of hashgrid_bwd by level (PERF_BWD_DEBUG=1)."""
import os, sys, ctypes
os.environ['PERF_BWD_DEBUG'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops, _lib
from perf_amd.grid import GridConfig
cfg = GridConfig(); dev = 'cuda'; n = 1 << 20
R = n // 128
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
t = (torch.arange(128, device=dev) + 0.5) / 128
x = ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5).contiguous()
dfeat = torch.randn(16, n, 2, device=dev)
amax = torch.zeros(24, device=dev); amax[:16] = dfeat.abs().amax(dim=(1, 2))
desc = cfg.desc()
need = _lib.load().perf_hashgrid_bwd_workspace_bytes(ctypes.byref(desc), n)
ws = torch.zeros(need // 4 + 64, dtype=torch.float32, device=dev)
out = torch.empty(cfg.n_params, device=dev)
for fixed in (False, True):
    for _ in range(2):
        ops._call('perf_hashgrid_bwd', ctypes.byref(desc), ops._p(x), ops._p(dfeat), ops._p(out), n, None, 0, ops._p(amax) if fixed else None,
                  None, None, ops._p(ws), ws.numel() * 4, ops._stream())
    torch.cuda.synchronize()
    need0 = _lib.load().perf_hashgrid_bwd_workspace_bytes(ctypes.byref(desc), 0)
    off = ((need0 - 16 - 4096 * 8 + 15) // 16 * 16) // 8          # debug slots follow the replica slabs
    cyc = ws[2 * off:2 * off + 1200].view(torch.int64).cpu().numpy()
    tiles = [max(1, -(-int(s) // 16384)) for s in cfg.size]
    tiles = [t_ if cfg.hashed[l] else 1 << (t_ - 1).bit_length() for l, t_ in enumerate(tiles)]
    if fixed:
        reps = [1 if cfg.hashed[l] else (8 if t_ == 1 else 3 if t_ <= 4 else 2 if t_ <= 16 else 1) for l, t_ in enumerate(tiles)]
    else:
        reps = [1 if cfg.hashed[l] else max(1, 16 // t_) for l, t_ in enumerate(tiles)]
    b = 0
    print('fixed' if fixed else 'fp32', '(wall_clock64 ticks @100MHz -> us = ticks/100)')
    for l in range(16):
        nb = tiles[l] * reps[l]
        c = cyc[b:b + nb]; b += nb
        print(f'  level {l:2d} tiles {tiles[l]:2d} x rep {reps[l]:2d}: mean {c.mean() / 100:8.1f} us  max {c.max() / 100:8.1f} us')
