"""Dev library of tools/exp/bwd_sort*.py: the grid backward's owner variants (PERF_BWD_SORT / PERF_BWD_RUNS) -- equality of the fixed-point gradients,
ms per call, per-workgroup times by level.    python tools/exp/bwd_sort.py [out.json]"""
import os, sys, ctypes, json
os.environ['PERF_BWD_DEBUG'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops, _lib
from perf_amd.grid import GridConfig

cfg = GridConfig(); dev = 'cuda'
desc = cfg.desc()
lib = _lib.load()


def batch(kind, n, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    if kind == 'rays':
        R = n // 128
        d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=-1)
        t = (torch.arange(128, device=dev) + 0.5) / 128
        x = ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5).contiguous()
    else:
        x = torch.rand(n, 3, device=dev, generator=g)
    dfeat = torch.randn(16, n, 2, device=dev, generator=g) * 1e-3
    amax = torch.zeros(24, device=dev); amax[:16] = dfeat.abs().amax(dim=(1, 2))
    return x, dfeat, amax


def setenv(variant, runs):
    # variant: PERF_BWD_SORT (commit 'counting-sorted per-tile records') / PERF_BWD_BITMAP (commit 'per-tile bitmaps'); the
    # shipped library only knows PERF_BWD_RUNS
    os.environ['PERF_BWD_BITMAP'] = str(variant); os.environ['PERF_BWD_SORT'] = str(variant); os.environ['PERF_BWD_RUNS'] = str(runs)


def call(x, dfeat, amax, n_dev=None, fixed=True, ws=None, out=None):
    n = x.shape[0]
    need = lib.perf_hashgrid_bwd_workspace_bytes(ctypes.byref(desc), n)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.zeros(need // 4 + 64, dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty(cfg.n_params, device=dev)
    ops._call('perf_hashgrid_bwd', ctypes.byref(desc), ops._p(x), ops._p(dfeat), ops._p(out), n, ops._nd(n_dev), 0,
              ops._p(amax) if fixed else None, None, None, None, 0, ops._p(ws), ws.numel() * 4, ops._stream())
    return out, ws


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def block_times(ws, n, fast):
    need0 = lib.perf_hashgrid_bwd_workspace_bytes(ctypes.byref(desc), 0)
    off = (need0 - 16 - 4096 * 8) // 8
    cyc = ws[2 * off:2 * off + 1200].view(torch.int64).cpu().numpy()
    tiles = [max(1, -(-int(s) // 16384)) for s in cfg.size]
    tiles = [t_ if cfg.hashed[l] else 1 << (t_ - 1).bit_length() for l, t_ in enumerate(tiles)]
    rs = [int(v) for v in os.environ.get('PERF_BWD_REPLICAS', '8,3,2').split(',')]
    reps = [1 if cfg.hashed[l] else (rs[0] if t_ == 1 else rs[1] if t_ <= 4 else rs[2] if t_ <= 16 else 1) for l, t_ in enumerate(tiles)]
    b = 0; rows = []
    for l in range(16):
        nb = tiles[l] * reps[l]
        c = cyc[b:b + nb]; b += nb
        rows.append((l, tiles[l], reps[l], float(c.mean()) / 100, float(c.max()) / 100))
    return rows


