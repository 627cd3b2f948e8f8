"""Dev tool: per-workgroup duration of the grid backward by level on a config-5 style grid (PERF_BWD_DEBUG=1; the first
4096 workgroups have debug slots).    python tools/exp/bwd_level_times.py [--levels 20] [--log2 22] [--kind random|rays]"""
import argparse, os, sys, ctypes
os.environ['PERF_BWD_DEBUG'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops, _lib
from perf_amd.grid import GridConfig

ap = argparse.ArgumentParser()
ap.add_argument('--levels', type=int, default=20)
ap.add_argument('--log2', type=int, default=22)
ap.add_argument('--kind', default='random')
args = ap.parse_args()
L, T = args.levels, args.log2
b = float(torch.exp(torch.log(torch.tensor(8192.0 / 16)) / (L - 1)))
cfg = GridConfig(n_levels=L, log2_hashmap_size=T, base_resolution=16, per_level_scale=b)
n = 1 << 20
g = torch.Generator(device='cuda').manual_seed(1)
if args.kind == 'rays':
    d = torch.nn.functional.normalize(torch.randn(n // 128, 3, device='cuda', generator=g), dim=-1)
    t = (torch.arange(128, device='cuda') + 0.5) / 128
    x = ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5).contiguous()
else:
    x = torch.rand(n, 3, device='cuda', generator=g)
dfeat = torch.randn(L, n, 2, device='cuda', generator=g) * 1e-3
amax = torch.zeros(24, device='cuda'); amax[:L] = dfeat.abs().amax(dim=(1, 2))
d = cfg.desc()
lib = _lib.load()
need = lib.perf_hashgrid_bwd_workspace_bytes(ctypes.byref(d), n)
ws = torch.zeros(need // 4 + 64, dtype=torch.float32, device='cuda')
out = torch.empty(cfg.n_params, device='cuda')
for _ in range(2):
    ops._call('perf_hashgrid_bwd', ctypes.byref(d), ops._p(x), ops._p(dfeat), ops._p(out), n, None, 0, ops._p(amax), None, None, None, 0, ops._p(ws), ws.numel() * 4, ops._stream())
torch.cuda.synchronize()
bitmap = os.environ.get('PERF_BWD_BITMAP', '1') != '0'
rs = [int(v) for v in os.environ.get('PERF_BWD_REPLICAS', '8,3,2').split(',')]


def tiles_of(l):
    nt = max(1, -(-int(cfg.size[l]) // 16384))
    return nt if cfg.hashed[l] else 1 << (nt - 1).bit_length()


def is_bitmap(l, on):
    nt = tiles_of(l)
    return on and ((cfg.hashed[l] and 255 < nt <= 2048) or (not cfg.hashed[l] and 32 <= nt <= 2048))


def replicas(l, fixed, on):            # plan_tiles' rule (perf_amd/csrc/hashgrid.hip)
    nt = tiles_of(l)
    if cfg.hashed[l] or is_bitmap(l, on) or nt > 64:
        return 1
    if not fixed:
        return max(1, 16 // nt)
    if on and any(is_bitmap(k, on) for k in range(L)) and 'PERF_BWD_REPLICAS' not in os.environ:
        return {1: 8, 2: 8, 4: 6, 8: 4}.get(nt, 2)
    return rs[0] if nt == 1 else rs[1] if nt <= 4 else rs[2] if nt <= 16 else 1


def slab_entries(fixed, on):
    return sum(replicas(l, fixed, on) * int(cfg.size[l]) for l in range(L) if replicas(l, fixed, on) > 1)


# the debug slots follow the replica slabs (the largest of the plans a call may pick) and the 256 bytes of shifts
slab = max(slab_entries(True, False), slab_entries(False, False), slab_entries(True, bitmap))
off = (((slab * 8 + 15) // 16 * 16) + 256) // 8
cyc = ws[2 * off:2 * off + 8192].view(torch.int64).cpu().numpy()
pos = 0
for l in range(L):
    nt = tiles_of(l)
    is_bm = is_bitmap(l, bitmap)
    if nt > (255 if cfg.hashed[l] else 64) and not is_bm:
        print(f'  level {l:2d} res {int(cfg.res[l]):5d} tiles {nt:5d}: global atomics'); continue
    r = replicas(l, True, bitmap)
    nb = nt * r
    if pos + nb > 4096:
        print(f'  level {l:2d}: beyond the debug slots'); break
    c = cyc[pos:pos + nb]; pos += nb
    print(f'  level {l:2d} res {int(cfg.res[l]):5d} {"hashed" if cfg.hashed[l] else "dense "} tiles {nt:5d} x rep {r}: mean {c.mean() / 100:8.1f} us  max {c.max() / 100:8.1f} us' + ('  [bitmap]' if is_bm else ''))
