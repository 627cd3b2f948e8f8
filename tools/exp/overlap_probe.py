"""Do the VALU-bound grid backward and the L2-bound encode overlap when issued on two streams?"""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig, MlpConfig
cfg = GridConfig(); dev = 'cuda'; n = 1 << 20
R = n // 128
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
t = (torch.arange(128, device=dev) + 0.5) / 128
x = ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5).contiguous()
dfeat = torch.randn(16, n, 2, device=dev)
amax = torch.zeros(24, device=dev); amax[:16] = dfeat.abs().amax(dim=(1, 2))
table = (torch.rand(cfg.n_params, device=dev) * 2 - 1).to(torch.bfloat16)
mlp = MlpConfig(16, 2, 3, 'Sigmoid')
w = (torch.randn(mlp.n_params, device=dev) * 0.2).to(torch.bfloat16)
sel = torch.ones(n, dtype=torch.uint8, device=dev)
out = torch.empty(cfg.n_params, device=dev)
side = torch.cuda.Stream()

def bwd():
    ops.hashgrid_bwd(cfg, x, dfeat, out=out, level_absmax=amax)

def fwd():
    f = ops.hashgrid_fwd(cfg, x, table)
    return ops.mlp_fwd(mlp, w, f, sel)

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3

def both():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        r = fwd()
    bwd()
    torch.cuda.current_stream().wait_stream(side)
    return r

def both_fwd_first():
    ev = torch.cuda.Event(); ev.record()
    bwd_stream = side
    with torch.cuda.stream(bwd_stream):
        bwd_stream.wait_event(ev)
        bwd()
    r = fwd()
    torch.cuda.current_stream().wait_stream(bwd_stream)
    return r

print('bwd alone      %.3f ms' % timeit(bwd))
print('fwd+mlp alone  %.3f ms' % timeit(fwd))
print('serial         %.3f ms' % timeit(lambda: (bwd(), fwd())))
print('two streams (bwd on main)  %.3f ms' % timeit(both))
print('two streams (fwd on main)  %.3f ms' % timeit(both_fwd_first))
