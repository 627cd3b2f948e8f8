"""Dev tool: do the L1-bound forward encode and the VALU-bound grid backward overlap when issued on two streams?
Times bwd alone, fwd alone, both back to back on one stream, and both on two streams (eager and inside one hipGraph)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig

cfg = GridConfig()
torch.manual_seed(0)
R, S = 8192, 128
n = R * S
d = torch.nn.functional.normalize(torch.randn(R, 3, device='cuda'), dim=-1)
t = ((torch.arange(S, device='cuda') + torch.rand(R, 1, device='cuda')) / S * 0.99)
x = ((d[:, None, :] * t[:, :, None]).reshape(-1, 3) * 0.5 + 0.5).contiguous()
dfeat = (torch.randn(16, n, 2, device='cuda') * torch.exp(torch.randn(1, n, 1, device='cuda'))).contiguous()
amax = torch.zeros(24, device='cuda'); amax[:16] = dfeat.abs().amax(dim=(1, 2))
out = torch.empty(cfg.n_params, device='cuda')
st = ops.headroom_state('cuda')
table = (torch.rand(cfg.n_params, device='cuda') * 2e-4 - 1e-4).to(torch.bfloat16)
side = torch.cuda.Stream()


def bwd():
    ops.hashgrid_bwd(cfg, x, dfeat, out=out, level_absmax=amax, hr_state=st)


def fwd():
    return ops.hashgrid_fwd(cfg, x, table)


def serial():
    bwd(); fwd()


def forked(first_side=True):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        f = fwd()
    bwd()
    cur.wait_stream(side)
    return f


def forked_bwd_first():
    cur = torch.cuda.current_stream()
    bwd_started = torch.cuda.Event()
    side.wait_stream(cur)
    bwd()
    with torch.cuda.stream(side):
        f = fwd()
    cur.wait_stream(side)
    return f


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 4)


res = {'bwd_ms': timeit(bwd), 'fwd_ms': timeit(fwd), 'serial_ms': timeit(serial), 'two_streams_ms': timeit(forked),
       'two_streams_bwd_first_ms': timeit(forked_bwd_first)}
for name, fn in (('graph_serial_ms', serial), ('graph_forked_ms', forked), ('graph_forked_bwd_first_ms', forked_bwd_first)):
    g = torch.cuda.CUDAGraph()
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        fn()
    torch.cuda.current_stream().wait_stream(s2)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        keep = fn()
    res[name] = timeit(g.replay)
print(json.dumps(res))
