"""Dev tool: hashgrid_bwd (fixed point) on a bench-like batch: 8192 rays x 128 samples from the origin, gradients of a
plausible dynamic range.  PERF_BWD_NO_LISTS=1 selects the byte-code owners for the hashed levels (A/B).
Run under `rocprofv3 --kernel-trace --stats --output-format csv` to split pre-pass / owners / reduce."""
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


def run():
    ops.hashgrid_bwd(cfg, x, dfeat, out=out, level_absmax=amax, hr_state=st)


for _ in range(10):
    run()
ref = out.clone()
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30):
    run()
b.record(); torch.cuda.synchronize()
print(json.dumps({'PERF_BWD_NO_LISTS': os.environ.get('PERF_BWD_NO_LISTS', ''), 'ms_per_call': a.elapsed_time(b) / 30,
                  'checksum': float(ref.double().abs().sum())}))
