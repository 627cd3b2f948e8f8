"""Dev probe (VERDICT-3 item 5a): what the fp32 feature-gradient write costs perf_mlp_bwd.  The same launch with and without
the dfeat output (need_dfeat=False skips the dX product AND its 137 MB store): an upper bound for what a packed 16-bit dfeat
(-69 MB of the store, the product stays) could save."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig, MlpConfig

cfg = GridConfig()
n = 8192 * 128
torch.manual_seed(0)
mlp = MlpConfig(n_levels=16, n_hidden_layers=1, n_output_dims=1, output_activation='Exponential')
w16 = (torch.randn(mlp.n_params, device='cuda') * 0.2).to(torch.bfloat16)
feat = (torch.randn(16, n, 2, device='cuda') * 0.3).to(torch.bfloat16)
dout = torch.randn(n, 1, device='cuda')
sel = torch.ones(n, dtype=torch.uint8, device='cuda')
res = {}
for tag, need in (('with_dfeat_fp32', True), ('without_dfeat', False)):
    for _ in range(5):
        ops.mlp_bwd(mlp, w16, feat, dout, sel, need_dfeat=need, want_absmax=need)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        ops.mlp_bwd(mlp, w16, feat, dout, sel, need_dfeat=need, want_absmax=need)
    b.record(); torch.cuda.synchronize()
    res[tag + '_ms'] = round(a.elapsed_time(b) / 30, 4)
res['note'] = ('1,048,576 samples, density net (32 -> 64 -> 1), bf16; the difference bounds what halving the dfeat store could save '
               '(the dX = W1^T dH1 MFMA product is skipped too without dfeat)')
print(json.dumps(res))
