"""Driver of tools/exp/mlp_pmc.sh: the geometry network's mlp_fwd / mlp_bwd (one hidden layer, 16 levels, 16 outputs) and the colour
network's (two hidden layers, 3 outputs) on 2^20 samples, a few launches each -- what rocprofv3 counts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from perf_amd import ops  # noqa: E402
from perf_amd.grid import MlpConfig  # noqa: E402

n = 1 << 20
g = torch.Generator().manual_seed(0)
for nh, n_out, act in ((1, 16, 'None'), (2, 3, 'Sigmoid')):
    cfg = MlpConfig(n_levels=16, n_hidden_layers=nh, n_output_dims=n_out, output_activation=act, exp_shift=0.0)
    w = torch.cat([(torch.rand(o * i, generator=g) * 2 - 1) * (6.0 / (i + o)) ** 0.5 for o, i in cfg.shapes]).to(torch.bfloat16).cuda()
    feat = (torch.rand(16, n, 2, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    dout = torch.randn(n, n_out, generator=g).cuda()
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        ops.mlp_fwd(cfg, w, feat, None)
        ops.mlp_bwd(cfg, w, feat, dout, None, want_absmax=True)
torch.cuda.synchronize()
