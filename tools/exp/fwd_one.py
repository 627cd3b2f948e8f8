#!/usr/bin/env python
"""One forward-encode variant (selected by the PERF_FWD_* switches of the environment), 24 launches on the training-batch
distribution of tools/exp/fwd_v2.py -- the workload of the L1 counter passes (tools/exp/r03_profile.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig
from tools.exp.fwd_v2 import samples

dev = torch.device('cuda', 0)
cfg = GridConfig()
g = torch.Generator().manual_seed(1)
t16 = ops.cast_params(((torch.rand(cfg.n_params, generator=g) * 2 - 1) * 0.5).to(dev), 'bf16')
x = samples(sys.argv[1] if len(sys.argv) > 1 else 'train', dev)
for _ in range(24):
    ops.hashgrid_fwd(cfg, x, t16)
torch.cuda.synchronize()
