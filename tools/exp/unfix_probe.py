"""perf_fixed_unfix alone on the benchmark's table (3.3 M entries): HIP-event time per call for dense and for sparse integer fields."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from perf_amd import ops  # noqa: E402
from perf_amd.scene import NeRFScene  # noqa: E402

scene = NeRFScene(dtype='bf16')
grid = scene.nerf.geo_mlp.grid
n = grid.n_params // 2
shifts = torch.full((24,), 10, dtype=torch.int32, device='cuda')
fm = torch.zeros(24, dtype=torch.int32, device='cuda')
for name, fill in (('dense', lambda: torch.randint(-1000, 1000, (n, 2), dtype=torch.int32, device='cuda')),
                   ('zeros', lambda: torch.zeros(n, 2, dtype=torch.int32, device='cuda'))):
    ts = []
    for _ in range(12):
        buf = fill()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.fixed_unfix(grid, buf, 0, n, shifts, fm); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(name, n, 'entries', sorted(ts)[len(ts) // 2], 'us (median of 12, incl. the memset of the maxima and launch overhead)')
