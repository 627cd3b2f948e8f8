"""Dev tool: how long every workgroup of hashgrid_bwd_kernel<fixed> takes, by level (the launch is as long as its slowest workgroup).

Builds a copy of the library with -DPERF_BWD_BLOCK_TIMES (hashgrid_bwd.hip: one thread of two workgroups per level and replica prints
its wall-clock duration), runs a few 1 M-sample calls on ray-shaped points and prints median / max / min per level.  Round 6 found the
launch ending with dense level 3 (8 tiles x 2 replicas: 357 us) while every change to the hashed owners (331-350 us) went unseen:
DESIGN.md 5.1.  `python tools/exp/bwd_block_times.py [n_samples]` on a GPU box; nothing in the tree is modified."""
import collections, ctypes, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from perf_amd import build as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
tmp = tempfile.mkdtemp(prefix='perf_blockt_')
B.build()
obj = os.path.join(tmp, 'hashgrid_bwd.o')
subprocess.run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + B.FLAGS + ['-DPERF_BWD_BLOCK_TIMES', '-c', os.path.join(B.CSRC, 'hashgrid_bwd.hip'), '-o', obj], check=True)
objs = [obj if s == 'hashgrid_bwd.hip' else os.path.join(B.HERE, 'build', s.replace('.hip', '.o')) for s in B.SOURCES]
lib = os.path.join(tmp, 'libperf_hip.so')
subprocess.run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs, check=True)

child = f'''
import sys; sys.path.insert(0, {ROOT!r})
from perf_amd import _lib
_lib.LIB_PATH = {lib!r}
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig
cfg = GridConfig(); n = {n}
R = n // 128
d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=-1)
t = (torch.arange(128, device="cuda") + 0.5) / 128
x = ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.45 + 0.5).contiguous()
dfeat = torch.randn(16, n, 2, device="cuda") * 1e-3
amax = torch.zeros(24, device="cuda"); amax[:16] = dfeat.abs().amax(dim=(1, 2))
hr = ops.headroom_state("cuda")
for _ in range(6):
    ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax, hr_state=hr)
torch.cuda.synchronize()
'''
out = subprocess.run([sys.executable, '-c', child], capture_output=True, text=True)
rows = [tuple(map(int, m.groups())) for m in re.finditer(r'BLOCKT level (\d+) tile (\d+) rep (\d+) of (\d+) ticks (\d+)', out.stdout)]
if not rows:
    sys.exit('no BLOCKT lines:\n' + out.stdout[-2000:] + out.stderr[-2000:])
by = collections.defaultdict(list)
for l, t, r, reps, ticks in rows:
    by[(l, reps)].append(ticks / 100.0)                     # wall_clock64: 100 MHz
print(f'{len(rows)} reports, {n} samples (us per workgroup; the minimum is the least disturbed by the printf)')
for (l, reps), v in sorted(by.items()):
    v.sort()
    print(f'  level {l:2d} x {reps} replicas: median {v[len(v) // 2]:7.1f}  max {max(v):7.1f}  min {min(v):7.1f}')
