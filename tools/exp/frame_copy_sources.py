#!/usr/bin/env python
"""Which Python lines of an eval frame issue torch kernels (copies, fills, element-wise ops: launches a captured frame replays every
time): one eager render_once of a 512x1024 frame (occupancy from a synthetic room, untrained fields) under the torch profiler."""
import sys, collections
sys.path.insert(0, '.')
import torch
from perf_amd import ops, synthetic
from perf_amd.scene import NeRFScene, Rays, SupInfoPool
from perf_amd.ops import pano_raygen

scene = NeRFScene(dtype='fp16')
H, W = 512, 1024
o, d = pano_raygen(torch.eye(4), H, W)
o = o.reshape(-1, 3); d = d.reshape(-1, 3)
dist, rgb = synthetic.room(d)
pool = SupInfoPool(); pool.register_rays(o, d, rgb, dist)
scene.set_train(); scene.prepare_occupancy(pool); scene.set_eval()
scene.renderer.sample_capacity = H * W * 64
rays = Rays(o, d)
with torch.no_grad():
    for _ in range(2):
        scene.render_once(rays, ['rgb', 'distance', 'n_marched_dev'])
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        scene.render_once(rays, ['rgb', 'distance', 'n_marched_dev'])
        torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith('aten::') and ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::mul', 'aten::ones', 'aten::zeros', 'aten::rand', 'aten::clone',
                                                     'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::add', 'aten::sub', 'aten::div', 'aten::cat', 'aten::index', 'aten::select'):
        st = [s for s in (ev.stack or []) if 'perf_amd' in s]
        cnt[(ev.name, st[0] if st else '?')] += 1
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(c, name, where)
print('--- device kernels of the call:')
k = collections.Counter(ev.name for ev in prof.events() if ev.device_type is not None and str(ev.device_type).endswith('CUDA'))
for n, c in k.most_common(40):
    print(c, n[:110])
