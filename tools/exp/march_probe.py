"""What the marching launches of the reference-faithful training step cost (8,192 jittered rays of the synthetic room, step 5e-4,
far 1.5: 3,001 lattice intervals per ray): perf_occ_march_count_head (K = 2) and the tail's perf_occ_march_write_points, on the
repeated-addition lattice (default) and on the single-rounding one.   python tools/exp/march_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from perf_amd import ops, synthetic
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays

torch.manual_seed(0)
scene = NeRFScene()
rays = gen_pano_rays(torch.eye(4), 1024, 2048)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
scene.set_train(); scene.prepare_occupancy(pool)
est = scene.estimator
idx = torch.randint(0, len(pool), (8192,), device='cuda')
o = pool.all_sup_rays.o[idx].contiguous(); d = pool.all_sup_rays.d[idx].contiguous()
u = torch.rand(8192, device='cuda')
step, far = 5e-4, 1.5
max_steps = 3001
aabb = est._aabb_host


REPS = int(os.environ.get('MARCH_PROBE_REPS', '50'))


def timed(fn, reps=None):
    reps = reps or REPS
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


out = {}
for lattice in ('repeated', 'single'):
    for tag, t0 in (('jittered', (u, step, 0.0)),) + ((('jittered_runs', (u, step, 0.0, ops.lattice_runs((u, step, 0.0), step, max_steps))),) if lattice == 'repeated' else ()) + ( ('no_jitter_table', (None, 0.0, 0.0, ops.lattice_table(0.0, step, max_steps, lattice))),
                    ('no_jitter_runs', (None, 0.0, 0.0))):
        if tag == 'jittered_runs':
            print('lattice_runs_us', round(timed(lambda: ops.lattice_runs((u, step, 0.0), step, max_steps)), 2), flush=True)
        f1 = lambda: ops.occ_march_count_head(o, d, t0, est.occ_bits(), 256, aabb, far, step, max_steps, est.occ_coarse(), 2, aabb, lattice=lattice)
        f0 = lambda: ops.occ_march_count(o, d, t0, est.occ_bits(), 256, aabb, far, step, max_steps, est.occ_coarse(), lattice=lattice)
        masks, counts, head = f1()
        ct = (counts - 2).clamp_(min=0)
        off, tot = ops.exclusive_scan_i32(ct)
        f2 = lambda: ops.occ_march_write(t0, masks, ct, off, 8192 * 64, step, max_steps, o, d, aabb, rank_lo=2, lattice=lattice)
        cf = counts.float()
        out[f'{lattice}/{tag}'] = {'max_count': int(counts.max()), 'p99_count': float(cf.quantile(0.99)), 'p90_count': float(cf.quantile(0.9)), 'count_head_us': round(timed(f1), 2), 'count_us': round(timed(f0), 2), 'write_tail_us': round(timed(f2), 2),
                                   'mean_count': float(counts.float().mean()), 'tail_total': int(tot.item())}
        print(lattice, tag, out[f'{lattice}/{tag}'], flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/march_probe.json', 'w'), indent=1)
