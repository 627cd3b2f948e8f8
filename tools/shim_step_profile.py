"""Where the time of the operator-shim training step goes (the boundary `north_star` literally names: tinycudann's nn.Module API +
nerfacc's functions + torch_efficient_distloss under the reference's OWN step: torch autograd, GradScaler-style scaling,
torch.optim.Adam -- modules/scene/nerf.py:186-297): the mirror configured to take exactly those paths (fused_steps = False,
fused_adam = False), `--steps` geometry + `--steps` colour steps of 8192 rays at the reference's settings after a warm-up that
lets the density field form (so that the sample counts are the episode's: ~16 k kept of ~35 k marched per step).

  python tools/shim_step_profile.py [--steps 200] [--out gpurun_out/r05_shim]          host timeline (torch.profiler) + wall times
  rocprofv3 --kernel-trace --stats ... -- python tools/shim_step_profile.py --no-profiler   kernel side

Writes <out>_host.json: wall ms per step, the top host-side operators by self CPU time, the top device kernels, the number of
device synchronisations per step (every .item() / boolean-mask indexing of the autograd formulation is one)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perf_amd import synthetic
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--warm-geo', type=int, default=600)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--out', default='gpurun_out/r05_shim')
ap.add_argument('--no-profiler', action='store_true')
ap.add_argument('--three-calls', action='store_true', help="a field's backward as three boundary calls (round 5) instead of perf_field_bwd: the host-cost A/B")
ap.add_argument('--fused-adam', action='store_true', help='the shim-level autograd step with the fused Adam kernel instead of torch.optim.Adam')
args = ap.parse_args()

if args.three_calls:
    from perf_amd import ops as _ops
    _ops.FIELD_BWD_ONE_CALL = False
H, W = 1024, 2048
rays = gen_pano_rays(torch.eye(4), H, W)
dist, rgb = synthetic.room(rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
torch.manual_seed(0)
scene = NeRFScene(dtype=args.dtype)
scene.fused_steps = False
scene.fused_adam = bool(args.fused_adam)
scene.set_train()
scene.prepare_occupancy(pool, 'direct')
scene.nerf.reset_geo()
tc = scene.train_conf
geo_opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
app_opt = scene.make_optimizer(scene.nerf.app_mlp, 0.0)


def geo_step(i, n):
    scene.update_lr(geo_opt, tc.geo_optimizer, min(i / n, 0.999))
    scene.train_one_step_geo(geo_opt, pool, progress=min(i / 1500.0, 1.0))


def app_step(i, n):
    scene.update_lr(app_opt, tc.app_optimizer, min(i / n, 0.999))
    scene.train_one_step_app(app_opt, pool, progress=min(i / n, 1.0))


for i in range(args.warm_geo):
    geo_step(i, 3000)
for i in range(20):
    app_step(i, 1500)
torch.cuda.synchronize()


def timed(fn, n, first, total):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn(first + i, total)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {'config': f'shim-level steps (autograd + {"fused Adam" if args.fused_adam else "torch.optim.Adam"}), 8192 rays, reference sampling, {W}x{H} synthetic room, '
                 f'{args.dtype}, after {args.warm_geo} warm-up geometry steps',
       'geo_ms_per_step': round(timed(geo_step, args.steps, args.warm_geo, 3000), 4),
       'app_ms_per_step': round(timed(app_step, args.steps, 20, 1500), 4)}
c = scene.sample_counters.tolist()
if not args.no_profiler:
    from torch.profiler import ProfilerActivity, profile
    for kind, fn, first, total in (('geo', geo_step, args.warm_geo + args.steps, 3000), ('app', app_step, 20 + args.steps, 1500)):
        n = min(args.steps, 50)
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for i in range(n):
                fn(first + i, total)
            torch.cuda.synchronize()
        ev = prof.key_averages()
        host = sorted(ev, key=lambda e: -e.self_cpu_time_total)[:25]
        devk = sorted(ev, key=lambda e: -getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0)))[:25]
        dt = lambda e: getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0))
        sync = [e for e in ev if 'Synchronize' in e.key or e.key in ('aten::item', 'aten::_local_scalar_dense', 'aten::nonzero')]
        out[kind] = {'profiled_steps': n,
                     'host_self_cpu_us_per_step': {e.key[:80]: [round(e.self_cpu_time_total / n, 1), round(e.count / n, 2)] for e in host},
                     'device_self_us_per_step': {e.key[:80]: [round(dt(e) / n, 1), round(e.count / n, 2)] for e in devk if dt(e) > 0},
                     'sync_like_calls_per_step': {e.key[:80]: round(e.count / n, 2) for e in sync},
                     'sum_device_us_per_step': round(sum(dt(e) for e in ev) / n, 1),
                     'sum_host_self_cpu_us_per_step': round(sum(e.self_cpu_time_total for e in ev) / n, 1)}
        try:
            prof.export_chrome_trace(f'{args.out}_{kind}_trace.json')
        except Exception:      # noqa: BLE001
            pass
os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
json.dump(out, open(args.out + '_host.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}, indent=1))
for kind in ('geo', 'app'):
    if kind in out:
        print(kind, 'top host ops:', list(out[kind]['host_self_cpu_us_per_step'].items())[:12])
        print(kind, 'sync-like:', out[kind]['sync_like_calls_per_step'])
