#!/usr/bin/env python
"""Headline benchmark: ray-samples/s of the panoramic-NeRF TRAINING hot path on MI355X.

Workload (BASELINE.json configs[1]/[2]): synthetic 2048x1024 panorama, 128 samples per ray, hash grid L=16/T=18 +
64-wide MLPs, 16-bit tables/activations with fp32 accumulation.  One "step" = one geometry-phase training step of
PeRF's NeRFScene.train_one_step_geo (modules/scene/nerf.py:186-257) on 8192 rays PER GPU: batch gather, stratified
sampling, density field forward (with grad) + colour field forward (no grad), compositing, depth + distortion loss,
backward through compositing / MLP / hash grid, [RCCL all-reduce of the flat gradient], Adam.  Every one of the
R*128 ray-samples is evaluated by both fields and composited (fixed-count mode: all-occupied grid, no early-stop
pruning), so value = n_gpus * rays_per_gpu * 128 * steps / seconds.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16/fp16 MFMA

# algorithmic bytes / flops per ray-sample of each kernel (DESIGN.md section "Kernels")
ALGO = {
    'perf_hashgrid_fwd': ('hbm', 16 * 8 * 2 * 2),             # 16 levels x 8 corners x 2 features x 2 B gathered
    'perf_hashgrid_bwd': ('hbm', 2 * 16 * 8 * 2 * 4),         # fp32 read-modify-write of every touched entry
    'perf_mlp_fwd': ('mfma', None),                           # filled per network below
    'perf_mlp_bwd': ('mfma', None),
}
GEO_FWD_FLOP = 2 * (32 * 64 + 64 * 1)
APP_FWD_FLOP = 2 * (32 * 64 + 64 * 64 + 64 * 3)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rays-per-gpu', type=int, default=8192)
    ap.add_argument('--spp', type=int, default=128)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16'])
    ap.add_argument('--mode', default='train_geo', choices=['train_geo', 'train_app', 'render'])
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='run the timed steps eagerly instead of replaying a hipGraph')
    ap.add_argument('--cpu-rays', type=int, default=512, help='rays of the bounded CPU-baseline sample')
    return ap.parse_args()


def cpu_baseline(spp, n_rays):
    """The oracle (CPU restatement of the same algorithm, torch fp32) timed on this box's host cores on a bounded
    sample of the same workload: one geometry training step (forward + backward) on n_rays x spp samples."""
    import numpy as np
    from oracle import perf_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    gs, as_ = O.geo_spec(), O.app_spec()
    geo = O.init_field_params(gs).requires_grad_(True); app = O.init_field_params(as_)
    o, d = O.pano_rays(torch.eye(4), 16, 32)
    o = o.reshape(-1, 3)[:n_rays].contiguous(); d = d.reshape(-1, 3)[:n_rays].contiguous()
    gt, _ = O.synthetic_room(d)
    occ = np.ones((8, 8, 8), bool)
    step = 0.99 / spp

    def one():
        out = O.occ_render(o, d, geo, app, occ, [-1, -1, -1, 1, 1, 1], training=True,
                           t0=np.zeros(len(o), np.float32), bg_color=torch.rand(len(o), 3), dist_noise=torch.rand(len(o), 1),
                           near=0.0, far=10.0, step=step, early_stop_eps=0.0, max_steps=spp)
        loss, _, _ = O.geo_step_loss(out, gt, 0.25)
        geo.grad = None
        loss.backward()
        return out['ray_indices'].numel()

    n = one()
    t0 = time.perf_counter()
    reps = 0
    while True:
        n = one(); reps += 1
        if time.perf_counter() - t0 > 10.0 or reps >= 20:
            break
    dt = (time.perf_counter() - t0) / reps
    return {'value': n / dt, 'unit': 'ray-samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'oracle geo training step (fwd+bwd), {len(o)} rays x ~{n // len(o)} samples, {reps} reps of {dt:.2f}s'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    # PERF_BENCH_BACKEND=gloo + PERF_BENCH_ONE_DEVICE=1: smoke-test the multi-rank path on a single-GPU box (dev tool)
    backend = os.environ.get('PERF_BENCH_BACKEND', 'nccl')
    if os.environ.get('PERF_BENCH_ONE_DEVICE'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from perf_amd import ops, synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays

    torch.manual_seed(0)
    scene = NeRFScene(dtype=args.dtype)
    tc = scene.train_conf
    tc.pixel_loss_batch_size = args.rays_per_gpu * world          # weak scaling: 8192 rays on every GPU
    rays = gen_pano_rays(torch.eye(4), args.height, args.width, device=dev)
    dist_map, rgb_map = synthetic.room(rays.d)
    pool = SupInfoPool()
    pool.register_rays(rays.o, rays.d, rgb_map, dist_map)

    # fixed-count sampling: all-occupied grid, 128 lattice intervals of 0.99/128 from the (jittered) origin
    scene.set_train()
    scene.estimator.set_binaries(torch.ones(256 ** 3, dtype=torch.uint8, device=dev))
    r = scene.renderer
    r.render_step_size = 0.99 / args.spp
    r.far_plane = 10.0
    r.early_stop_eps = 0.0
    r.max_steps = args.spp
    r.sample_capacity = args.rays_per_gpu * args.spp if args.mode != 'render' else None
    scene.nerf.reset_geo()
    gen = torch.Generator(device=dev); gen.manual_seed(1234)          # same index stream on every rank

    use_graph = (world == 1) and (not args.no_graph) and args.mode != 'render'
    graphed = None
    if args.mode == 'train_geo':
        opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
        conf = tc.geo_optimizer

        def eager_step(i):
            scene.update_lr(opt, conf, min(i / 3000.0, 0.999))
            scene.train_one_step_geo(opt, pool, progress=0.25, generator=None if world == 1 else gen)
        if use_graph:
            graphed = scene.make_graphed_step('geo', opt, pool)

        def step(i):
            if graphed is not None:
                graphed(scene.lr_at(conf, min(i / 3000.0, 0.999)), 0.25)
            else:
                eager_step(i)
    elif args.mode == 'train_app':
        opt = scene.make_optimizer(scene.nerf.app_mlp, 0.0)
        conf = tc.app_optimizer

        def eager_step(i):
            scene.update_lr(opt, conf, min(i / 1500.0, 0.999))
            scene.train_one_step_app(opt, pool, progress=0.25, generator=None if world == 1 else gen)
        if use_graph:
            graphed = scene.make_graphed_step('app', opt, pool)

        def step(i):
            if graphed is not None:
                graphed(scene.lr_at(conf, min(i / 1500.0, 0.999)), 0.25)
            else:
                eager_step(i)
    else:
        scene.set_eval()
        n_batches = (args.height * args.width) // 32768
        flat_o = rays.o.reshape(-1, 3); flat_d = rays.d.reshape(-1, 3)
        from perf_amd.scene import Rays

        def step(i):
            b = i % n_batches
            with torch.no_grad():
                scene.render_once(Rays(flat_o[b * 32768:(b + 1) * 32768], flat_d[b * 32768:(b + 1) * 32768]), ['rgb', 'distance'])
        eager_step = step

    rays_per_step = args.rays_per_gpu if args.mode != 'render' else 32768
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # The timed region is never instrumented (HIP events cannot be recorded inside a graph replay, and eager event
    # pairs would add host work per launch); per-kernel times come from an instrumented re-run of the same steps.
    instrument_inline = False
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if instrument_inline:
        kern = ops.stop_kernel_timing()
    else:
        # same K steps again, launched eagerly with a HIP event pair around every kernel (same kernels, same shapes)
        ops.start_kernel_timing()
        for i in range(args.steps):
            eager_step(args.warmup + args.steps + i)
        kern = ops.stop_kernel_timing()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step = rays_per_step * args.spp
    value = world * samples_per_step * args.steps / elapsed

    if rank == 0:
        # ---- roofline of the dominant kernel, from HIP events recorded over the timed region -----------------
        total = {k: n * ms for k, (n, ms) in kern.items()}
        table = {}
        for k, (n, ms) in sorted(kern.items(), key=lambda kv: -total[kv[0]]):
            per_step = n / args.steps
            table[k] = {'launches_per_step': round(per_step, 2), 'ms_per_launch': round(ms, 4), 'ms_per_step': round(per_step * ms, 4)}
        dom = max((k for k in total if k in ALGO), key=lambda k: total[k])
        n_l, ms_l = kern[dom]
        kind, per_sample = ALGO[dom]
        if kind == 'hbm':
            achieved = samples_per_step * per_sample / (ms_l * 1e-3) / 1e9
            roof = {'kernel': dom, 'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None,
                    'algorithmic_bytes_per_ray_sample': per_sample, 'ms_per_launch': round(ms_l, 4)}
        else:
            flop = (GEO_FWD_FLOP if args.mode != 'train_app' else APP_FWD_FLOP) * (1 if dom == 'perf_mlp_fwd' else 3)
            achieved = samples_per_step * flop / (ms_l * 1e-3) / 1e12
            roof = {'kernel': dom, 'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(achieved / MFMA_PEAK_TFLOPS, 5), 'traffic': None, 'ms_per_launch': round(ms_l, 4)}
        # HBM traffic per launch from the committed rocprofv3 PMC passes of this same workload (profiles/)
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))['kernels']
            if roof['kernel'] in pmc and args.mode == 'train_geo' and args.rays_per_gpu == 8192 and args.spp == 128:
                roof['traffic'] = pmc[roof['kernel']]['hbm_bytes_per_launch']
                roof['traffic_source'] = 'profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, gfx950 x2 fetch correction)'
                roof['algorithmic_bytes_per_launch'] = samples_per_step * per_sample if kind == 'hbm' else None
        except Exception:
            pass
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.spp, args.cpu_rays)
        line = {
            'metric': 'ray-samples/sec (panoramic NeRF training step: both fields evaluated + composited, fwd+bwd+Adam)',
            'value': value, 'unit': 'ray-samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{args.width}x{args.height} synthetic room panorama, {args.spp} samples/ray fixed-count, '
                                   f'hash grid L16/F2/T18 + 64-wide MLPs, mode={args.mode}',
                       'rays_per_gpu_per_step': rays_per_step, 'ray_samples_per_gpu_per_step': samples_per_step,
                       'parallelism': f'dp{world} (rays sharded, one RCCL all-reduce of the flat gradient per step)' if world > 1 else 'single GPU',
                       'per_gpu_value': value / world,
                       'launch': 'hipGraph replay of the whole step' if graphed is not None else 'eager',
                       'kernel_timing': 'HIP events around every launch in an eager re-run of the same steps right after the timed region'},
            'roofline': roof, 'cpu_baseline': cpu, 'kernels': table,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
