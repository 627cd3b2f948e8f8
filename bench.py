#!/usr/bin/env python
"""Headline benchmark: ray-samples/s of the panoramic-NeRF TRAINING hot path on MI355X (+ PSNR@iter).

Workload (BASELINE.json configs[1]/[2]): synthetic 2048x1024 panorama, 128 samples per ray, hash grid L=16/T=18 +
64-wide MLPs, 16-bit tables/activations with fp32 accumulation.  One "step" = one geometry-phase training step of PeRF's
NeRFScene.train_one_step_geo (modules/scene/nerf.py:186-257) on 8192 rays PER GPU, with everything the reference's step
does (modules/scene/nerf_renderer.py:145-183):

  batch gather -> stratified marching (128 lattice intervals per ray) -> density field WITHOUT gradient on every marched
  sample (the sigma pass inside OccGridEstimator.sampling) -> transmittance scan + visibility compaction (T >= 1e-4)
  -> density field WITH gradient on the kept samples -> colour field (no grad) -> compositing -> depth + distortion loss
  -> backward through compositing / MLP / hash grid -> [RCCL all-reduce of the flat gradient] -> Adam.

Sample counts are decided on the GPU and stay there (device-side counts, capacity-sized launches), so the whole step is
ONE hipGraph.  value = ray-samples that were evaluated by BOTH fields and composited (the KEPT samples, read once from a
device counter after the timed region) per second -- the extra no-grad density evaluation of every marched sample is
work the step does on top and is not counted.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus N ...            (spawns its own N ranks through torch.distributed.run on 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: one process per GPU over RCCL, rays sharded, weights replicated, sharded gradient exchange (perf_amd/dp.py).  The line's
`value` is WEAK scaling (8192 rays per GPU per step, like N = 1); the `strong` block beside it is BASELINE config 3 as
SURVEY.md 8(e) defines it: the reference's global batch of 8192 rays split over the GPUs.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16/fp16 MFMA

# Algorithmic bytes per ray-sample of the gather/scatter kernels, at SURVEY.md 8(d)'s 16-bit figures:
#   encode: 16 levels x 8 corners x 2 features x 2 B gathered = 512 B
#   grid gradient: read-modify-write of every touched entry at 16-bit = 2 x 16 x 8 x 2 x 2 B = 1024 B
#   (the kernel accumulates in fp32/fixed point in LDS and never does an HBM RMW; the fp32 figure would be 2048 B)
ALGO_BYTES = {'perf_hashgrid_fwd': 16 * 8 * 2 * 2, 'perf_hashgrid_bwd': 2 * 16 * 8 * 2 * 2}
# what the counters say limits each kernel (profiles/): `bound` names the roofline the fraction is priced
# against, `limiter` what actually stalls the kernel
LIMITER = {'perf_hashgrid_fwd': 'L1 misses in flight: 35 L1->L2 requests per sample at a 160-180 cycle round trip (tables are L2/Infinity-Cache '
                                'resident; the tag rate is not the limit: -20 % accesses changed nothing, profiles/r03_fwd_l1_counters.json)',
           'perf_hashgrid_bwd': 'VALU issue (owner test + enqueue per sample visit); no HBM read-modify-write happens',
           'perf_mlp_fwd': 'HBM: features in, outputs out (inputs requested one tile ahead)',
           'perf_mlp_bwd': 'per-tile dependency chain at two waves per SIMD (MFMA -> pack -> LDS transpose -> MFMA) beside the 137 MB dfeat store'}
GEO_FWD_FLOP = 2 * (32 * 64 + 64 * 1)
APP_FWD_FLOP = 2 * (32 * 64 + 64 * 64 + 64 * 3)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rays-per-gpu', type=int, default=8192, help='weak scaling: rays per GPU; strong: the GLOBAL batch')
    ap.add_argument('--spp', type=int, default=128)
    ap.add_argument('--dtype', default=None, choices=['bf16', 'fp16'], help='default: perf_amd.tcnn.DEFAULT_DTYPE')
    ap.add_argument('--mode', default='train_geo', choices=['train_geo', 'train_app', 'render'])
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='scaling mode of the HEADLINE value at N > 1 (the other mode is reported beside it): weak = --rays-per-gpu '
                         'rays on every GPU; strong = that many rays in total, split over the GPUs')
    ap.add_argument('--strict-two-evaluations', action='store_true',
                    help='headline with the density field ENCODED AND EVALUATED twice per kept sample like the reference (default: the gradient pass '
                         'starts from the features and densities the sampling pass computed -- bit-identical parameters; the strict line is reported beside it)')
    ap.add_argument('--dp-mode', default='sharded', choices=['sharded', 'allreduce'], help='gradient exchange at N > 1 (perf_amd/dp.py)')
    ap.add_argument('--watchdog-seconds', type=float, default=240.0,
                    help='N > 1: if the optional measurements after the eager headline (graph-captured step, strong scaling, PSNR episode) '
                         'take longer than this, print the eager headline and leave')
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='run the timed steps eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-prepass', action='store_true',
                    help="round-1 'fixed-count' line: early_stop_eps = 0, i.e. WITHOUT the sampling-pass density evaluation and the "
                         'visibility compaction of the reference step (not the headline)')
    ap.add_argument('--cpu-rays', type=int, default=512, help='rays of the bounded CPU-baseline sample')
    ap.add_argument('--no-psnr', action='store_true')
    ap.add_argument('--no-render-block', action='store_true', help='skip the full-panorama inference measurement (`render` block)')
    ap.add_argument('--no-config4', action='store_true', help='skip the render_dense traverse (`config4` block)')
    ap.add_argument('--config4-poses', type=int, default=600, help='BASELINE config 4: poses of the dense trajectory')
    ap.add_argument('--no-train-app', action='store_true', help='skip the colour-phase training step at bench scale (`train_app` block)')
    ap.add_argument('--no-config5', action='store_true', help='skip the BASELINE config 5 panorama (`config5` block)')
    ap.add_argument('--config5-log2', type=int, nargs='*', default=[28, 30],
                    help='BASELINE config 5: log2 of the hashed levels\' table size(s); the whole 4096x2048x256 panorama is rendered once per size')
    ap.add_argument('--no-reuse-line', action='store_true', help='skip the extra measurement with the other setting of NeRFScene.reuse_sampling_features')
    ap.add_argument('--psnr-geo-iters', type=int, default=3000, help='configs/nerf.yaml:25 raw_phase_iter_geo')
    ap.add_argument('--psnr-app-iters', type=int, default=1500, help='configs/nerf.yaml:26 raw_phase_iter_app')
    ap.add_argument('--comm-dtype', default='fp32', choices=['fp32', 'bf16'], help='payload of the gradient all-reduce (N > 1)')
    ap.add_argument('--sustain-seconds', type=float, default=1.0, help='length of the second, longer measurement (0 = off)')
    return ap.parse_args()


def cpu_baseline(spp, n_rays):
    """The oracle (CPU restatement of the same algorithm, torch fp32) timed on this box's host cores on a bounded
    sample of the same workload: one geometry training step (sampling-pass sigma + compaction, forward, backward) on
    n_rays x spp samples."""
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
                           near=0.0, far=1.5, step=step, early_stop_eps=1e-4, max_steps=spp)
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
            'note': "NOT the reference's own CPU path -- it has none: NGPNeRF exits without tiny-cuda-nn (modules/fields/ngp_nerf.py:13-21); "
                    'this is oracle/perf_oracle.py, the CPU restatement of the same algorithm (torch fp32)',
            'sample': f'oracle geo training step (sampling-pass sigma + compaction + fwd + bwd), {len(o)} rays x ~{n // len(o)} kept samples, '
                      f'{reps} reps of {dt:.2f}s'}


def psnr_at_iters(args, dev, dist_mod, rank, world, start_dense=None):
    """PSNR@iter: one training episode at the reference's settings (occupancy from the supervision, step 5e-4,
    early stop 1e-4, 8192-ray global batch, --psnr-geo-iters geometry then --psnr-app-iters colour iterations) on the
    synthetic room panorama; PSNR of the eval render against the panorama at colour iterations {0, 1/3, end}.
    -> (block, scene, extras): extras carries what the `faithful` / `config4` blocks are built from (phase wall times, hipGraph
    node counts, the supervision maps; start_dense(dist_map) is called before the training so that config 4's trajectory
    sampler runs beside it)."""
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays, psnr
    torch.manual_seed(0)
    scene = NeRFScene(dtype=args.dtype)
    scene.dp_mode, scene.comm_dtype = args.dp_mode, args.comm_dtype
    scene.count_graph_nodes = world == 1 and not os.environ.get('PERF_DP_SINGLE_RANK')      # (graphs that hold RCCL collectives are left alone)
    rays = gen_pano_rays(torch.eye(4), args.height, args.width, device=dev)
    dist_map, rgb_map = synthetic.room(rays.d)
    pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb_map, dist_map)
    dense_future = start_dense(dist_map) if start_dense is not None else None
    n_app = args.psnr_app_iters
    marks = sorted({0, n_app // 3, n_app})
    curve, times = {}, {}
    phase_t = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()

    def probe(tag):
        torch.cuda.synchronize(); t = time.perf_counter()
        if rank == 0:
            out = scene.render(rays, ['rgb', 'distance'])
            torch.cuda.synchronize(); times.setdefault('renders', []).append(time.perf_counter() - t)
            curve[tag] = {'psnr_db': round(psnr(out['rgb'], rgb_map), 3),
                          'mean_abs_distance_err': round(float((out['distance'] - dist_map).abs().mean()), 5)}
            scene.set_train()
        torch.cuda.synchronize(); times['probe'] = times.get('probe', 0.0) + time.perf_counter() - t

    def cb(phase, i):
        n_it = args.psnr_geo_iters if phase == 'geo' else n_app
        if i == scene.EAGER_HEAD or i == n_it - 1:        # wall time of the graph-replayed part of a phase (one sync at either end)
            torch.cuda.synchronize()
            phase_t.setdefault(phase, []).append((i, time.perf_counter(), times.get('probe', 0.0)))
        if phase == 'geo' and i == args.psnr_geo_iters - 1 and 0 in marks:
            probe('app_iter_0')
        if phase == 'app' and (i + 1) in marks:
            probe(f'app_iter_{i + 1}')

    scene.train_one_episode(pool, args.psnr_geo_iters, n_app, callback=cb)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    block = {'schedule': f'{args.psnr_geo_iters} geometry + {n_app} colour iterations, global batch {scene.train_conf.pixel_loss_batch_size} rays, '
                         f'{args.width}x{args.height} synthetic room, reference sampling (step 5e-4, early stop 1e-4), seed 0',
             'curve': curve, 'train_seconds': round(total - times.get('probe', 0.0), 3), 'render_seconds_total': round(times.get('probe', 0.0), 3),
             'render_seconds_each': [round(t, 4) for t in times.get('renders', [])],
             'launch': 'hipGraph replay per step' if (scene.graph_steps and scene.dp_graph_ok()) else 'eager',
             'note': 'the fp32 oracle cannot run this size on a CPU; HIP-vs-oracle PSNR parity after equal iterations is asserted at 256x512 by '
                     'tests/test_gpu_psnr.py against tests/golden/psnr_curve.json'}
    faithful = {}
    for phase, marks_t in phase_t.items():
        if len(marks_t) >= 2:
            (i0, ta, pa), (i1, tb, pb) = marks_t[0], marks_t[-1]
            if i1 > i0:
                faithful[f'{phase}_ms_per_step'] = round(((tb - ta) - (pb - pa)) / (i1 - i0) * 1e3, 4)
    for phase, n_nodes in getattr(scene, 'graph_nodes', {}).items():
        faithful[f'{phase}_launches_per_step'] = n_nodes
    if faithful:
        faithful['what'] = ("the reference-faithful episode above (occupancy from the supervision, step 5e-4, variable sample counts, 8192-ray batches): "
                            'wall time per hipGraph-replayed step (probe renders excluded); launches = nodes of the captured step graph')
        ec = scene.sample_counters.tolist()
        if ec[2] > 0:
            faithful['mean_kept_samples_per_step'] = round(ec[1] / ec[2], 1)
            faithful['mean_marched_samples_per_step'] = round(ec[0] / ec[2], 1)
    return block, scene, {'faithful': faithful or None, 'rays': rays, 'dist_map': dist_map, 'rgb_map': rgb_map, 'pool': pool,
                          'dense_future': dense_future}


def render_block(args, dev, rays):
    """`north_star`: rays/s on a synthetic 2048x1024 panorama at 128 samples per ray.  Fixed-count eval render (all-occupied
    grid, 128 lattice intervals per ray, the reference's early stop at T < 1e-4) of the whole panorama from a fresh
    initialisation (nothing terminates early: every ray-sample is evaluated by both fields and composited), in the reference's
    32,768-ray batches (nerf.py:86) with device-side counts; warm; no PSNR reduction inside the timed region."""
    from perf_amd import ops
    from perf_amd.scene import NeRFScene, Rays
    torch.manual_seed(0)
    scene = NeRFScene(dtype=args.dtype)
    scene.set_eval()
    scene.estimator.set_binaries(torch.ones(256 ** 3, dtype=torch.uint8, device=dev))
    r = scene.renderer
    r.render_step_size = 0.99 / args.spp; r.far_plane = 1.5; r.early_stop_eps = 1e-4; r.max_steps = args.spp; r.head_samples = None
    B = 32768
    r.sample_capacity = B * args.spp
    flat_o = rays.o.reshape(-1, 3); flat_d = rays.d.reshape(-1, 3)
    n_rays = flat_o.shape[0]
    n_batches = (n_rays + B - 1) // B
    outs = {'rgb': torch.empty(n_rays, 3, device=dev), 'distance': torch.empty(n_rays, 1, device=dev)}

    def pano():
        with torch.no_grad():
            for b in range(n_batches):
                lo, hi = b * B, min((b + 1) * B, n_rays)
                res = scene.render_once(Rays(flat_o[lo:hi], flat_d[lo:hi]), ['rgb', 'distance', 'n_marched_dev', 'n_samples_dev'])
                outs['rgb'][lo:hi].copy_(res['rgb']); outs['distance'][lo:hi].copy_(res['distance'])
                ops.step_bookkeeping(None, None, scene.sample_counters, res['n_marched_dev'], res['n_samples_dev'])
    pano()                                             # warm-up (allocator, kernel attributes)
    scene.sample_counters.zero_()
    reps = 2
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        pano()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    c = scene.sample_counters.tolist()
    marched, kept = c[0] / reps, c[1] / reps
    ops.start_kernel_timing()
    pano()
    kern = ops.stop_kernel_timing()
    enc = kern.get('perf_hashgrid_fwd')
    algo = 2 * ALGO_BYTES['perf_hashgrid_fwd']         # one density + one colour encode per ray-sample = 1,024 B
    blk = {'what': f'{args.width}x{args.height} fixed-count eval panorama, {args.spp} samples/ray, {n_batches} batches of {B} rays, fresh '
                   'initialisation (nothing pruned), both fields + compositing, device-side counts; mean of 2 warm renders',
           'seconds_per_panorama': round(el, 5), 'rays_per_s': n_rays / el, 'ray_samples_per_s': kept / el,
           'marched_samples': marched, 'kept_samples': kept, 'dtype': args.dtype,
           'roofline': {'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS, 'algorithmic_bytes_per_ray_sample': algo,
                        'achieved': round(algo * kept / el / 1e9, 1), 'frac': round(algo * kept / el / 1e9 / HBM_PEAK_GBS, 4),
                        'definition': 'whole render: 1,024 B x kept ray-samples / wall time of the panorama'},
           'kernel_ms_per_panorama': {k: round(n * ms, 3) for k, (n, ms) in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:8]}}
    if enc:
        per_launch = ALGO_BYTES['perf_hashgrid_fwd'] * (marched + kept) / 2 / n_batches          # mean live samples of the two encodes of a batch
        blk['roofline']['encode_kernel'] = {'ms_per_launch': round(enc[1], 4), 'launches': enc[0],
                                            'algorithmic_GBps': round(per_launch / (enc[1] * 1e-3) / 1e9, 1),
                                            'frac_of_hbm_peak': round(per_launch / (enc[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return blk


def config4_block(args, dev, scene, extras, n_poses=600):
    """BASELINE config 4: CoreRunner.render_dense (core_exp_runner.py:223-246) -- a 600-pose dense trajectory through the scene the
    PSNR episode just trained, 512x1024 panoramic frames, fp16 inference, every frame ONE hipGraph replay (rays generated on
    the device from the pose).  The trajectory sampler (host, annealed tour) was started before the training and ran beside it."""
    import numpy as np
    fh, fw = 512, 1024
    t0 = time.perf_counter()
    dense = extras['dense_future'].result()
    t_wait = time.perf_counter() - t0
    poses = []
    for i in range(dense.n_poses):
        p = dense.sample_pose(i).clone().float()
        p[:3, :3] = torch.eye(3)                                        # core_exp_runner.py:232
        poses.append(p)
    for net in (scene.nerf.geo_mlp, scene.nerf.app_mlp):               # fp16 inference of the bf16-trained fields: the 16-bit
        net.dtype_name = 'fp16'                                         # working copy is re-cast from the fp32 master
    scene.nerf.dtype_name = 'fp16'
    scene.set_eval()
    out = {'what': f'render_dense: {len(poses)} poses, {fw}x{fh} frames, fp16 inference of the scene trained above, reference sampling '
                   '(step 5e-4, early stop 1e-4, two-phase sampler), one hipGraph replay per frame',
           'pose_sampler': {'start_call_s': round(extras.get('dense_start_s', 0.0), 4), 'wait_s': round(t_wait, 4),
                            'where': 'forked host worker, started before the training episode'}}
    for tag, batch, n_frames in (('frame_as_one_batch', fh * fw, len(poses)), ('reference_batches_of_32768', 32768, max(len(poses) // 5, 1))):
        frame = scene.make_graphed_render(fh, fw, ('rgb', 'distance'), batch_size=batch)
        for p in poses[:2]:
            frame(p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for p in poses[:n_frames]:
            last = frame(p)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out[tag] = {'frames': n_frames, 'seconds': round(el, 4), 'frames_per_s': round(n_frames / el, 1), 'rays_per_s': n_frames * fh * fw / el,
                    'batches_per_frame': (fh * fw + batch - 1) // batch, 'per_ray_sample_capacity': frame.state['per_ray'],
                    'last_frame_rgb_sum': float(last['rgb'].double().sum())}
    one = out['frame_as_one_batch']
    out['wall_s_600_frames_including_sampler'] = round(one['seconds'] * len(poses) / one['frames'] + extras.get('dense_start_s', 0.0) + t_wait, 4)
    return out


def config5_block(args):
    """BASELINE config 5 as far as ONE GPU goes (the run the >= 50 % HBM criterion is judged on): the whole
    4096x2048 panorama x 256 samples per ray through both L = 20 fields whose 16-bit tables exceed every cache (T = 2^28: 9.2 GiB
    per encoder; 2^30: 31 GiB, 64-bit entry offsets), NeRFOCCRenderer.render per batch of 4 rows, + compositing.  Per table size:
    ray-samples/s and the encode kernel's algorithmic and MOVED fraction of the HBM peak (the latter from the committed
    rocprofv3 PMC pass of the same batches, profiles/r06_config5_pmc.json).  Each table size THREE times: with tcnn's table layout
    (`T<k>`), with the opt-in line-local layout (`T<k>_line_local`, perf_amd.grid.GridConfig: no reference result exists for
    these grids, the layout is this build's to choose) and with its overlapping-run variant (`T<k>_line_overlap`: the same 2^T
    entries per hashed level hold 3/4 as many distinct vertices, a cell's x corner pair is always one request).
    tests/test_gpu_config5.py checks the same workload through size-independent properties."""
    from perf_amd import panorama as C
    pmc = None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r06_config5_pmc.json')))
    except Exception:      # noqa: BLE001
        pass
    out = {}
    for log2_t in args.config5_log2:
        for layout in ('tcnn', 'line_local', 'line_overlap'):
            key = f'T{log2_t}' + ('' if layout == 'tcnn' else '_' + layout)
            torch.cuda.empty_cache()
            free, _total = torch.cuda.mem_get_info()
            need = 2 * (C.LEVELS5 << (log2_t + 2)) + (8 << 30)            # upper bound: two encoders of L x 2^T entries x 2 x 2 B, + working set
            if need > free:
                out[key] = {'skipped': f'needs ~{need >> 30} GiB, {free >> 30} GiB free'}
                continue
            out[key] = C.render_panorama_block(log2_t, pmc=pmc, layout=layout)
    return out


def _flush_native_stdout():
    """Python's and the C library's stdout buffers (RCCL writes its banner with printf)."""
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:       # noqa: BLE001 -- cosmetic
        pass


def _free_port():
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close(); return port


def relaunch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
    127.0.0.1) and pass their output through."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not env.get('PERF_BENCH_ONE_DEVICE'):
        sys.stderr.write(f'bench.py: --gpus {args.gpus} but only {n_dev} HIP device(s) are visible\n')
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(relaunch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    # PERF_BENCH_BACKEND=gloo + PERF_BENCH_ONE_DEVICE=1: smoke-test the multi-rank path on a single-GPU box (dev tool, tests)
    backend = os.environ.get('PERF_BENCH_BACKEND', 'nccl')
    if os.environ.get('PERF_BENCH_ONE_DEVICE'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    # PERF_DP_SINGLE_RANK=1 (dev): a world of ONE rank on the real RCCL backend takes the data-parallel path -- what the
    # exchange machinery costs on one GPU, link time excluded
    single_rank_dp = world == 1 and os.environ.get('PERF_DP_SINGLE_RANK') == '1'
    if single_rank_dp:
        for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', str(_free_port())), ('RANK', '0'), ('WORLD_SIZE', '1')):
            os.environ.setdefault(k, v)
    if world > 1 or single_rank_dp:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # Second line of defence behind perf_amd.scene._capture's drain: should ProcessGroupNCCL's watchdog thread still query
        # the end event of an eager collective while RCCL's stream is part of a capture (HIP: hipErrorCapturedEvent), let it log
        # and retire instead of rethrowing -- a rethrow in that thread is std::terminate for the rank, and the job's line is lost.
        os.environ.setdefault('TORCH_NCCL_RETHROW_CUDA_ERRORS', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    # The library travels prebuilt (in-tree libperf_hip.so); should it be missing or stale (ABI mismatch) it is rebuilt from
    # the HIP sources -- never replaced by another code path.
    from perf_amd import _lib
    try:
        _lib.load()
    except _lib.PerfError:
        if rank == 0:
            from perf_amd import build as _build
            _build.build(force=True)
        if world > 1:
            dist.barrier()
        _lib.load()
    from perf_amd import ops, synthetic, tcnn
    from perf_amd.scene import NeRFScene, Rays, SupInfoPool, gen_pano_rays
    args.dtype = args.dtype or tcnn.DEFAULT_DTYPE
    reuse_default = not args.strict_two_evaluations

    rays = gen_pano_rays(torch.eye(4), args.height, args.width, device=dev)
    dist_map, rgb_map = synthetic.room(rays.d)
    pool = SupInfoPool()
    pool.register_rays(rays.o, rays.d, rgb_map, dist_map)

    def build(reuse_features, scaling, graph=True, dp_mode=None, mode=None):
        """Fresh scene + optimizer + (graphed) step of the benchmark workload -> dict(scene, step, eager_step, rays_per_step, graphed)."""
        mode = mode or args.mode
        torch.manual_seed(0)                                              # the same random streams on every rank (scene.py)
        scene = NeRFScene(dtype=args.dtype)
        tc = scene.train_conf
        scene.comm_dtype = args.comm_dtype
        scene.dp_mode = dp_mode or args.dp_mode
        scene.reuse_sampling_features = reuse_features
        if scaling == 'weak':
            rays_local = args.rays_per_gpu                               # 8192 rays on every GPU
        else:
            assert args.rays_per_gpu % world == 0
            rays_local = args.rays_per_gpu // world                      # the reference's 8192-ray batch split over the GPUs
        # (NeRFScene._DP_OFF: the plain single-GPU step on this rank's share of the workload, every rank for itself)
        tc.pixel_loss_batch_size = rays_local * (1 if NeRFScene._DP_OFF else world)
        # fixed-count marching: all-occupied grid, 128 lattice intervals of 0.99/128 from the (jittered) origin; the
        # reference's early termination (T >= 1e-4 after the no-grad density pass) then prunes what it prunes
        scene.set_train()
        scene.estimator.set_binaries(torch.ones(256 ** 3, dtype=torch.uint8, device=dev))
        r = scene.renderer
        r.render_step_size = 0.99 / args.spp
        r.far_plane = 1.5                                                  # the reference's value (nerf_renderer.py:150); the lattice ends at 0.99
        r.early_stop_eps = 0.0 if args.no_prepass else 1e-4
        r.max_steps = args.spp
        # one-phase density pass: fixed-count marching walks through empty space, no ray terminates inside its first few
        # samples, so the two-phase early-terminating sampler (renderer.head_samples, the default of training / eval on real
        # scenes) would only add launches here
        r.head_samples = None
        rays_per_step = rays_local if mode != 'render' else 32768
        r.sample_capacity = rays_per_step * args.spp                     # = the marched count: nothing is ever truncated
        scene.nerf.reset_geo()
        scene.sample_counters.zero_()
        use_graph = graph and (not args.no_graph) and mode != 'render' and scene.dp_graph_ok()
        graphed = None
        if mode in ('train_geo', 'train_app'):
            kind = 'geo' if mode == 'train_geo' else 'app'
            net = scene.nerf.geo_mlp if kind == 'geo' else scene.nerf.app_mlp
            conf = tc.geo_optimizer if kind == 'geo' else tc.app_optimizer
            n_sched = 3000.0 if kind == 'geo' else 1500.0
            opt = scene.make_optimizer(net, 0.0)
            step_fn = scene.train_one_step_geo if kind == 'geo' else scene.train_one_step_app

            def eager_step(i):
                scene.update_lr(opt, conf, min(i / n_sched, 0.999))
                step_fn(opt, pool, progress=0.25)
            if use_graph:
                n_rows = opt.SCHEDULE_ROWS        # the schedule of every iteration this run can reach, installed on the device
                graphed = scene.make_graphed_step(kind, opt, pool, schedule=([scene.lr_at(conf, min(i / n_sched, 0.999)) for i in range(n_rows)],
                                                                             [0.5] * n_rows, 0))

            def step(i):
                if graphed is not None:
                    graphed()                     # (nothing but the graph: learning rate and ramp come from the device-side schedule)
                else:
                    eager_step(i)
        else:
            scene.set_eval()
            n_batches = (args.height * args.width) // 32768
            flat_o = rays.o.reshape(-1, 3); flat_d = rays.d.reshape(-1, 3)

            def step(i):
                bb = i % n_batches
                with torch.no_grad():
                    res = scene.render_once(Rays(flat_o[bb * 32768:(bb + 1) * 32768], flat_d[bb * 32768:(bb + 1) * 32768]),
                                            ['rgb', 'distance', 'n_marched_dev', 'n_samples_dev'])
                    ops.step_bookkeeping(None, None, scene.sample_counters, res['n_marched_dev'], res['n_samples_dev'])
            eager_step = step
        return {'scene': scene, 'step': step, 'eager_step': eager_step, 'rays_per_step': rays_per_step, 'graphed': graphed,
                'scaling': scaling}

    def timed(run, n_steps, first):
        """n_steps steps bracketed by barrier + synchronize on both sides; -> (seconds = max over ranks, marched, kept)."""
        scene, step = run['scene'], run['step']
        scene.sample_counters.zero_()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(first + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        cnt = scene.sample_counters.clone()
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)                 # whole-job sample counts
        c = cnt.tolist()
        return el, int(c[0]), int(c[1]), {'skipped_for_overflow': int(c[4]), 'skipped_for_truncation': int(c[5])}

    def measure(reuse_features, scaling, graph=True, dp_mode=None, mode=None):
        """Build, W warmup steps, K timed steps -> (run, result dict)."""
        run = build(reuse_features, scaling, graph, dp_mode, mode)
        for i in range(args.warmup):
            run['step'](i)
        el, marched, kept, health = timed(run, args.steps, args.warmup)
        tcf = run['scene'].train_conf
        res = {'value': kept / el, 'ms_per_step': el / args.steps * 1e3, 'steps': args.steps, 'scaling': scaling,
               'global_batch_rays': tcf.pixel_loss_batch_size, 'rays_per_gpu_per_step': run['rays_per_step'],
               'marched_samples_per_gpu_per_step': marched / args.steps / world, 'kept_samples_per_gpu_per_step': kept / args.steps / world,
               'launch': ('hipGraph replay of the whole step' + (' (collectives captured with it)' if world > 1 else ''))
                         if run['graphed'] is not None else 'eager', **health}
        return run, res

    def comm_times(r, first, n_steps):
        """Eager data-parallel steps with every collective issued synchronously between two HIP events on the launch stream
        (perf_amd/dp.py: ShardedExchange._timed; NeRFScene._apply_grad for the all-reduce mode) -> ms per step per collective."""
        sc = r['scene']
        timing = {}
        sc._dp_timing = timing
        for net in (sc.nerf.geo_mlp, sc.nerf.app_mlp):
            ex = getattr(net, '_dp_exchange', None)
            if ex is not None:
                ex.timing = timing
        try:
            for i in range(n_steps):
                r['eager_step'](first + i)
            torch.cuda.synchronize()
        finally:
            sc._dp_timing = None
            for net in (sc.nerf.geo_mlp, sc.nerf.app_mlp):
                ex = getattr(net, '_dp_exchange', None)
                if ex is not None:
                    ex.timing = None
        per = {k: round(sum(a.elapsed_time(b) for a, b in v) / n_steps, 4) for k, v in timing.items()}
        t = torch.tensor([per.get(k, 0.0) for k in sorted(per)], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per = {k: round(float(v), 4) for k, v in zip(sorted(per), t.tolist())}
        sizes = None
        ex = getattr(sc.nerf.geo_mlp, '_dp_exchange', None)
        if ex is not None:
            sizes = {'reduce_scatter_in_bytes': int(ex.payload.numel() * 4), 'all_reduce_small_bytes': int(ex.ar.numel() * 4),
                     'all_gather_w16_out_bytes': int((ex.w16_full.numel() - ex.n_net) * ex.w16_full.element_size()), 'units': ex.units}
        return {'dp_mode': sc.dp_mode, 'ms_per_step_per_collective': per, 'comm_ms_per_step': round(sum(per.values()), 4), 'bytes': sizes,
                'how': 'eager steps, each collective synchronous between two HIP events on the launch stream (nothing overlaps it), max over ranks'}

    other = 'strong' if args.scaling == 'weak' else 'weak'
    notes = []
    watchdog = None
    if world > 1:
        # N > 1: the eager, collective-by-collective step first -- a line that is safe to print -- then everything optional
        # (graph capture with the RCCL collectives inside, the other scaling mode, the PSNR episode) under a watchdog that
        # prints the safe line and leaves should any of it hang on this node
        run, head = measure(reuse_default, args.scaling, graph=False)
        safe = _line(args, world, head, run, None, None, None, None, None, None,
                     notes=['eager data-parallel step only: the optional measurements did not finish within the watchdog'])

        def bail():
            if rank == 0:
                _flush_native_stdout()
                print(json.dumps(safe), flush=True)
            os._exit(0)
        watchdog = threading.Timer(args.watchdog_seconds, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            if run['scene'].dp_graph_ok() and not args.no_graph and args.mode != 'render':
                run_g, head_g = measure(reuse_default, args.scaling, graph=True)
                head_g['eager'] = {'value': head['value'], 'ms_per_step': head['ms_per_step']}
                run, head = run_g, head_g
            else:
                notes.append('data-parallel steps launched eagerly (RCCL graph-capture probe failed or PERF_DP_GRAPH=0 / non-RCCL backend)')
        except Exception as e:       # noqa: BLE001
            notes.append(f'graph-captured data-parallel step failed ({type(e).__name__}: {e}); eager line kept')
    else:
        run, head = measure(reuse_default, args.scaling)
    scene, step, eager_step = run['scene'], run['step'], run['eager_step']
    tc, r = scene.train_conf, scene.renderer

    # a second, longer measurement of the same loop (the K-step region above is only tens of milliseconds long)
    sustained = None
    if args.sustain_seconds > 0:
        n_long = int(min(20000, max(args.steps, args.sustain_seconds / max(head['ms_per_step'] * 1e-3, 1e-6))))
        if world > 1:                                                  # every rank must run the same number of steps
            t = torch.tensor([n_long], device=dev); dist.broadcast(t, 0); n_long = int(t.item())
        el2, m2, k2, h2 = timed(run, n_long, args.warmup + args.steps)
        sustained = {'steps': n_long, 'seconds': round(el2, 4), 'value': k2 / el2, 'ms_per_step': el2 / n_long * 1e3,
                     'kept_samples_per_step_per_gpu': k2 / n_long / world, 'marched_samples_per_step_per_gpu': m2 / n_long / world, **h2}

    # per-kernel times IN THE STATE THE HEADLINE WAS TIMED IN: a fresh build of the same workload, the same W warm-up steps,
    # then the same K steps launched eagerly with a HIP event pair around every C-ABI launch on the launch stream (events
    # cannot be recorded inside a graph replay; same kernels, same shapes, same device-side counts).  `kernels_late`: the same
    # pass on the scene the sustained loop left behind (the density field has formed: early termination prunes a third of the
    # samples) -- gather / scatter kernels only.
    def kernel_pass(r, first):
        r['scene'].sample_counters.zero_()
        ops.start_kernel_timing()
        for i in range(args.steps):
            r['eager_step'](first + i)
        k = ops.stop_kernel_timing()
        c = r['scene'].sample_counters.tolist()
        return k, (c[0] / max(args.steps, 1), c[1] / max(args.steps, 1))             # per step, this rank

    first = args.warmup + args.steps + (sustained['steps'] if sustained else 0)
    kern_late, ev_late = kernel_pass(run, first)
    comm_block = None
    if world > 1 or single_rank_dp:
        # per-collective times: eager data-parallel steps with every collective issued synchronously between two events
        try:
            comm_block = comm_times(run, first + args.steps, args.steps)
        except Exception as e:       # noqa: BLE001
            notes.append(f'per-collective timing failed ({type(e).__name__}: {e})')
    run_k = build(reuse_default, args.scaling, graph=False)
    for i in range(args.warmup):
        run_k['eager_step'](i)
    kern, ev_counts = kernel_pass(run_k, args.warmup)
    del run_k

    # the other setting of NeRFScene.reuse_sampling_features, from the same fresh initialisation: the reference evaluates the
    # density field twice on the kept samples (no-grad inside sampling, with grad in the renderer: same parameters, same
    # positions); by default the gradient pass starts from the features the sampling pass encoded -- bit-identical parameters
    # (tests/test_gpu_counts.py), one encode fewer.  Both lines are reported.
    reuse_block = None
    if args.mode == 'train_geo' and not args.no_prepass and not args.no_reuse_line and world == 1:
        _, alt = measure(not reuse_default, args.scaling)
        reuse_block = dict(alt, what=('strict reference order: the kept samples are encoded and their density evaluated a second time for the gradient pass'
                                      if reuse_default else 'gradient pass starts from the features and densities the sampling pass computed')
                                     + '; parameters bit-identical to the headline path')

    # the COLOUR phase's step (train_one_step_app, nerf.py:259-297: a third of every episode's iterations) at the same scale: the
    # same 8192 x 128 fixed-count batch from a fresh initialisation (kept = marched), W warm-up + K timed hipGraph replays, and
    # its own per-kernel table from an eager pass in the same state
    app_block = None
    if world == 1 and args.mode == 'train_geo' and not args.no_train_app and not args.no_prepass:
        run_a, res_a = measure(reuse_default, args.scaling, mode='train_app')
        del run_a
        run_ak = build(reuse_default, args.scaling, graph=False, mode='train_app')
        for i in range(args.warmup):
            run_ak['eager_step'](i)
        kern_a, ev_a = kernel_pass(run_ak, args.warmup)
        del run_ak
        table_a, _ = _kernel_table(args, kern_a, ev_a, True)
        app_block = {'what': 'train_one_step_app (nerf.py:259-297) on the benchmark batch: batch draw -> marching -> density field WITHOUT gradient on every '
                             'marched sample -> transmittance scan + compaction -> colour field WITH gradient (encode + 32->64->64->3 MLP) -> compositing -> '
                             'colour smooth-L1 -> backward through compositing / MLP / hash grid -> Adam; one hipGraph replay per step',
                     'value': res_a['value'], 'unit': 'ray-samples/s', 'ms_per_step': res_a['ms_per_step'], 'steps': res_a['steps'], 'warmup': args.warmup,
                     'marched_samples_per_step': res_a['marched_samples_per_gpu_per_step'], 'kept_samples_per_step': res_a['kept_samples_per_gpu_per_step'],
                     'launch': res_a['launch'], 'skipped_for_overflow': res_a['skipped_for_overflow'], 'skipped_for_truncation': res_a['skipped_for_truncation'],
                     'kernels': table_a}

    other_block = None
    if world > 1 and args.mode != 'render':
        try:
            _, other_block = measure(reuse_default, other, graph=head['launch'] != 'eager')
        except Exception as e:       # noqa: BLE001
            notes.append(f'{other}-scaling measurement failed ({type(e).__name__}: {e})')
    # N > 1: what the exchange costs.  (a) the SAME per-rank workload as a plain single-GPU step, every rank for itself (no
    # collective): exposed communication = data-parallel step - this; (b) the single all-reduce of the flat gradient that
    # `north_star` names (dp_mode = 'allreduce'), for comparison with the sharded exchange of the headline.
    if (world > 1 or single_rank_dp) and args.mode != 'render':
        try:
            NeRFScene._DP_OFF = True
            _, plain = measure(reuse_default, args.scaling)
            NeRFScene._DP_OFF = False
            comm_block = dict(comm_block or {})
            comm_block['single_rank_step_ms'] = plain['ms_per_step']
            comm_block['exposed_comm_ms'] = head['ms_per_step'] - plain['ms_per_step']
            comm_block['exposed_comm_definition'] = ('ms_per_step of the data-parallel step minus ms_per_step of the plain single-GPU step on the same per-rank '
                                                     'workload (every rank for itself, hipGraph replay, max over ranks)')
        except Exception as e:       # noqa: BLE001
            NeRFScene._DP_OFF = False
            notes.append(f'single-rank reference step failed ({type(e).__name__}: {e})')
        try:
            alt_mode = 'allreduce' if args.dp_mode == 'sharded' else 'sharded'
            alt_run, alt = measure(reuse_default, args.scaling, graph=False, dp_mode=alt_mode)
            comm_block = dict(comm_block or {})
            comm_block['other_exchange'] = {'dp_mode': alt_mode, 'value': alt['value'], 'ms_per_step': alt['ms_per_step'], 'launch': alt['launch'],
                                            'what': 'one all-reduce of the flat fp32 gradient per step, Adam on every rank (the exchange north_star names)'
                                                    if alt_mode == 'allreduce' else 'int32 reduce-scatter -> sliced Adam -> all-gather'}
            # ... and what ITS collective costs by itself (the A/B a first run on real links is read with), fp32 and bf16 payload
            try:
                ct = comm_times(alt_run, args.warmup + args.steps, max(2, min(args.steps, 10)))
                comm_block['other_exchange']['ms_per_step_per_collective'] = ct['ms_per_step_per_collective']
                if alt_mode == 'allreduce':
                    sc_alt = alt_run['scene']
                    comm_block['other_exchange']['payload_bytes'] = int(sc_alt.nerf.geo_mlp.params.numel() + sc_alt.DP_EXTRA) * 4
                    sc_alt.comm_dtype = 'bf16'
                    try:
                        el16, _, kept16, _ = timed(alt_run, args.steps, args.warmup + args.steps + 16)
                        ct16 = comm_times(alt_run, args.warmup + 2 * args.steps + 16, max(2, min(args.steps, 10)))
                        comm_block['other_exchange']['bf16_payload'] = {'value': kept16 / el16, 'ms_per_step': el16 / args.steps * 1e3,
                                                                        'ms_per_step_per_collective': ct16['ms_per_step_per_collective'],
                                                                        'payload_bytes': comm_block['other_exchange']['payload_bytes'] // 2}
                    finally:
                        sc_alt.comm_dtype = 'fp32'
            except Exception as e:       # noqa: BLE001
                notes.append(f'per-collective times of the comparison exchange failed ({type(e).__name__}: {e})')
        except Exception as e:       # noqa: BLE001
            notes.append(f'comparison exchange failed ({type(e).__name__}: {e})')

    psnr_block = None
    faithful_block = render_blk = config4_blk = None
    if not args.no_psnr and args.mode == 'train_geo':
        try:
            start_dense = None
            extras_box = {}
            if world == 1 and not args.no_config4:
                def start_dense(dist_map):
                    from perf_amd.pose_sampler import CirclePoseSampler, DenseTravelPoseSampler
                    t0 = time.perf_counter()
                    sparse = CirclePoseSampler(dist_map.reshape(args.height, args.width).cpu(), traverse_ratios=[.2, .4, .6], n_anchors_per_ratio=[8, 8, 8])
                    fut = DenseTravelPoseSampler.start(sparse, n_dense_poses=args.config4_poses)
                    extras_box['dense_start_s'] = time.perf_counter() - t0
                    return fut
            import numpy as np
            np.random.seed(0)
            psnr_block, psnr_scene, extras = psnr_at_iters(args, dev, dist, rank, world, start_dense=start_dense)
            extras.update(extras_box)
            faithful_block = extras['faithful']
            if world == 1 and not args.no_config4 and rank == 0:
                config4_blk = config4_block(args, dev, psnr_scene, extras, n_poses=args.config4_poses)
            del psnr_scene, extras
        except Exception as e:       # noqa: BLE001
            if world == 1:
                raise
            notes.append(f'PSNR episode failed ({type(e).__name__}: {e})')
    if world == 1 and not args.no_render_block and args.mode == 'train_geo':
        render_blk = render_block(args, dev, rays)
    config5_blk = None
    if world == 1 and not args.no_config5 and args.mode == 'train_geo' and args.config5_log2:
        config5_blk = config5_block(args)
    if watchdog is not None:
        watchdog.cancel()

    line = None
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.spp, args.cpu_rays)
        line = _line(args, world, head, run, sustained, kern, ev_counts, reuse_block, other_block, psnr_block, cpu=cpu, notes=notes,
                     late=(kern_late, ev_late), blocks={'train_app': app_block, 'faithful': faithful_block, 'render': render_blk, 'config4': config4_blk, 'config5': config5_blk, 'comm': comm_block})
    # The JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which a pipe buffers until the
    # process exits -- every rank flushes it out first, then rank 0 prints.
    _flush_native_stdout()
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1 or single_rank_dp:
        dist.destroy_process_group()


def _kernel_table(args, kern, ev_counts, prepass, only_priced=False):
    """Per-kernel rows of an eager timing pass: launches and ms per step; for the gather / scatter kernels also live samples per
    launch, algorithmic GB/s (SURVEY.md 8(d), 16-bit figures) and the fraction of the HBM peak."""
    table = {}
    marched_ev, kept_ev = ev_counts
    total = {k: n * ms for k, (n, ms) in kern.items()}
    for k, (n, ms) in sorted(kern.items(), key=lambda kv: -total[kv[0]]):
        if only_priced and k not in ALGO_BYTES:
            continue
        per_step = n / args.steps
        row = {'launches_per_step': round(per_step, 2), 'ms_per_launch': round(ms, 4), 'ms_per_step': round(per_step * ms, 4)}
        if k in ALGO_BYTES:
            # live samples per launch: the sampling-pass encode runs on the marched samples, every other on the kept ones
            if k == 'perf_hashgrid_fwd' and (prepass or args.mode == 'render'):
                live = (marched_ev + (per_step - 1) * kept_ev) / per_step
            else:
                live = kept_ev
            gbs = ALGO_BYTES[k] * live / (ms * 1e-3) / 1e9
            row.update({'live_samples_per_launch': round(live), 'algorithmic_bytes_per_sample': ALGO_BYTES[k],
                        'algorithmic_GBps': round(gbs, 1), 'frac_of_hbm_peak': round(gbs / HBM_PEAK_GBS, 4),
                        'limiter': LIMITER[k]})
        elif k in ('perf_mlp_fwd', 'perf_mlp_bwd'):
            row['limiter'] = LIMITER[k]
        table[k] = row
    return table, total


def _line(args, world, head, run, sustained, kern, ev_counts, reuse_block, other_block, psnr_block, cpu=None, notes=None, late=None, blocks=None):
    """The JSON line of one run (rank 0)."""
    scene = run['scene']
    tc, r = scene.train_conf, scene.renderer
    table, roof = {}, None
    late_table = None
    prepass = r.early_stop_eps > 0 and args.mode != 'render'
    if late and late[0]:
        late_table, _ = _kernel_table(args, late[0], late[1], prepass, only_priced=True)
    if kern:
        table, total = _kernel_table(args, kern, ev_counts, prepass)
        dom = max((k for k in total if k in ALGO_BYTES), key=lambda k: total[k])
        roof = {'kernel': dom, 'bound': 'hbm', 'stalls_on': BOUND[dom], 'achieved': table[dom]['algorithmic_GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': table[dom]['frac_of_hbm_peak'], 'traffic': None, 'limiter': LIMITER[dom],
                'priced_against': 'hbm (algorithmic bytes / HBM peak, as SURVEY.md 8(d) prescribes: a gather / scatter kernel); `stalls_on` '
                                  'names what the counters say the kernel actually waits for',
                'algorithmic_bytes_per_ray_sample': ALGO_BYTES[dom], 'live_samples_per_launch': table[dom]['live_samples_per_launch'],
                'ms_per_launch': table[dom]['ms_per_launch'],
                'definition': 'achieved = algorithmic bytes (SURVEY.md 8(d), 16-bit figures) x live samples of a launch / mean launch duration '
                              '(HIP events on the launch stream)',
                'state': 'the state `value` was timed in: a fresh build of the workload, the same warm-up, the same K steps launched eagerly '
                         '(kept = marched: nothing is pruned yet); `kernels_late` holds the same kernels after the sustained loop'}
        # HBM traffic per launch from the committed rocprofv3 PMC passes of this same workload (profiles/)
        for name in PMC_FILES:
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', name)))
                if (pmc.get('workload') == _workload_key(args) and dom in pmc['kernels']):
                    roof['traffic'] = pmc['kernels'][dom]['hbm_bytes_per_launch']
                    roof['traffic_source'] = f'profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections; raw CSVs beside it)'
                    break
            except Exception:      # noqa: BLE001
                continue
    step_desc = {'train_geo': 'geometry training step', 'train_app': 'colour training step', 'render': 'eval render batch (32768 rays)'}[args.mode]
    reuse = scene.reuse_sampling_features and r.early_stop_eps > 0 and args.mode == 'train_geo'
    line = {
        'metric': 'ray-samples/sec (panoramic NeRF ' + step_desc + ': kept samples evaluated by both fields + composited'
                  + (', fwd+bwd+Adam' if args.mode != 'render' else '') + ') + PSNR@iter',
        'value': head['value'], 'unit': 'ray-samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': head['scaling'], 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': f'{args.width}x{args.height} synthetic room panorama, {args.spp} marched samples/ray, '
                               f'hash grid L16/F2/T18 + 64-wide MLPs, mode={args.mode}, '
                               + ('reference step incl. sampling-pass sigma + visibility compaction (early stop 1e-4)' if r.early_stop_eps > 0
                                  else 'fixed-count WITHOUT sampling-pass sigma / compaction (early_stop_eps = 0)')
                               + ('; the gradient pass of the density field starts from the features (and densities) the sampling pass computed' if reuse else ''),
                   'rays_per_gpu_per_step': head['rays_per_gpu_per_step'], 'marched_samples_per_gpu_per_step': head['marched_samples_per_gpu_per_step'],
                   'kept_samples_per_gpu_per_step': head['kept_samples_per_gpu_per_step'],
                   'counted': 'kept samples (device counter, read once after the timed region); the no-grad density pass over all marched samples is extra work',
                   'parallelism': (f'dp{world} (rays sharded, weights replicated; per step: int32 reduce-scatter of the fixed-point gradient '
                                   f'fields -> Adam on 1/{world} of the table -> all-gather of the 16-bit copy), {head["scaling"]} scaling, '
                                   f'global batch {head["global_batch_rays"]} rays' if args.dp_mode == 'sharded' else
                                   f'dp{world} (one all-reduce of the flat {args.comm_dtype} gradient per step), {head["scaling"]} scaling, '
                                   f'global batch {head["global_batch_rays"]} rays') if world > 1 else 'single GPU',
                   'per_gpu_value': head['value'] / world,
                   'launch': head['launch'],
                   'kernel_timing': 'HIP events around every launch of K eager steps from a fresh build + the same warm-up (the state of the timed '
                                    'region); kernels_late: the same pass after the sustained loop'},
        'health': {k: head[k] for k in ('skipped_for_overflow', 'skipped_for_truncation')},
        'sustained': sustained,
        ('strict_two_evaluations' if reuse else 'with_feature_reuse'): reuse_block,
        'psnr': psnr_block, 'roofline': roof, 'cpu_baseline': cpu, 'kernels': table or None, 'kernels_late': late_table,
    }
    for k, v in (blocks or {}).items():
        if v is not None:
            line[k] = v
    # the scalars a reader of a TRUNCATED line needs, flat and early (the blocks they come from follow in full)
    summ = {}
    try:
        if psnr_block:
            summ['episode_train_seconds'] = psnr_block.get('train_seconds')
            last = sorted(psnr_block.get('curve', {}).items(), key=lambda kv: int(kv[0].rsplit('_', 1)[1]))
            if last:
                summ['episode_psnr_db'] = last[-1][1].get('psnr_db')
        fb = (blocks or {}).get('faithful')
        if fb:
            summ.update(faithful_geo_ms=fb.get('geo_ms_per_step'), faithful_app_ms=fb.get('app_ms_per_step'), faithful_geo_graph_nodes=fb.get('geo_launches_per_step'))
        c4 = (blocks or {}).get('config4')
        if c4 and c4.get('frame_as_one_batch'):
            summ['config4_frames_per_s'] = round(c4['frame_as_one_batch']['frames_per_s'], 1)
        rb = (blocks or {}).get('render')
        if rb and rb.get('ray_samples_per_s'):
            summ['render_ray_samples_per_s'] = rb['ray_samples_per_s']
        ta = (blocks or {}).get('train_app')
        if ta and ta.get('ms_per_step'):
            summ['train_app_ms_per_step'] = ta['ms_per_step']
        for key, blk in ((blocks or {}).get('config5') or {}).items():
            if isinstance(blk, dict) and 'roofline' in blk:
                summ[f'config5_{key}'] = {'seconds_per_panorama': blk['seconds_per_panorama'], 'ray_samples_per_s': round(blk['ray_samples_per_s']),
                                          'encode_ms': blk['roofline']['ms_per_launch'], 'encode_algorithmic_frac': blk['roofline']['frac'],
                                          'encode_moved_frac': blk['roofline'].get('moved_frac')}
    except Exception:       # noqa: BLE001 -- a convenience copy, never a failure
        pass
    if summ:
        line = {**{k: line[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step')}, 'summary': summ,
                **{k: v for k, v in line.items() if k not in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step')}}
    if 'eager' in head:
        line['config']['eager_launch'] = head['eager']
    if world > 1:
        line[('strong' if head['scaling'] == 'weak' else 'weak')] = other_block
    if notes:
        line['notes'] = notes
    if summ:
        # ... and once more as the LAST key: a log that keeps only the final ~2,000 characters of the line still carries every headline scalar
        line['summary_again'] = {'ms_per_step': line['ms_per_step'], 'roofline_frac': (line.get('roofline') or {}).get('frac'), **summ}
    return line


# what the counters say bounds the dominant kernels (profiles/r03_*): the encode is bound by the L1's request rate, the
# grid backward by VALU issue; neither by HBM bandwidth -- `frac` is nevertheless priced against HBM
BOUND = {'perf_hashgrid_fwd': 'l1-miss', 'perf_hashgrid_bwd': 'valu'}
PMC_FILES = ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json')


def _workload_key(args):
    return f'{args.mode}:{args.rays_per_gpu}x{args.spp}:prepass={0 if args.no_prepass else 1}'


if __name__ == '__main__':
    main()
