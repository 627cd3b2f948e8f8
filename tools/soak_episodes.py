"""Soak: N training episodes back to back on one scene object, the way core_exp_runner.py:170-221 drives NeRFScene (every
episode resets the geometry field, rebuilds the occupancy and trains 3000 + 1500 iterations) -- watches what a single
episode cannot show: the closed-loop fixed-point headroom across `reset_geo`, the device-side health counters, graph
re-capture, memory growth, PSNR drift.

  python tools/soak_episodes.py [--episodes 25] [--dtype bf16] [--scene room|doorway|pillars] [--no-shrink] [--rccl-single-rank]
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/soak_episodes.py --one-device ...    (data-parallel soak:
      the ranks share GPU 0 over gloo -- what a one-GPU box can say about the sharded exchange with lagged units: steps dropped by the
      job-wide gate, parameters identical on every rank; rank 0 prints)"""
import argparse, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perf_amd import synthetic, tcnn
from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays, psnr

ap = argparse.ArgumentParser()
ap.add_argument('--episodes', type=int, default=25)
ap.add_argument('--geo', type=int, default=3000)
ap.add_argument('--app', type=int, default=1500)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--scene', default='room', choices=sorted(synthetic.SCENES), help='synthetic scene family (perf_amd/synthetic.py)')
ap.add_argument('--no-shrink', action='store_true', help='keep the sample capacity where the phase starts it (NeRFScene.auto_shrink_capacity = False; the digests must not change)')
ap.add_argument('--eager', action='store_true', help='no hipGraph replays (the digests must not change)')
ap.add_argument('--no-reuse', action='store_true', help='strict two-encode order (the digests must not change)')
ap.add_argument('--head', type=int, default=-1, help='renderer.head_samples (0: one-phase sampler; the kept samples -- and the digests -- must not change)')
ap.add_argument('--autograd', action='store_true', help='the drop-in path: torch autograd through the tinycudann / nerfacc module API and torch.optim.Adam (what an unmodified PeRF runs)')
ap.add_argument('--height', type=int, default=512)
ap.add_argument('--width', type=int, default=1024)
ap.add_argument('--batch', type=int, default=8192, help='global batch in rays (pixel_loss_batch_size)')
ap.add_argument('--one-device', action='store_true', help='multi-rank launch (torch.distributed.run) whose ranks all use GPU 0 over gloo')
ap.add_argument('--out', default=None, help='rank 0 also writes the summary JSON here')
ap.add_argument('--rccl-single-rank', action='store_true', help='a world of one rank on the RCCL backend takes the data-parallel path (PERF_DP_SINGLE_RANK)')
args = ap.parse_args()
if args.rccl_single_rank:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29593', RANK='0', WORLD_SIZE='1', PERF_DP_SINGLE_RANK='1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
if world > 1 and not args.rccl_single_rank:
    import torch.distributed as dist
    if args.one_device:
        torch.cuda.set_device(0)
        dist.init_process_group('gloo')
    else:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl', device_id=torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))))
torch.manual_seed(0)
scene = NeRFScene(dtype=args.dtype)
scene.train_conf.pixel_loss_batch_size = args.batch
scene.graph_steps = not args.eager
if args.autograd:
    scene.fused_steps = False; scene.fused_adam = False
if args.head >= 0:
    scene.renderer.head_samples = args.head or None
scene.reuse_sampling_features = not args.no_reuse
scene.auto_shrink_capacity = not args.no_shrink
H, W = args.height, args.width
rays = gen_pano_rays(torch.eye(4), H, W)
dist_gt, rgb = synthetic.SCENES[args.scene](rays.d)
pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist_gt)
rows = []
for ep in range(args.episodes):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    scene.train_one_episode(pool, args.geo, args.app)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    scene.set_eval()
    out = scene.render(rays, ['rgb', 'distance'])
    c = scene.sample_counters.tolist()
    digest = hashlib.sha256(scene.nerf.geo_mlp.params.detach().cpu().numpy().tobytes() + scene.nerf.app_mlp.params.detach().cpu().numpy().tobytes()).hexdigest()[:16]
    rows.append({'episode': ep, 'params_sha256_16': digest, 'seconds': round(t1 - t0, 3), 'psnr_dB': round(psnr(out['rgb'], rgb), 3),
                 'mean_abs_distance_err': round(float((out['distance'] - dist_gt).abs().mean()), 5),
                 'skipped_for_overflow': int(c[4]), 'skipped_for_truncation': int(c[5]), 'grid_gradient_mode': scene.nerf.app_mlp.grid_grad_accum,
                 'fp32_repairs_app_net': scene.nerf.app_mlp.fp32_redo_count(),
                 'sample_capacity': scene.renderer.sample_capacity, 'mem_alloc_MB': round(torch.cuda.memory_allocated() / 2 ** 20, 1),
                 'mem_reserved_MB': round(torch.cuda.memory_reserved() / 2 ** 20, 1)})
    if world > 1 and not args.rccl_single_rank:
        # every rank must hold the same parameters after the episode's sync_params(); the health counters are per rank
        import torch.distributed as dist
        t = torch.tensor([int(digest, 16) % (1 << 52), int(c[4]), int(c[5])], dtype=torch.float64, device='cuda')
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        rows[-1]['ranks_hold_identical_parameters'] = bool(lo[0] == hi[0])
        rows[-1]['skipped_for_overflow'], rows[-1]['skipped_for_truncation'] = int(hi[1]), int(hi[2])
    if rank == 0:
        print(json.dumps(rows[-1]), flush=True)
ps = [r['psnr_dB'] for r in rows]
digest = rows[-1]['params_sha256_16']
summary = json.dumps({'params_sha256_16': digest, 'data_parallel': bool(args.rccl_single_rank) or world > 1, 'world': world, 'dp_units': scene.dp_units, 'scene': args.scene, 'auto_shrink_capacity': not args.no_shrink, 'config': f'{args.episodes} episodes of {args.geo} + {args.app} iterations, {args.batch}-ray batches, {W}x{H} panorama, {args.dtype}',
                  'psnr_min_max': [min(ps), max(ps)], 'seconds_min_max': [min(r['seconds'] for r in rows), max(r['seconds'] for r in rows)],
                  'skipped_for_overflow_total': rows[-1]['skipped_for_overflow'], 'skipped_for_truncation_total': rows[-1]['skipped_for_truncation'],
                  'mem_reserved_MB_first_last': [rows[0]['mem_reserved_MB'], rows[-1]['mem_reserved_MB']], 'episodes': rows})
if rank == 0:
    print(summary)
    if args.out:
        open(args.out, 'w').write(summary)
if world > 1 and not args.rccl_single_rank:
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
