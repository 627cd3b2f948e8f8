"""Worker of tests/test_gpu_psnr.py::test_data_parallel_psnr_at_iter_matches_the_oracle_curve: WORLD_SIZE ranks share one GPU
(gloo) and train the schedule of tests/golden/psnr_curve.json through the DATA-PARALLEL path with its DEFAULT exchange
(perf_amd/dp.py: sharded, lagged units -- not bit-identical to the single process, so its quality claim needs its own
test): every rank takes its slice of each golden batch and of the batch's random draws, losses are normalised by the global
batch, the int32 reduce-scatter / small all-reduce / all-gather of the 16-bit copy run every step.  Rank 0 writes
{seed: curve} with the marks psnr_parity_lib.run_hip reports.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/psnr_dp_worker.py <out.json> <seed> [<seed> ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def run_dp(scene_data, geo0, app0, draws, n_geo, n_app, marks, dtype, rank, world):
    from perf_amd.scene import NeRFScene, Rays, SupInfoPool
    from tests import psnr_parity_lib as P
    o, d, dist_map, rgb, occ = scene_data
    batch = draws[0]['idx'].numel()
    per = batch // world
    sc = NeRFScene(dtype=dtype)
    assert sc.dp_mode == 'sharded' and sc.dp_units == 'lagged', (sc.dp_mode, sc.dp_units)      # the defaults are what is tested
    pool = SupInfoPool(); pool.register_rays(o.cuda(), d.cuda(), rgb.cuda(), dist_map.cuda())
    sc.train_conf.pixel_loss_batch_size = batch
    sc.set_train()
    sc.estimator.set_binaries(torch.from_numpy(occ.reshape(-1)).cuda())
    sc.nerf.reset_geo()
    with torch.no_grad():
        sc.nerf.geo_mlp.params.copy_(geo0.cuda()); sc.nerf.app_mlp.params.copy_(app0.cuda())
    state = {'idx': None}

    def draw(bs, rank=0, world_size=1, **kw):                        # this rank's slice of the golden batch
        idx = state['idx'][rank * (bs // world_size):(rank + 1) * (bs // world_size)]
        return (Rays(pool.all_sup_rays.o[idx], pool.all_sup_rays.d[idx]), pool.all_sup_colors[idx], pool.all_sup_distances[idx],
                pool.all_sup_normals[idx])
    pool.rand_ray_color_data = draw
    cut = lambda dr: {k: dr[k][rank * per:(rank + 1) * per].cuda().contiguous() for k in ('jitter', 'bg', 'noise')}
    rays = Rays(o.cuda(), d.cuda())
    curve = {}
    opt = sc.make_optimizer(sc.nerf.geo_mlp, 0.0)
    conf = sc.train_conf.geo_optimizer
    dls = []
    skipped, seen = [], 0

    diag = os.environ.get('PERF_DP_DIAG') == '1' and rank == 0
    hist = []

    def note(tag):                                                   # (one read-back per step: nothing next to gloo's host copies)
        nonlocal seen
        now = int(sc.sample_counters[4].item())
        if diag:                                                     # what the exchange saw in this step (tools/exp/dp_margin_sweep.py)
            net = sc.nerf.geo_mlp if tag.startswith('geo') else sc.nerf.app_mlp
            ex = getattr(net, '_dp_exchange', None)
            if ex is not None:
                hist.append((tag, ex.level_absmax[:16].tolist(), ex.shifts[:16].tolist(), ex.field_max[:16].tolist(), ex.job_flags.tolist(),
                             int(ex.n_total.item())))
                if now != seen and len(hist) >= 2:
                    for h in hist[-2:]:
                        print('DIAG', h[0], 'flags', h[4], 'n', h[5], '\n   absmax', ['%.2e' % v for v in h[1]], '\n   shifts', h[2], '\n   field_max(log2)',
                              [0 if v <= 0 else v.bit_length() for v in h[3]], flush=True)
        if now != seen:
            skipped.append(tag); seen = now
    for i in range(n_geo):
        dr = draws[i]; state['idx'] = dr['idx'].cuda()
        sc.update_lr(opt, conf, i / n_geo)
        # (no prefetch: the next batch is injected, not drawn)
        sc.train_one_step_geo(opt, pool, progress=i / n_app, rand=cut(dr), prefetch_next=False)
        dls.append(sc.last_losses['depth_loss'])
        note(f'geo{i}')
    # the loss head normalises by the GLOBAL batch: a rank's depth loss is its share of the job's
    dl = torch.stack(dls).double()
    dist.all_reduce(dl)
    dls = [float(v) for v in dl.cpu()]
    sc.sync_params()
    ev = sc.render(rays, ['rgb', 'distance', 'opacities'])
    curve['geo_end_depth_err'] = float((ev['distance'].cpu() - dist_map).abs().mean())
    curve['geo_end_opacity'] = float(ev['opacities'].mean())
    for k in P.GEO_MARKS:
        if k <= len(dls):
            curve[f'geo_depth_loss@{k}'] = float(np.mean(dls[k - 10:k]))
    sc.set_train()
    opt = sc.make_optimizer(sc.nerf.app_mlp, 0.0)
    for i in range(n_app):
        dr = draws[n_geo + i]; state['idx'] = dr['idx'].cuda()
        sc.update_lr(opt, conf, i / n_app)
        sc.train_one_step_app(opt, pool, progress=i / n_app, rand=cut(dr))
        note(f'app{i}')
        if (i + 1) in marks:
            curve[f'psnr@app{i + 1}'] = P.psnr(sc.render(rays, ['rgb'])['rgb'].cpu(), rgb); sc.set_train()
    c = sc.sample_counters.tolist()
    curve['skipped_for_overflow'], curve['skipped_for_truncation'] = int(c[4]), int(c[5])
    curve['collectives_per_step'] = 3
    curve['steps_skipped_for_overflow'] = skipped
    return curve


def main():
    out_path, seeds = sys.argv[1], [int(s) for s in sys.argv[2:]]
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0'))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    from perf_amd import tcnn
    from tests import psnr_parity_lib as P
    golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'psnr_curve.json')))
    cfg = golden['config']
    h, w = cfg['pano']
    scene_data = P.make_scene(h, w)
    res = {}
    for row in golden['seeds']:
        if row['seed'] not in seeds:
            continue
        sd = row['seed']
        geo0, app0 = P.init_params(sd)
        draws = P.make_draws(scene_data[0].shape[0], cfg['batch'], cfg['geo_iters'] + cfg['app_iters'], sd)
        assert P.draws_digest(draws) == row['draws_digest']
        res[str(sd)] = run_dp(scene_data, geo0, app0, draws, cfg['geo_iters'], cfg['app_iters'], tuple(cfg['marks']), tcnn.DEFAULT_DTYPE, rank, world)
        # the same seed from the initialisation moved by one fp32 ulp up / down (tests/test_gpu_psnr.py: a seed's statistic is the mean
        # of the three members)
        res[str(sd)]['one_ulp_members'] = []
        for towards in (float('inf'), -float('inf')):
            g = torch.nextafter(geo0, torch.full_like(geo0, towards)); a = torch.nextafter(app0, torch.full_like(app0, towards))
            c = run_dp(scene_data, g, a, draws, cfg['geo_iters'], cfg['app_iters'], tuple(cfg['marks']), tcnn.DEFAULT_DTYPE, rank, world)
            res[str(sd)]['one_ulp_members'].append({k: c[k] for k in c if k.startswith('psnr') or k.startswith('skipped')})
    if rank == 0:
        json.dump({'world': world, 'curves': res}, open(out_path, 'w'))
    dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    main()
