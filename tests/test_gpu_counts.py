"""Device-side sample counts (capacity-sized launches + n_dev), the sync-free / hipGraph-captured variable-count paths
built on them (training step, BASELINE config 4 eval frames), and the boundary's re-entrancy.

Bar: a capacity-sized call with a device-side count must give BIT-IDENTICAL results on the live rows to the exact-size
call; a graph replay must equal the eager step bit for bit; a graphed 512x1024 frame must equal the eager render."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import perf_oracle as O  # noqa: E402

AABB = [-1., -1, -1, 1, 1, 1]


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    from perf_amd import ops as _ops
    return _ops


def _cfg(**kw):
    from perf_amd.grid import GridConfig
    return GridConfig(**kw)


def _nd(n):
    return torch.tensor([n], dtype=torch.int64, device='cuda')


def _ray_points(n, g):
    R = n // 64 + 1
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    t = (torch.arange(64) + 0.5) / 64
    return ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.45 + 0.5)[:n].contiguous()


@pytest.mark.parametrize('n_live,cap', [(5000, 8192), (4099, 4099), (0, 1024), (33, 70000)])
def test_per_sample_ops_with_device_counts_equal_exact_calls(ops, n_live, cap):
    """hashgrid_fwd / mlp_fwd / mlp_bwd / hashgrid_bwd (fp32 and fixed point) / points_from_rays: capacity `cap` with
    n_dev = n_live  ==  exact call on the first n_live rows (rows beyond n_live hold garbage on purpose)."""
    from perf_amd.grid import MlpConfig
    cfg = _cfg()
    g = torch.Generator().manual_seed(7 + n_live)
    x_live = _ray_points(max(n_live, 1), g)[:n_live]
    x = torch.full((cap, 3), float('nan'))
    x[:n_live] = x_live
    x = x.cuda()
    spec = O.geo_spec()
    params = O.init_field_params(spec); params[spec.n_net:] *= 1e4
    w16 = ops.cast_params(params.cuda(), 'bf16')
    nd = _nd(n_live)
    mlp = MlpConfig(n_levels=16, n_hidden_layers=1, n_output_dims=1, output_activation='Exponential')
    # ---- forward
    feat_c = ops.hashgrid_fwd(cfg, x, w16[spec.n_net:], n_dev=nd)
    feat_e = ops.hashgrid_fwd(cfg, x[:n_live].contiguous(), w16[spec.n_net:])
    assert torch.equal(feat_c[:, :n_live], feat_e)
    sel = (torch.rand(cap, generator=g) > 0.1).to(torch.uint8).cuda()
    out_c = ops.mlp_fwd(mlp, w16[:spec.n_net], feat_c, sel, n_dev=nd)
    out_e = ops.mlp_fwd(mlp, w16[:spec.n_net], feat_e, sel[:n_live].contiguous())
    assert torch.equal(out_c[:n_live], out_e)
    # ---- backward
    dout = torch.randn(cap, 1, generator=g).cuda()
    dfc, dwc, amc = ops.mlp_bwd(mlp, w16[:spec.n_net], feat_c, dout, sel, want_absmax=True, n_dev=nd)
    dfe, dwe, ame = ops.mlp_bwd(mlp, w16[:spec.n_net], feat_e, dout[:n_live].contiguous(), sel[:n_live].contiguous(), want_absmax=True)
    assert torch.equal(dfc[:, :n_live], dfe) and torch.equal(amc, ame)
    # (the weight gradient is a two-stage sum whose partial count follows the capacity: equal up to fp32 association)
    assert float((dwc - dwe).abs().max()) <= 1e-5 * max(1.0, float(dwe.abs().max()))
    for absmax in (None, amc):
        gc = ops.hashgrid_bwd(cfg, x, dfc, level_absmax=absmax, n_dev=nd)
        ge = ops.hashgrid_bwd(cfg, x[:n_live].contiguous(), dfe, level_absmax=absmax)
        if absmax is None:
            assert float((gc - ge).abs().max()) <= 1e-5 * max(1e-12, float(ge.abs().max()))     # fp32 LDS atomics: order
        else:
            assert torch.equal(gc, ge)                  # fixed point: exact integer sums, headroom from the LIVE count
    # ---- positions
    R = 64
    o = torch.zeros(R, 3).cuda(); d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).cuda()
    ri = torch.randint(0, R, (cap,), generator=g).cuda()
    ts = torch.rand(cap, generator=g).cuda(); te = ts + 0.01
    pc, sc_ = ops.points_from_rays(o, d, ri, ts, te, AABB, n_dev=nd)
    pe, se = ops.points_from_rays(o, d, ri[:n_live].contiguous(), ts[:n_live].contiguous(), te[:n_live].contiguous(), AABB)
    assert torch.equal(pc[:n_live], pe) and torch.equal(sc_[:n_live], se)


def _room_scene(dtype='fp16', h=64, w=128, batch=2048, train_steps=0):
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    torch.manual_seed(0)
    scene = NeRFScene(dtype=dtype)
    rays = gen_pano_rays(torch.eye(4), h, w)
    dist, rgb = synthetic.room(rays.d)
    pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
    scene.train_conf.pixel_loss_batch_size = batch
    scene.set_train(); scene.prepare_occupancy(pool); scene.nerf.reset_geo()
    if train_steps:
        opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
        for i in range(train_steps):
            scene.update_lr(opt, scene.train_conf.geo_optimizer, 0.05 + 0.15 * i / train_steps)
            scene.train_one_step_geo(opt, pool, progress=0.5)
    return scene, pool, rays, dist, rgb


def test_sync_free_sampling_equals_synced_sampling():
    """NeRFOCCRenderer with sample_capacity (device-side counts, no read-back) == the count -> read -> allocate path:
    same packed_info, same samples on the live rows, same per-ray outputs -- on a trained scene so that the visibility
    compaction really drops samples."""
    scene, pool, rays, dist, rgb = _room_scene(train_steps=60)
    scene.set_eval()
    o, d = rays.o.reshape(-1, 3)[:4096].contiguous(), rays.d.reshape(-1, 3)[:4096].contiguous()
    r = scene.renderer
    near, far = torch.zeros(4096, 1).cuda(), torch.ones(4096, 1).cuda()
    with torch.no_grad():
        ref = r.render(scene.nerf, scene.estimator, o, d, near, far)
        n = ref['ray_indices'].numel()
        marched_ref = int(scene.estimator.sampling_ex(o, d, near_plane=r.near_plane, far_plane=r.far_plane,
                                                      render_step_size=r.render_step_size).ray_indices.numel())
        assert 0 < n < marched_ref, 'the scene must be trained enough for early termination to drop samples'
        evaluated = {}
        for head in (None, 8, 3, 64, 1000):
            # head: two-phase early termination -- density on the first `head` samples of every ray, then on the rest of the
            # rays still alive.  Same kept samples, bit for bit; far fewer density evaluations on an opaque scene.
            r.sample_capacity = marched_ref + 1000
            r.head_samples = head
            got = r.render(scene.nerf, scene.estimator, o, d, near, far)
            r.sample_capacity = None
            assert int(got['n_samples_dev'].item()) == n, head
            evaluated[head] = int(got['n_marched_dev'].item())
            assert torch.equal(got['packed_info'], ref['packed_info']), head
            for k in ('ray_indices', 't_starts', 't_ends', 'weights', 'trans'):
                assert torch.equal(got[k][:n], ref[k]), (head, k)
            for k in ('rgb', 'distance', 'opacities'):
                assert torch.equal(got[k], ref[k]), (head, k)
        r.head_samples = 8
    assert evaluated[None] == evaluated[1000] == marched_ref            # one phase (or a head that covers every ray): everything
    assert n <= evaluated[3] <= evaluated[8] <= evaluated[64] <= marched_ref and evaluated[8] < marched_ref


def test_truncated_capacity_is_detected_and_rendered_again():
    """NeRFScene.render: a per-ray capacity that is too small truncates batches; the single read-back at the end of the
    render detects it, raises the capacity and renders again -- the result equals the synced render."""
    scene, pool, rays, dist, rgb = _room_scene(train_steps=30)
    ref = scene.render(rays, ['rgb', 'distance'], batch_size=2048, sync_free=False)
    scene._eval_spp_cap = 2                                       # certainly too small
    got = scene.render(rays, ['rgb', 'distance'], batch_size=2048)
    assert scene._eval_spp_cap > 2
    for k in ref:
        assert torch.equal(got[k], ref[k]), k


def test_graph_replay_of_the_variable_count_step_equals_eager():
    """The reference-faithful geometry and colour steps (marching, no-grad density pass, visibility compaction with a
    data-dependent sample count, both fields, losses, backward, Adam) captured as hipGraphs: N replays == N eager steps,
    bit for bit (same batches: the batch draw is replaced by a fixed slice; same random draws: injected)."""
    from perf_amd.scene import Rays
    results = {}
    for mode in ('eager', 'graph'):
        scene, pool, rays, dist, rgb = _room_scene(train_steps=40, batch=1024)
        g = torch.Generator(device='cuda'); g.manual_seed(5)
        B = 1024
        idx = torch.randint(0, len(pool), (B,), device='cuda', generator=g)
        pool.rand_ray_color_data = lambda bs, **kw: (Rays(pool.all_sup_rays.o[idx], pool.all_sup_rays.d[idx]), pool.all_sup_colors[idx],
                                                      pool.all_sup_distances[idx], pool.all_sup_normals[idx])
        rand = {'jitter': torch.rand(B, device='cuda', generator=g), 'noise': torch.rand(B, 1, device='cuda', generator=g),
                'bg': torch.rand(B, 3, device='cuda', generator=g)}
        scene.renderer.sample_capacity = B * 128
        scene.sample_counters.zero_()
        out = {}
        for kind in ('geo', 'app'):
            net = scene.nerf.geo_mlp if kind == 'geo' else scene.nerf.app_mlp
            opt = scene.make_optimizer(net, 0.0)
            step = scene.train_one_step_geo if kind == 'geo' else scene.train_one_step_app
            conf = scene.train_conf.geo_optimizer
            if mode == 'eager':
                for i in range(6):
                    scene.update_lr(opt, conf, 0.1)
                    step(opt, pool, progress=0.5, rand=rand)
            else:
                orig = step
                wrapped = lambda o_, p_, progress, **kw: orig(o_, p_, progress=progress, rand=rand)
                setattr(scene, 'train_one_step_geo' if kind == 'geo' else 'train_one_step_app', wrapped)
                scene.update_lr(opt, conf, 0.1)
                wrapped(opt, pool, progress=0.5)                       # step 1 eagerly (also the warm-up)
                replay = scene.make_graphed_step(kind, opt, pool, warmup=0)
                for i in range(5):
                    replay(scene.lr_at(conf, 0.1), 0.5)
            out[kind] = (net.params.detach().clone(), opt.exp_avg.clone(), int(opt.step_count))
        out['counters'] = scene.sample_counters.tolist()
        results[mode] = out
    for kind in ('geo', 'app'):
        pe, me, se = results['eager'][kind]; pg, mg, sg = results['graph'][kind]
        assert se == sg == 6
        assert torch.equal(pe, pg) and torch.equal(me, mg), kind
    ce, cg = results['eager']['counters'], results['graph']['counters']
    assert ce == cg and ce[2] == 12 and 0 < ce[1] < ce[0], (ce, cg)          # compaction dropped samples; 12 steps counted
    assert ce[3] <= 1024 * 128 and ce[4] == 0 and ce[5] == 0, ce             # largest batch; nothing skipped


def test_adam_gate_skips_a_batch_without_samples(ops):
    """perf_adam_step_dev with a zero gate leaves parameters, moments and (through perf_step_bookkeeping) the step count
    untouched -- the reference returns before optimizer.step() when a batch has no samples (nerf.py:204-206)."""
    n = 4096 + 7
    g = torch.Generator().manual_seed(1)
    p = torch.randn(n, generator=g).cuda(); m = torch.zeros(n).cuda(); v = torch.zeros(n).cuda()
    grad = torch.randn(n, generator=g).cuda()
    step = torch.zeros(1, dtype=torch.int32, device='cuda'); lr = torch.full((1,), 1e-2, device='cuda')
    counters = ops.step_counters('cuda')
    p0 = p.clone()
    for gate_val, want_step in ((0, 0), (17, 1), (0, 1)):
        gate = _nd(gate_val)
        before = p.clone()
        ops.step_bookkeeping(step, gate, counters, _nd(100), gate)
        ops.adam_step_dev(p, m, v, grad, step, lr, gate=gate)
        assert int(step.item()) == want_step
        assert torch.equal(p, before) == (gate_val == 0)
    ref = torch.nn.Parameter(p0.clone()); ref.grad = grad.clone()
    torch.optim.Adam([ref], lr=1e-2).step()
    assert float((p - ref.detach()).abs().max()) < 2e-6
    assert counters.tolist() == [300, 17, 3, 100, 0, 0, 0, 0]


def test_step_gate_skips_overflowed_and_truncated_batches(ops):
    """perf_step_bookkeeping decides on the DEVICE whether the optimizer step is taken: a raised fixed-point overflow flag (local,
    or the job-wide sum remote_flags[0]) or a batch that marched more samples than the capacity (locally, or on any rank:
    remote_flags[1]) closes the gate that perf_adam_step_dev reads -- a corrupted or truncated gradient is never applied; the
    events are counted and the local flag is consumed.  overflow_redone: the caller repaired the flagged gradient in place
    (fp32 redo launch) -- counted, but the step is TAKEN."""
    n = 1000
    p = torch.linspace(-1, 1, n).cuda(); m = torch.zeros(n).cuda(); v = torch.zeros(n).cuda()
    grad = torch.ones(n).cuda()
    step = torch.zeros(1, dtype=torch.int32, device='cuda'); lr = torch.full((1,), 1e-2, device='cuda')
    counters = ops.step_counters('cuda')
    eff = torch.zeros(1, dtype=torch.int64, device='cuda')
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    remote = torch.zeros(2, dtype=torch.float32, device='cuda')
    cases = [  # (local flag, remote overflow, remote truncated, marched, capacity, redone, taken)
        (0, 0.0, 0.0, 500, 1000, False, True), (1, 0.0, 0.0, 500, 1000, False, False), (0, 2.0, 0.0, 500, 1000, False, False),
        (0, 0.0, 0.0, 1001, 1000, False, False), (0, 0.0, 0.0, 1000, 1000, False, True), (0, 0.0, 0.0, 5000, 0, False, True),
        (0, 0.0, 1.0, 500, 1000, False, False), (1, 0.0, 0.0, 500, 1000, True, True), (1, 0.0, 0.0, 1001, 1000, True, False)]
    taken = 0
    for lf, ro, rt, marched, cap, redone, want in cases:
        flag.fill_(lf); remote[0] = ro; remote[1] = rt
        before = p.clone()
        ops.step_bookkeeping(step, _nd(10), counters, _nd(marched), _nd(10), capacity=cap, overflow=flag, remote_flags=remote, eff_gate=eff,
                             overflow_redone=redone)
        ops.adam_step_dev(p, m, v, grad, step, lr, gate=eff)
        taken += int(want)
        assert int(eff.item()) == int(want) and int(step.item()) == taken and int(flag.item()) == 0
        assert torch.equal(p, before) == (not want)
    c = counters.tolist()
    assert c[2] == len(cases) and c[3] == 5000 and c[4] == 4 and c[5] == 3, c


def test_an_overflowed_fixed_point_gradient_is_repaired_not_dropped(ops):
    """Never drop a step: with the headroom forced down to its floor the packed fixed-point fields of the grid gradient
    overflow; the predicated repair launch behind the backward (perf_hashgrid_bwd, redo_flag) rewrites the table gradient
    with fp32 LDS accumulation -- equal to the fp32-mode gradient up to the order of fp32 atomics -- and is a no-op when the
    flag is clear (the fixed-point table stays, bit for bit).  End to end: the flagged step is TAKEN and counted, eagerly and
    as a graph replay."""
    from perf_amd.grid import MlpConfig
    cfg = _cfg()
    g = torch.Generator().manual_seed(11)
    n = 60000
    x = _ray_points(n, g).cuda()
    spec = O.geo_spec()
    params = O.init_field_params(spec); params[spec.n_net:] *= 1e4
    w16 = ops.cast_params(params.cuda(), 'bf16')
    mlp = MlpConfig(n_levels=16, n_hidden_layers=1, n_output_dims=1, output_activation='Exponential')
    feat = ops.hashgrid_fwd(cfg, x, w16[spec.n_net:])
    dout = (torch.rand(n, 1, generator=g) + 0.5).cuda()        # one sign: an entry's contributions add up coherently
    dfeat, dw, amax = ops.mlp_bwd(mlp, w16[:spec.n_net], feat, dout, want_absmax=True)
    flag = ops.overflow_flag(x.device); flag.zero_()
    ref32 = ops.hashgrid_bwd(cfg, x, dfeat)                                   # fp32 accumulation
    hr = ops.headroom_state(x.device)
    fixed = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax, hr_state=hr)
    assert int(flag.item()) == 0
    kept = fixed.clone()
    ops.hashgrid_bwd_redo(cfg, x, dfeat, kept, hr_state=hr)                   # flag clear: nothing happens
    assert torch.equal(kept, fixed) and int(hr[2 * 24 + 1].item()) == 0
    hr2 = ops.headroom_state(x.device); hr2[:24] = -24                        # headroom at its floor of 4 bits
    broken = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax, hr_state=hr2)
    assert int(flag.item()) == 1, 'the forced overflow did not raise the flag'
    ops.hashgrid_bwd_redo(cfg, x, dfeat, broken, hr_state=hr2)
    assert float((broken - ref32).abs().max()) <= 1e-5 * float(ref32.abs().max())
    assert int(hr2[2 * 24 + 1].item()) == 1                                   # the repair ran once
    flag.zero_()
    # ---- end to end: a geometry step whose gradient overflows is taken (eager, then as a graph replay)
    results = {}
    for accum in ('fixed', 'fp32'):
        scene, pool, rays, dist, rgb = _room_scene(dtype='bf16', batch=1024, train_steps=3)
        geo = scene.nerf.geo_mlp
        geo.grid_grad_accum = accum
        scene.renderer.sample_capacity = 1024 * 128
        scene.sample_counters.zero_()
        opt = scene.make_optimizer(geo, 1e-3)
        torch.manual_seed(3)
        if accum == 'fixed':
            geo.headroom_state()[:24] = -24
        scene.update_lr(opt, scene.train_conf.geo_optimizer, 0.1)
        scene.train_one_step_geo(opt, pool, progress=0.5)
        results[accum] = (geo.params.detach().clone(), opt.step_count, scene.sample_counters.tolist())
        lr = scene.lr_at(scene.train_conf.geo_optimizer, 0.1)
        if accum == 'fixed':
            replay = scene.make_graphed_step('geo', opt, pool, warmup=0)
            geo.headroom_state()[:24] = -24
            before = geo.params.detach().clone()
            replay(1e-3, 0.5)
            torch.cuda.synchronize()
            c = scene.sample_counters.tolist()
            assert opt.step_count == 2 and c[4] == 2 and not torch.equal(before, geo.params.detach()), (opt.step_count, c)
    pf, sf, cf = results['fixed']; p3, s3, c3 = results['fp32']
    assert sf == s3 == 1 and cf[4] == 1 and c3[4] == 0, (sf, s3, cf, c3)       # taken AND counted
    # the repaired step equals the fp32-mode step (Adam's first step is lr * sign-like: compare where the gradient is not noise)
    assert float((pf - p3).abs().max()) <= 2.1 * lr and float((pf - p3).abs().mean()) <= 2e-3 * lr, (float((pf - p3).abs().max()), float((pf - p3).abs().mean()))


def test_bookkeeping_in_the_repair_launch_changes_nothing(ops):
    """perf_field_bwd_book: the step's bookkeeping done by ONE thread of the predicated repair launch (+ perf_adam_step_dev's
    clear_flag consuming the overflow flag) against the stand-alone perf_step_bookkeeping launch between backward and Adam: the same
    parameters bit for bit, the same step count, counters, device-side schedule position and learning rate -- over eager geometry and
    colour steps and graph replays; then steps whose fixed-point gradient is forced to overflow (repaired in fp32 -- arrival-order
    atomics, so compared to a tolerance --, taken, counted, the flag consumed); the captured step is one node shorter."""
    from perf_amd.scene import FusedAdam
    keep = FusedAdam.book_in_repair_launch
    out = {}
    try:
        for mode in (True, False):
            FusedAdam.book_in_repair_launch = mode
            scene, pool, rays, dist, rgb = _room_scene(dtype='bf16', batch=1024, train_steps=0)
            scene.count_graph_nodes = True
            geo, app = scene.nerf.geo_mlp, scene.nerf.app_mlp
            scene.renderer.sample_capacity = 1024 * 128
            scene.sample_counters.zero_()
            flag = ops.overflow_flag('cuda'); flag.zero_()
            torch.manual_seed(5)
            og, oa = scene.make_optimizer(geo, 1e-3), scene.make_optimizer(app, 1e-3)
            seen = []

            def pair():
                scene.update_lr(og, scene.train_conf.geo_optimizer, 0.1)
                scene.train_one_step_geo(og, pool, progress=0.5)
                seen.append(int(flag.item()))
                scene.update_lr(oa, scene.train_conf.app_optimizer, 0.1)
                scene.train_one_step_app(oa, pool, progress=0.5)
                seen.append(int(flag.item()))

            pair(); pair()
            lrs = [1e-3 * 0.99 ** k for k in range(16)]
            replay = scene.make_graphed_step('geo', og, pool, warmup=0, schedule=(lrs, [min(2 * 0.05 * k, 1.0) for k in range(16)], 0))
            for k in range(3):
                replay()
            torch.cuda.synchronize()
            exact = dict(geo=geo.params.detach().clone(), app=app.params.detach().clone(), steps=(og.step_count, oa.step_count),
                         counters=scene.sample_counters.tolist(), it=int(og.sched_iter.item()), lr=float(og.lr_dev.item()),
                         ratio=float(scene._ratio_dev.item()), hr=geo.headroom_state().clone(), m=og.exp_avg.clone())
            # ---- flagged steps: one replayed, one eager
            geo.headroom_state()[:24] = -24
            replay()
            seen.append(int(flag.item()))
            og.clear_schedule()
            geo.headroom_state()[:24] = -24
            pair()
            torch.cuda.synchronize()
            assert seen == [0] * len(seen), seen                          # consumed by every step, the flagged ones included
            out[mode] = dict(exact=exact, geo=geo.params.detach().clone(), steps=(og.step_count, oa.step_count),
                             counters=scene.sample_counters.tolist(), nodes=scene.graph_nodes['geo'],
                             tol=2.2 * (lrs[3] + scene.lr_at(scene.train_conf.geo_optimizer, 0.1)))
    finally:
        FusedAdam.book_in_repair_launch = keep
    a, b = out[True], out[False]
    ea, eb = a['exact'], b['exact']
    assert torch.equal(ea['geo'], eb['geo']) and torch.equal(ea['app'], eb['app']) and torch.equal(ea['m'], eb['m']) and torch.equal(ea['hr'], eb['hr'])
    assert ea['steps'] == eb['steps'] == (5, 2) and ea['counters'] == eb['counters'] and ea['counters'][4] == 0, (ea['steps'], ea['counters'], eb['counters'])
    assert ea['it'] == eb['it'] == 3 and ea['lr'] == eb['lr'] == pytest.approx(lrs[2]) and ea['ratio'] == eb['ratio']
    assert a['steps'] == b['steps'] == (7, 3) and a['counters'][4] == b['counters'][4] == 2 and a['counters'][2] == b['counters'][2], (a['steps'], a['counters'], b['counters'])
    assert float((a['geo'] - b['geo']).abs().max()) <= a['tol'], (float((a['geo'] - b['geo']).abs().max()), a['tol'])   # (two Adam steps on arrival-order sums)
    assert a['nodes'] == b['nodes'] - 1, (a['nodes'], b['nodes'])


def test_dp_slots_carry_the_statistics_exactly(ops):
    """perf_dp_slot_pack / perf_dp_slot_unpack: integers travel as 16-bit pieces in the fp32 all-reduce buffer and come out
    exactly; the summed slots equal what an all-gather of perf_dp_stats_pack blocks would hold; the job flags add up."""
    from perf_amd import _lib
    world = 3
    dev = 'cuda'
    slots = [torch.empty(world * _lib.DP_SLOT, device=dev) for _ in range(world)]
    amax = [torch.rand(24, device=dev) * 10 ** (r - 1) for r in range(world)]
    fmax = [torch.randint(0, 2 ** 31 - 1, (24,), device=dev, dtype=torch.int32) for r in range(world)]
    fmax[1][3] = 2 ** 31 - 1; fmax[2][5] = 65535; fmax[0][7] = 65536
    lives = [123456789012, 0, 77]
    flags = [0, 1, 0]; marched = [10, 10, 5000]
    for r in range(world):
        fl = torch.tensor([flags[r]], dtype=torch.int32, device=dev)
        ops.dp_slot_pack(amax[r], fmax[r], _nd(lives[r]), 10 ** 13, fl, _nd(marched[r]), 1000, r, world, slots[r])
    summed = sum(slots)                                                       # what the SUM all-reduce leaves on every rank
    stats = torch.zeros(world * _lib.DP_STATS, dtype=torch.int32, device=dev)
    job = torch.zeros(2, device=dev); total = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.dp_slot_unpack(summed, world, stats, job, total)
    st = stats.view(world, -1)
    for r in range(world):
        assert torch.equal(st[r, :24], amax[r].view(torch.int32)) and torch.equal(st[r, 24:48], fmax[r])
        ref = ops.dp_stats_pack(amax[r], fmax[r], _nd(lives[r]), 10 ** 13)
        assert torch.equal(st[r, :50], ref[:50])
    assert int(total.item()) == sum(lives) and job.tolist() == [1.0, 1.0]


def test_autograd_path_runs_beyond_the_health_poll_with_a_capacity_set():
    """fused_steps = False (the autograd formulation of the training step, taken e.g. with density_loss_weight > 0) with
    renderer.sample_capacity set: the health poll after OVERFLOW_CHECK_EVERY steps used to raise a NameError."""
    from perf_amd import scene as S
    scene, pool, rays, dist, rgb = _room_scene(batch=256)
    scene.fused_steps = False
    scene.renderer.sample_capacity = 256 * 128
    opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
    before = scene.nerf.geo_mlp.params.detach().clone()
    for i in range(S.OVERFLOW_CHECK_EVERY + 2):
        scene.update_lr(opt, scene.train_conf.geo_optimizer, 0.1)
        scene.train_one_step_geo(opt, pool, progress=0.5)
    assert opt.step_count == S.OVERFLOW_CHECK_EVERY + 2
    assert not torch.equal(before, scene.nerf.geo_mlp.params.detach())


def test_hashgrid_bwd_is_reentrant(ops):
    """Two host threads call perf_hashgrid_bwd (+ perf_mlp_bwd) concurrently on two streams, many times; every result equals
    the serial one bit for bit (fixed-point accumulation is order independent), and each thread sees its own
    perf_last_error()."""
    from perf_amd import _lib
    from perf_amd.grid import MlpConfig
    cfg = _cfg()
    mlp = MlpConfig(n_levels=16, n_hidden_layers=1, n_output_dims=1, output_activation='Exponential')
    g = torch.Generator().manual_seed(3)
    spec = O.geo_spec()
    params = O.init_field_params(spec); params[spec.n_net:] *= 1e4
    w16 = ops.cast_params(params.cuda(), 'bf16')
    jobs = []
    for n in (30011, 65536):
        x = _ray_points(n, g).cuda()
        feat = ops.hashgrid_fwd(cfg, x, w16[spec.n_net:])
        dout = torch.randn(n, 1, generator=g).cuda()
        dfeat, dw, amax = ops.mlp_bwd(mlp, w16[:spec.n_net], feat, dout, want_absmax=True)
        jobs.append((x, feat, dout, ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax), dw))
    torch.cuda.synchronize()
    errors = []

    def worker(k):
        try:
            x, feat, dout, ref_g, ref_w = jobs[k]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(25):
                    dfeat, dw, amax = ops.mlp_bwd(mlp, w16[:spec.n_net], feat, dout, want_absmax=True)
                    got = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax)
                    if not (torch.equal(got, ref_g) and torch.equal(dw, ref_w)):
                        errors.append((k, it, float((got - ref_g).abs().max())))
                        break
                # an invalid call in THIS thread must not leak its message into the other thread's error slot
                bad = _lib.GridDesc(); bad.n_levels = 99 if k == 0 else 0
                rc = _lib.load().perf_hashgrid_bwd_workspace_bytes(__import__('ctypes').byref(bad), 16)
                msg = _lib.load().perf_last_error().decode()
                if rc != -1 or (('99' in msg) != (k == 0)):
                    errors.append((k, 'error slot', msg))
            st.synchronize()
        except Exception as e:           # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_gather_supervision_direct(ops):
    """SupInfoPool.rand_ray_color_data's gather (sup_info.py:256-259) against torch indexing, incl. repeated indices and
    skipped (NULL) sources."""
    g = torch.Generator().manual_seed(11)
    n_pool, n = 5003, 8192
    src = {k: torch.randn(n_pool, w, generator=g).cuda() for k, w in (('o', 3), ('d', 3), ('color', 3), ('dist', 1), ('normal', 3))}
    idx = torch.randint(0, n_pool, (n,), generator=g).cuda()
    idx[:10] = idx[10:20]
    out = ops.gather_supervision(idx, src['o'], src['d'], src['color'], src['dist'], src['normal'])
    for k in src:
        assert torch.equal(out[k], src[k][idx]), k
    part = ops.gather_supervision(idx, o_all=src['o'], dist_all=src['dist'])
    assert part['d'] is None and part['color'] is None and torch.equal(part['dist'], src['dist'][idx])
    empty = ops.gather_supervision(idx[:0], src['o'], src['d'], src['color'], src['dist'], src['normal'])
    assert empty['o'].shape == (0, 3)


def test_contract_to_unisphere_matches_reference_golden(golden_dir):
    """The `unbounded` contraction (ngp_nerf.py:43-65; never enabled by PeRF) on the GPU against the vector produced by the
    reference's own function, and NGPDensityField(unbounded=True) runs through it."""
    from perf_amd.fields import NGPDensityField, contract_to_unisphere
    g = np.load(f'{golden_dir}/field_bits.npz')
    x = torch.from_numpy(g['ct_x']).cuda()
    y = contract_to_unisphere(x, torch.tensor(AABB).cuda()).cpu().numpy()
    assert np.abs(y - g['ct_y']).max() <= 1e-6
    f = NGPDensityField(AABB, unbounded=True, dtype='fp16')
    out = f(x * 3.0)
    assert out.shape == (x.shape[0], 1) and torch.isfinite(out).all() and float(out.min()) >= 0.0


def test_march_per_ray_span_check_handles_unnormalised_directions(ops):
    """The coarse empty-space skip assumes a 64-interval chunk spans few cells; a direction of length 4 breaks that, and the
    kernel must then take the exhaustive test for that ray: results equal the oracle (which has no skip)."""
    from perf_amd.nerfacc_impl import OccGridEstimator
    g = torch.Generator().manual_seed(2)
    res = 64
    occ = (torch.rand(res, res, res, generator=g) < 0.02).numpy()
    est = OccGridEstimator(AABB, resolution=res).cuda()
    est.set_binaries(torch.from_numpy(occ.reshape(-1)).cuda())
    R = 257
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    d[::3] *= 4.0                                                   # unnormalised
    o = (torch.rand(R, 3, generator=g) - 0.5) * 0.5
    step = 2e-3
    sm = est.sampling_ex(o.cuda(), d.cuda(), near_plane=0.0, far_plane=1.0, render_step_size=step, early_stop_eps=0.0)
    ri, ts, te, packed = O.occ_march(o.numpy(), d.numpy(), occ, np.asarray(AABB, np.float32), 0.0, 1.0, step, None, None)
    assert np.array_equal(sm.ray_indices.cpu().numpy(), ri) and np.array_equal(sm.t_starts.cpu().numpy(), ts)


@pytest.mark.parametrize('head', [None, 4])
def test_reusing_the_sampling_features_changes_nothing(head):
    """NeRFScene.reuse_sampling_features: the gradient pass of the density field starts from the features the sampler's own
    density pass encoded (compacted with the samples) instead of encoding the kept samples again -- same parameters, same
    positions, so 5 training steps must leave bit-identical parameters and optimizer state."""
    from perf_amd.scene import Rays
    out = {}
    for reuse in (False, True):
        scene, pool, rays, dist, rgb = _room_scene(train_steps=30, batch=1024)
        g = torch.Generator(device='cuda'); g.manual_seed(8)
        B = 1024
        idx = torch.randint(0, len(pool), (B,), device='cuda', generator=g)
        pool.rand_ray_color_data = lambda bs, **kw: (Rays(pool.all_sup_rays.o[idx], pool.all_sup_rays.d[idx]), pool.all_sup_colors[idx],
                                                      pool.all_sup_distances[idx], pool.all_sup_normals[idx])
        rand = {'jitter': torch.rand(B, device='cuda', generator=g), 'noise': torch.rand(B, 1, device='cuda', generator=g)}
        scene.renderer.sample_capacity = B * 128
        scene.renderer.head_samples = head
        scene.reuse_sampling_features = reuse
        opt = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
        for i in range(5):
            scene.update_lr(opt, scene.train_conf.geo_optimizer, 0.1)
            scene.train_one_step_geo(opt, pool, progress=0.5, rand=rand)
        out[reuse] = (scene.nerf.geo_mlp.params.detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone())
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)


def test_device_batch_draw(ops):
    """perf_draw_train_batch: one launch = uniform batch indices + the gathered supervision rows + the per-ray uniforms, from a
    counter-based generator.  Deterministic in (seed, counter); the counter advances by one per launch; two ranks' slices are
    the halves of the single-process batch; draws are uniform."""
    g = torch.Generator().manual_seed(11)
    n_pool, B = 100000, 8192
    o = torch.randn(n_pool, 3, generator=g).cuda(); d = torch.randn(n_pool, 3, generator=g).cuda()
    c = torch.rand(n_pool, 3, generator=g).cuda(); t = torch.rand(n_pool, 1, generator=g).cuda(); nrm = torch.randn(n_pool, 3, generator=g).cuda()
    counter = torch.zeros(1, dtype=torch.int64, device='cuda')
    a = ops.draw_train_batch(1234, counter, 0, n_pool, B, 0, o, d, c, t, nrm, want_bg=True, want_indices=True)
    assert int(counter.item()) == 1
    idx = a['indices']
    assert int(idx.min()) >= 0 and int(idx.max()) < n_pool
    assert torch.equal(a['o'], o[idx]) and torch.equal(a['d'], d[idx]) and torch.equal(a['color'], c[idx])
    assert torch.equal(a['dist'], t[idx]) and torch.equal(a['normal'], nrm[idx])
    for k in ('jitter', 'noise', 'bg'):
        u = a[k].float()
        assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 0.02, k
    assert abs(float(idx.float().mean()) / n_pool - 0.5) < 0.02 and idx.unique().numel() > 0.9 * B * (1 - B / (2 * n_pool))
    # same (seed, counter) -> same draw; the next counter -> another one
    counter.zero_()
    b = ops.draw_train_batch(1234, counter, 0, n_pool, B, 0, o, d, c, t, nrm, want_bg=True, want_indices=True)
    assert all(torch.equal(a[k], b[k]) for k in ('indices', 'jitter', 'noise', 'bg'))
    nxt = ops.draw_train_batch(1234, counter, 0, n_pool, B, 0, o, d, c, t, nrm, want_bg=True, want_indices=True)
    assert not torch.equal(nxt['indices'], a['indices']) and int(counter.item()) == 2
    other_seed = ops.draw_train_batch(99, torch.zeros(1, dtype=torch.int64, device='cuda'), 0, n_pool, B, 0, o, d, c, t, nrm, want_indices=True)
    assert not torch.equal(other_seed['indices'], a['indices'])
    # two ranks: slices of the same global batch
    halves = []
    for rank in range(2):
        cnt = torch.zeros(1, dtype=torch.int64, device='cuda')
        halves.append(ops.draw_train_batch(1234, cnt, 0, n_pool, B // 2, rank * (B // 2), o, d, c, t, nrm, want_bg=True, want_indices=True))
    for k in ('indices', 'jitter', 'noise', 'bg', 'o'):
        assert torch.equal(torch.cat([halves[0][k], halves[1][k]]), a[k]), k
    # a sub-range of the pool (rand_mode 'only_last')
    r = ops.draw_train_batch(5, torch.zeros(1, dtype=torch.int64, device='cuda'), 70000, n_pool, 4096, 0, o, d, c, t, want_indices=True)
    assert int(r['indices'].min()) >= 70000 and int(r['indices'].max()) < n_pool and r['normal'] is None


@pytest.mark.parametrize('head', [2, None])
def test_device_rng_episode_graph_replay_equals_eager(head):
    """A short episode with the device generator (the default): graph-replayed steps == eager steps, bit for bit (same seed,
    same counter sequence), and the captured step holds no torch random op -- with the two-phase sampler (head = 2: the
    counting pass writes the heads) and with the one-phase sampler bench.py uses."""
    res = {}
    for mode in ('eager', 'graph'):
        scene, pool, rays, dist, rgb = _room_scene(batch=1024)
        assert scene.device_rng
        scene.renderer.head_samples = head
        scene.graph_steps = (mode != 'eager')
        scene.train_one_episode(pool, 12, 8)
        assert scene._geo_pre is None
        res[mode] = (scene.nerf.geo_mlp.params.detach().clone(), scene.nerf.app_mlp.params.detach().clone(), int(scene._rng_counter.item()))
    assert res['eager'][2] == res['graph'][2] == 20
    for mode in ('graph',):
        assert torch.equal(res['eager'][0], res[mode][0]) and torch.equal(res['eager'][1], res[mode][1]), mode


def test_a_shrunk_capacity_does_not_outlive_the_density_field_it_was_measured_on():
    """ADVICE-5: the health poll lowers a sample capacity that proved far too large and the captured step is captured again;
    the lowered value belongs to THAT density field.  A caller that drives make_graphed_step itself (bench.py,
    tools/mini_perf_loop.py -- not train_one_episode, which restores the capacity on its own) and then starts a new phase on a
    fresh geometry network (reset_geo + make_optimizer: transparent again, marching several times more samples) must get the
    capacity back: no batch of the new phase is truncated, none of its steps skipped."""
    from perf_amd import scene as S
    scene, pool, rays, dist, rgb = _room_scene(h=128, w=256, batch=4096)
    scene.CAPACITY_MIN_ROWS = 8192                 # (the default floor of 65,536 rows is above this small batch's counts)
    opt = scene.make_optimizer(scene.nerf.geo_mlp, 1e-2)
    replay = scene.make_graphed_step('geo', opt, pool, warmup=3)
    cap0 = scene.renderer.sample_capacity
    assert cap0 == 4096 * scene.TRAIN_SAMPLES_PER_RAY
    first = None
    for i in range(10 * S.OVERFLOW_CHECK_EVERY):
        replay(1e-2, 0.5)
        if i == 0:
            first = int(scene._last_counts[0].item())
    late = int(replay.state['counts'][0].item())
    shrunk = scene.renderer.sample_capacity
    print(f'[capacity] marched: first step {first}, late {late}; capacity {cap0} -> {shrunk}')
    assert shrunk < cap0 and scene._capacity_unshrunk == cap0, 'the field must have formed far enough for the poll to shrink'
    assert first > shrunk, 'a fresh field must march more than the shrunk capacity holds, or this test shows nothing'
    c = scene.sample_counters.tolist()
    assert c[5] == 0
    # -- a new phase on a fresh geometry network, driven by hand
    scene.nerf.reset_geo()
    opt = scene.make_optimizer(scene.nerf.geo_mlp, 1e-2)
    assert scene.renderer.sample_capacity == cap0 and scene._capacity_unshrunk is None
    replay = scene.make_graphed_step('geo', opt, pool, warmup=3)
    for i in range(2 * S.OVERFLOW_CHECK_EVERY):
        replay(1e-2, 0.5)
    scene._poll_health(force=True)
    c2 = scene.sample_counters.tolist()
    assert c2[5] == 0, f'{c2[5]} steps of the new phase were skipped for truncation'
    assert c2[2] - c[2] == 3 + 2 * S.OVERFLOW_CHECK_EVERY          # every step of the new phase was taken
