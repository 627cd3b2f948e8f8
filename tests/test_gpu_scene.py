"""GPU parity of the host-side mirror (fields / renderer / training steps) against the CPU oracle, and the
drop-in shim packages.  Fields are compared with the oracle's 16-bit emulation (quant=...), which rounds the
same operands the kernels round."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import perf_oracle as O  # noqa: E402

AABB = [-1., -1, -1, 1, 1, 1]


def _params(gain=1e4):
    gs, as_ = O.geo_spec(), O.app_spec()
    geo = O.init_field_params(gs, 1337); app = O.init_field_params(as_, 4242)
    geo[gs.n_net:] *= gain; app[as_.n_net:] *= gain
    return geo, app


def _nerf(dtype, geo, app):
    from perf_amd.fields import NGPNeRF
    nerf = NGPNeRF(aabb=AABB, dtype=dtype)
    with torch.no_grad():
        nerf.geo_mlp.params.copy_(geo.cuda()); nerf.app_mlp.params.copy_(app.cuda())
    return nerf


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_field_queries(dtype):
    geo, app = _params()
    nerf = _nerf(dtype, geo, app)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2000, 3, generator=g) * 2.2 - 1.1            # some points outside the aabb (selector = 0)
    aabb = torch.tensor(AABB)
    sig = nerf.query_density(x.cuda()).cpu()
    rgb = nerf.query_rgb(x.cuda()).cpu()
    rs = O.query_density(x, geo, O.geo_spec(), aabb, quant=dtype)
    rr = O.query_rgb(x, app, O.app_spec(), aabb, quant=dtype)
    ulp = 2.0 ** -8 if dtype == 'bf16' else 2.0 ** -11
    # sigma = exp(logit): relative error = absolute logit error (a few 16-bit ulps of |logit| <~ 8)
    assert ((sig - rs).abs() <= 0.25 * (64 * ulp) * rs.abs() + 1e-6).all()
    assert (rgb - rr).abs().max() < 32 * ulp
    assert float(sig[(x.abs() > 1).any(-1)].abs().max()) == 0.0
    # state_dict keys are what PeRF checkpoints hold (nerf.py:374-380)
    assert set(nerf.state_dict().keys()) == {'aabb', 'geo_mlp.params', 'app_mlp.params'}
    assert nerf.geo_mlp.params.numel() == 6644288 and nerf.app_mlp.params.numel() == 6648384


BAND = {'fp16': 4e-3, 'bf16': 2e-2}


def _kept_counts_checked(hip_packed, pre, band=2e-2):
    """Per-ray kept counts of the HIP path, after checking that wherever they differ from the oracle's the disputed
    samples sit within `band` (relative) of the early-stop threshold on the oracle's own canonical scan.  The band is NOT
    about the rounding of exp (the threshold is applied to the scan itself): the scan sums sigma * delta, and the two sides
    evaluate sigma = exp(logit) from 16-bit fields whose roundings differ by a few ulps of the storage type on a logit of
    magnitude <~ 8 -- relative 8 x 2^-11 = 4e-3 (fp16), 8 x 2^-8 = 3e-2 (bf16); BAND holds what is asserted per type.  Returns
    the counts to hand to O.occ_render(kept_counts=...), so that everything else is compared on the SAME sample set."""
    hc = hip_packed[:, 1].astype(np.int64)
    packed = pre['packed_info']
    keep = pre['keep']
    oc = np.array([int(keep[s0:s0 + c].sum()) for s0, c in packed], np.int64)
    assert (hc <= packed[:, 1]).all()
    thr = float(pre['threshold'])
    for r in np.nonzero(hc != oc)[0]:
        lo, hi = int(min(hc[r], oc[r])), int(max(hc[r], oc[r]))
        ex = pre['exsum'][packed[r, 0] + lo:packed[r, 0] + hi]
        assert (np.abs(ex - thr) <= band * thr).all(), (int(r), lo, hi, ex, thr)
    assert (hc != oc).sum() <= max(2, len(hc) // 50), 'too many rays disagree on the early-stop sample'
    return hc


def _glue_setup(golden_dir):
    g = np.load(f'{golden_dir}/render_glue.npz')
    res = int(g['res'])
    occ = np.unpackbits(g['binaries'])[:res ** 3].reshape(res, res, res).astype(bool)
    return g, res, occ


# absolute tolerance of O(1) renderer outputs / relative tolerances (L2, max-abs per grid level) of gradients per 16-bit type.
# The oracle's quant=... emulation rounds the same 16-bit operands the kernels round, so what is left is accumulation order:
# measured on MI355X (round 3) -- outputs <= 8e-6 for both types; gradients fp16 2.0e-3 / 7e-4, bf16 4.6e-3 / 6.1e-3.
OUT_TOL = {'fp16': 5e-5, 'bf16': 1e-4}
GRAD_TOL = {'fp16': (5e-3, 5e-3), 'bf16': (1.5e-2, 2e-2)}


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_renderer_matches_oracle(golden_dir, mode, dtype):
    """NeRFOCCRenderer mirror on HIP == oracle.occ_render (itself pinned on the reference's renderer glue)."""
    from perf_amd.nerfacc_impl import OccGridEstimator
    from perf_amd.renderer import NeRFOCCRenderer
    g, res, occ = _glue_setup(golden_dir)
    geo, app = _params(float(g['grid_gain']))
    nerf = _nerf(dtype, geo, app)
    nerf.train(mode == 'train')
    est = OccGridEstimator(AABB, resolution=res).cuda()
    est.set_binaries(torch.from_numpy(occ.reshape(-1)).cuda())
    rend = NeRFOCCRenderer(max_radius=2, bg_color='rand_noise')
    rend.render_step_size = float(g['step'])
    o = torch.from_numpy(g['o']); d = torch.from_numpy(g['d'])
    R = o.shape[0]
    rand = {'jitter': torch.from_numpy(g[f'{mode}_jitter']).cuda(), 'bg': torch.from_numpy(g[f'{mode}_bg']).cuda(),
            'noise': torch.from_numpy(g[f'{mode}_noise']).cuda()}
    out = rend.render(nerf, est, o.cuda(), d.cuda(), torch.zeros(R, 1).cuda(), torch.ones(R, 1).cuda(),
                      geo_inference=False, app_inference=True, rand=rand)
    t0 = (np.zeros(R, np.float32) + g[f'{mode}_jitter'] * np.float32(float(g['step']))).astype(np.float32) if mode == 'train' else None
    kw = dict(training=(mode == 'train'), t0=t0, bg_color=torch.from_numpy(g[f'{mode}_bg']),
              dist_noise=torch.from_numpy(g[f'{mode}_noise']), step=float(g['step']), quant=dtype)
    pre = O.occ_render(o, d, geo, app, occ, AABB, return_pre=True, **kw)['pre']
    # ray bookkeeping: bit-exact; a sample may only flip across the early-stop threshold if its scan value sits on it
    counts = _kept_counts_checked(out['packed_info'].cpu().numpy(), pre, BAND[dtype])
    ref = O.occ_render(o, d, geo, app, occ, AABB, kept_counts=counts, **kw)
    gri, rri = out['ray_indices'].cpu().numpy(), ref['ray_indices'].numpy()
    assert np.array_equal(gri, rri)
    assert np.array_equal(out['packed_info'].cpu().numpy(), ref['packed_info'])
    assert np.array_equal(out['t_starts'].cpu().numpy(), ref['t_starts'].numpy())
    assert np.array_equal(out['t_ends'].cpu().numpy(), ref['t_ends'].numpy())
    tol = OUT_TOL[dtype]
    errs = {k: float((out[k].detach().cpu() - ref[k].detach()).abs().max()) for k in ('weights', 'trans', 'rgb', 'distance', 'opacities')}
    print(f'[renderer {mode} {dtype}] max abs errors {errs}')
    for k, e in errs.items():
        assert e < tol, (k, e)      # 16-bit fields: a few ulps of the storage type on O(1) outputs


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_geo_step_gradient_matches_oracle(golden_dir, dtype):
    """One geometry training step: d loss / d geo params from the HIP path vs autograd through the oracle."""
    from perf_amd.nerfacc_impl import OccGridEstimator
    from perf_amd.renderer import NeRFOCCRenderer
    from perf_amd.distloss import flatten_eff_distloss
    g, res, occ = _glue_setup(golden_dir)
    geo, app = _params(float(g['grid_gain']))
    nerf = _nerf(dtype, geo, app)
    nerf.train()
    est = OccGridEstimator(AABB, resolution=res).cuda()
    est.set_binaries(torch.from_numpy(occ.reshape(-1)).cuda())
    rend = NeRFOCCRenderer(max_radius=2, bg_color='rand_noise')
    rend.render_step_size = float(g['step'])
    o = torch.from_numpy(g['o']); d = torch.from_numpy(g['d'])
    R = o.shape[0]
    gt_dist, _ = O.synthetic_room(d)
    rand = {k2: torch.from_numpy(g[f'train_{k}']).cuda() for k2, k in (('jitter', 'jitter'), ('bg', 'bg'), ('noise', 'noise'))}
    out = rend.render(nerf, est, o.cuda(), d.cuda(), torch.zeros(R, 1).cuda(), torch.ones(R, 1).cuda(), app_inference=True, rand=rand)
    dl = torch.nn.functional.smooth_l1_loss(out['distance'], gt_dist.cuda(), beta=1e-2, reduction='mean')
    distl = flatten_eff_distloss(out['weights'], (out['t_ends'] + out['t_starts']) * .5, out['t_ends'] - out['t_starts'],
                                 out['ray_indices'], packed_info=out['packed_info'])
    ((dl + distl * 0.1 * 0.5) * 128.0).backward()
    grad = nerf.geo_mlp.params.grad.cpu()
    assert nerf.app_mlp.params.grad is None
    geo_r = geo.clone().requires_grad_(True)
    t0 = (np.zeros(R, np.float32) + g['train_jitter'] * np.float32(float(g['step']))).astype(np.float32)
    kw = dict(training=True, t0=t0, bg_color=torch.from_numpy(g['train_bg']), dist_noise=torch.from_numpy(g['train_noise']),
              step=float(g['step']), quant=dtype)
    counts = _kept_counts_checked(out['packed_info'].cpu().numpy(), O.occ_render(o, d, geo, app, occ, AABB, return_pre=True, **kw)['pre'], BAND[dtype])
    ref = O.occ_render(o, d, geo_r, app, occ, AABB, kept_counts=counts, **kw)
    assert np.array_equal(out['ray_indices'].cpu().numpy(), ref['ray_indices'].numpy())
    loss, rdl, rdistl = O.geo_step_loss(ref, gt_dist, progress=0.25)
    loss.backward()
    k16 = 8.0 if dtype == 'bf16' else 1.0
    assert abs(float(dl) - float(rdl)) < 2e-3 * k16 * max(1.0, abs(float(rdl)))
    assert abs(float(distl) - float(rdistl)) < 2e-2 * k16 * max(1e-3, abs(float(rdistl)))
    worst = _assert_field_gradient_close(grad, geo_r.grad, O.geo_spec(), *GRAD_TOL[dtype])
    print(f'[geo gradient {dtype}] worst level (rel L2, max-abs, level): {max(worst)}')


def _assert_field_gradient_close(grad, ref, spec, tol_l2=3e-2, tol_max=4e-2):
    """Flat field gradient [network | grid] against the oracle's autograd: relative L2 of the network part, and PER GRID
    LEVEL both the relative L2 and the max-abs error (normalised by the level's largest reference entry) -- a whole-table
    L2 would hide an error confined to one level.  The HIP side multiplies 16-bit operands (features, weights,
    activations) where the oracle's quant=... emulation rounds the same operands, so a few 16-bit ulps per product."""
    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-12))
    n_net = spec.n_net
    assert rel(grad[:n_net], ref[:n_net]) < tol_l2, rel(grad[:n_net], ref[:n_net])
    lv = spec.lv
    worst = []
    for l in range(lv.n_levels):
        lo, hi = n_net + 2 * int(lv.offset[l]), n_net + 2 * int(lv.offset[l] + lv.size[l])
        a, b = grad[lo:hi], ref[lo:hi]
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, l
            continue
        # entries the oracle leaves at exactly zero carry (next to) nothing here either: the touched-entry set is index
        # bookkeeping; a few samples whose gradient is exactly 0 on one side (dead ReLUs, underflow) may be ~1e-5 on the other
        if bool((b == 0).any()):
            assert float(a[b == 0].abs().max()) <= 1e-3 * float(b.abs().max()), (l, float(a[b == 0].abs().max()), float(b.abs().max()))
        worst.append((rel(a, b), float((a - b).abs().max() / b.abs().max()), l))
        assert worst[-1][0] < tol_l2 and worst[-1][1] < tol_max, worst[-1]
    return worst


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_app_step_gradient_matches_oracle(golden_dir, dtype):
    """One colour training step (nerf.py:259-297): d loss / d app params from the explicit HIP chain vs autograd through the
    oracle on the same batch, random draws and sample set -- the geometry parameters must receive no gradient."""
    from perf_amd.scene import NeRFScene, Rays, SupInfoPool
    g, res, occ = _glue_setup(golden_dir)
    geo, app = _params(float(g['grid_gain']))
    o = torch.from_numpy(g['o']); d = torch.from_numpy(g['d'])
    R = o.shape[0]
    gt_dist, gt_rgb = O.synthetic_room(d)
    scene = NeRFScene(dtype=dtype)
    scene.renderer.render_step_size = float(g['step'])
    scene.train_conf.pixel_loss_batch_size = R
    scene.set_train()
    from perf_amd.nerfacc_impl import OccGridEstimator
    scene.estimator = OccGridEstimator(AABB, resolution=res).cuda(); scene.estimator.train()
    scene.estimator.set_binaries(torch.from_numpy(occ.reshape(-1)).cuda())
    with torch.no_grad():
        scene.nerf.geo_mlp.params.copy_(geo.cuda()); scene.nerf.app_mlp.params.copy_(app.cuda())
    pool = SupInfoPool(); pool.register_rays(o.cuda(), d.cuda(), gt_rgb.cuda(), gt_dist.cuda())
    pool.rand_ray_color_data = lambda bs, **kw: (Rays(pool.all_sup_rays.o, pool.all_sup_rays.d), pool.all_sup_colors,
                                                  pool.all_sup_distances, pool.all_sup_normals)
    rand = {k: torch.from_numpy(g[f'train_{k}']).cuda() for k in ('jitter', 'bg', 'noise')}
    captured = {}

    class _Catch:                                     # an "optimizer" that only records the gradient it is handed
        param_groups = [{'lr': 0.0}]
        def step(self):
            captured['grad'] = scene.nerf.app_mlp.params.grad.detach().clone()
    scene.fused_adam = False
    orig = scene._apply_grad
    def apply(net, grad, optimizer, dist_info, overlap, **kw):
        captured['grad'] = grad[:net.params.numel()].detach().clone(); captured['net'] = net
    scene._apply_grad = apply
    scene.train_one_step_app(_Catch(), pool, progress=0.5, rand=rand)
    scene._apply_grad = orig
    assert captured['net'] is scene.nerf.app_mlp
    grad = captured['grad'].cpu()
    # packed_info of the step = the renderer's on the same inputs
    out = scene.renderer.render(scene.nerf, scene.estimator, o.cuda(), d.cuda(), torch.zeros(R, 1).cuda(), torch.ones(R, 1).cuda(),
                                geo_inference=True, app_inference=True, rand=rand)
    t0 = (np.zeros(R, np.float32) + g['train_jitter'] * np.float32(float(g['step']))).astype(np.float32)
    kw = dict(training=True, t0=t0, bg_color=torch.from_numpy(g['train_bg']), dist_noise=torch.from_numpy(g['train_noise']),
              step=float(g['step']), quant=dtype)
    counts = _kept_counts_checked(out['packed_info'].cpu().numpy(), O.occ_render(o, d, geo, app, occ, AABB, return_pre=True, **kw)['pre'], BAND[dtype])
    app_r = app.clone().requires_grad_(True)
    geo_r = geo.clone().requires_grad_(True)
    ref = O.occ_render(o, d, geo_r, app_r, occ, AABB, kept_counts=counts, geo_grad=False, app_grad=True, **kw)
    loss, cl = O.app_step_loss(ref, gt_rgb)
    loss.backward()
    assert geo_r.grad is None
    assert abs(float(scene.last_losses['color_loss']) - float(cl)) < 2e-3 * (8.0 if dtype == 'bf16' else 1.0) * max(1.0, abs(float(cl)))
    worst = _assert_field_gradient_close(grad, app_r.grad, O.app_spec(), *GRAD_TOL[dtype])
    print(f'[app gradient {dtype}] worst level (rel L2, max-abs, level): {max(worst)}')


def test_shims_resolve_and_run():
    import perf_amd
    perf_amd.install_shims()
    import tinycudann as tcnn
    from nerfacc import accumulate_along_rays, render_weight_from_density
    from nerfacc.estimators.occ_grid import OccGridEstimator
    from torch_efficient_distloss import flatten_eff_distloss
    net = tcnn.NetworkWithInputEncoding(
        n_input_dims=3, n_output_dims=1,
        encoding_config={"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 18,
                         "base_resolution": 16, "per_level_scale": 1.4472692012786865},
        network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
                        "n_hidden_layers": 1})
    assert [n for n, _ in net.named_parameters()] == ['params'] and net.params.dtype == torch.float32
    x = torch.rand(1000, 3, device='cuda')
    y = net(x)
    assert y.shape == (1000, 1) and y.dtype in (torch.bfloat16, torch.float16)
    y.float().sum().backward()
    assert net.params.grad is not None and net.params.grad.shape == net.params.shape
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)      # PeRF hands .parameters() to Adam (nerf.py:171)
    opt.step()
    y2 = net(x)
    assert not torch.equal(y, y2)                          # the cached 16-bit copy was refreshed after the update
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                            "base_resolution": 16, "per_level_scale": 1.38, "interpolation": "Smoothstep"})
    xe = torch.rand(500, 3, device='cuda', requires_grad=True)
    f = enc(xe)
    assert f.shape == (500, 32)
    f.float().sum().backward()
    assert xe.grad is not None and enc.params.grad is not None
    est = OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=32, levels=1).cuda()
    assert set(est.state_dict().keys()) == {'resolution', 'aabbs', 'occs', 'binaries'}
    est.eval()
    with pytest.raises(RuntimeError):
        est.update_every_n_steps(step=0, occ_eval_fn=lambda x: x[:, 0])
    est.train()
    est.update_every_n_steps(step=0, occ_eval_fn=lambda x: (x.norm(dim=-1) < 0.8).float(), occ_thre=1e-2, ema_decay=0.1,
                             warmup_steps=256, n=1)
    o = torch.zeros(64, 3, device='cuda'); d = torch.nn.functional.normalize(torch.randn(64, 3, device='cuda'), dim=-1)
    ri, ts, te = est.sampling(o, d, sigma_fn=lambda a, b, c: torch.ones_like(a) * 3.0, near_plane=0., far_plane=1.5,
                              render_step_size=5e-3, stratified=True, cone_angle=0., alpha_thre=0.)
    assert ri.dtype == torch.int64 and ri.numel() > 0 and bool((ri[1:] >= ri[:-1]).all())
    sig = torch.full_like(ts, 2.0, requires_grad=True)
    w, T, a = render_weight_from_density(ts, te, sig, ray_indices=ri, n_rays=64)
    op = accumulate_along_rays(w, values=None, ray_indices=ri, n_rays=64)
    col = accumulate_along_rays(w.detach(), values=torch.rand(ri.numel(), 3, device='cuda'), ray_indices=ri, n_rays=64)
    assert op.shape == (64, 1) and col.shape == (64, 3)
    loss = op.sum() + flatten_eff_distloss(w, (ts + te) * .5, te - ts, ri)
    loss.backward()
    assert sig.grad is not None and torch.isfinite(sig.grad).all()


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_short_training_improves_psnr(dtype):
    """Equal-iteration training on the synthetic room: loss goes down and eval PSNR goes up."""
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays, psnr
    torch.manual_seed(0)
    scene = NeRFScene(dtype=dtype)
    rays = gen_pano_rays(torch.eye(4), 128, 256)
    dist, rgb = synthetic.room(rays.d)
    pool = SupInfoPool()
    pool.register_rays(rays.o, rays.d, rgb, dist)
    p0 = None
    tc = scene.train_conf
    tc.pixel_loss_batch_size = 2048
    scene.set_train()
    scene.prepare_occupancy(pool)
    scene.nerf.reset_geo()
    opt_g = scene.make_optimizer(scene.nerf.geo_mlp, 0.0)
    n_geo, n_app = 150, 100
    for i in range(n_geo):
        scene.update_lr(opt_g, tc.geo_optimizer, i / n_geo)
        scene.train_one_step_geo(opt_g, pool, progress=i / n_app)
        if i == 5:
            d0 = float(scene.last_losses['depth_loss'])
    d1 = float(scene.last_losses['depth_loss'])
    assert d1 < 0.5 * d0, (d0, d1)
    out0 = scene.render(rays, ['rgb', 'distance'])
    p0 = psnr(out0['rgb'], rgb)
    opt_a = scene.make_optimizer(scene.nerf.app_mlp, 0.0)
    for i in range(n_app):
        scene.update_lr(opt_a, tc.app_optimizer, i / n_app)
        scene.train_one_step_app(opt_a, pool, progress=i / n_app)
    out1 = scene.render(rays, ['rgb', 'distance'])
    p1 = psnr(out1['rgb'], rgb)
    derr = float((out1['distance'] - dist).abs().mean())
    assert p1 > p0 + 3.0 and p1 > 15.0, (p0, p1)
    assert derr < 0.05, derr
    print(f'[{dtype}] PSNR {p0:.2f} -> {p1:.2f} dB, mean |distance err| {derr:.4f}')


def test_prop_sampling_matches_oracle():
    """a7: proposal-guided hierarchical resampling (NGPDensityField proposal nets, 128 -> 64 -> 64 samples)."""
    from perf_amd.fields import NGPDensityField
    from perf_amd.nerfacc_impl import PropNetEstimator
    torch.manual_seed(0)
    R = 33
    o = torch.zeros(R, 3); d = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1)
    props = [NGPDensityField(AABB, n_levels=5, max_resolution=128, dtype='fp16'),
             NGPDensityField(AABB, n_levels=5, max_resolution=256, dtype='fp16')]
    specs = []
    for p_ in props:
        lv = O.grid_levels(5, 2, 17, 16, float(np.exp((np.log(p_.max_resolution) - np.log(16)) / 4)))
        spec = O.FieldSpec(lv, 1, 1, 'None')
        with torch.no_grad():
            p_.mlp_base.params[spec_n_net(p_):] *= 3e4             # non-trivial densities
        specs.append((spec, p_.mlp_base.params.detach().cpu().clone()))
    taus = [torch.rand(R) for _ in range(3)]

    def gpu_fn(net):
        def fn(ts, te):
            pos = o.cuda()[:, None, :] + d.cuda()[:, None, :] * (ts + te)[..., None] / 2.0
            return net(pos).squeeze(-1)
        return fn

    est = PropNetEstimator()
    ts, te = est.sampling([gpu_fn(p_) for p_ in props], [128, 64], 64, R, 1e-2, 2.0, stratified=True,
                          taus=[t.cuda() for t in taus])

    def cpu_fn(spec, params):
        def fn(ts_, te_):
            pos = o[:, None, :] + d[:, None, :] * torch.from_numpy(ts_ + te_)[..., None] / 2.0
            w1 = params[:spec_net_params(spec)]
            sig = O.query_density(pos.reshape(-1, 3), _pad_net(params, spec), spec, torch.tensor(AABB), quant='fp16', shift=1.0)
            return sig.reshape(ts_.shape).numpy()
        return fn

    rts, rte = O.prop_sampling([cpu_fn(*sp) for sp in specs], [128, 64], 64, R, 1e-2, 2.0, taus=[t.numpy() for t in taus])
    assert ts.shape == (R, 64)
    # the first level depends on nothing learned: bit-exact; later levels inherit 16-bit field differences
    assert np.abs(ts.cpu().numpy() - rts).max() < 2e-2
    assert bool((te >= ts).all()) and bool((ts[:, 1:] >= ts[:, :-1]).all())


def test_prop_renderer_end_to_end():
    """a7: NeRFPropRenderer.render (mirror of nerf_renderer.py:26-102, dead in the reference) runs end to end on the HIP
    kernels -- proposal resampling 128 -> 64 -> 64, field queries, dense alpha compositing -- and its outputs equal an fp32
    restatement of :72-99 evaluated on the renderer's own samples and the oracle's 16-bit field emulation."""
    from perf_amd.fields import NGPDensityField
    from perf_amd.nerfacc_impl import PropNetEstimator
    from perf_amd.renderer import NeRFPropRenderer
    torch.manual_seed(0)
    dtype = 'fp16'
    geo, app = _params()
    nerf = _nerf(dtype, geo, app)
    nerf.eval()
    props = [NGPDensityField(AABB, n_levels=5, max_resolution=128, dtype=dtype), NGPDensityField(AABB, n_levels=5, max_resolution=256, dtype=dtype)]
    for p_ in props:
        with torch.no_grad():
            p_.mlp_base.params[spec_n_net(p_):] *= 3e4
    R = 257
    g = torch.Generator().manual_seed(4)
    o = (torch.rand(R, 3, generator=g) - 0.5) * 0.2
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    rend = NeRFPropRenderer(max_radius=2, bg_color='black')
    out = rend.render(nerf, props, PropNetEstimator(), o.cuda(), d.cuda(), torch.zeros(R, 1).cuda(), torch.ones(R, 1).cuda())
    ts, te = out['t_starts'].cpu(), out['t_ends'].cpu()
    assert ts.shape == (R, 64) and bool((te >= ts).all()) and bool((ts[:, 1:] >= ts[:, :-1] - 1e-7).all())
    assert float(ts.min()) >= 1e-2 - 1e-6 and float(te.max()) <= 2.0 + 1e-5
    # fp32 restatement on the same samples
    pos = o[:, None, :] + d[:, None, :] * (ts + te)[..., None] / 2.0
    aabb = torch.tensor(AABB)
    sig = O.query_density(pos.reshape(-1, 3), geo, O.geo_spec(), aabb, quant=dtype).reshape(R, 64)
    rgb = O.query_rgb(pos.reshape(-1, 3), app, O.app_spec(), aabb, quant=dtype).reshape(R, 64, 3)
    alpha = 1.0 - torch.exp(-sig * (te - ts))
    T = torch.cumprod(torch.cat([torch.ones(R, 1), 1.0 - alpha[:, :-1]], -1), -1)
    w = T * alpha
    assert (out['weights'].cpu() - w).abs().max() < 5e-3
    assert (out['opacities'].cpu() - w.sum(1, keepdim=True)).abs().max() < 5e-3
    assert (out['rgb'].cpu() - (w[..., None] * rgb).sum(1)).abs().max() < 5e-3              # black background: no extra term
    noise_free = (w * (ts + te) / 2.0).sum(1, keepdim=True)
    dd = out['distance'].cpu() - noise_free                                                  # + U[0,1) * (1 - opacity), :99
    rest = (1.0 - w.sum(1, keepdim=True)).clamp_min(0)
    assert bool((dd >= -5e-3).all()) and bool((dd <= rest + 5e-3).all())


def spec_n_net(field):
    return field.mlp_base.mlp.n_params


def spec_net_params(spec):
    return spec.n_net


def _pad_net(params, spec):
    """The product stores the first layer with its input width padded to 16 (tcnn layout); the oracle's FieldSpec has the
    same layout (spec.n_in), so the parameter vector is shared as it is."""
    assert params.numel() == spec.n_params
    return params


def test_encoding_double_backward():
    """next-1: tcnn.Encoding (Smoothstep) with input gradient and double backward, as SphereDistanceField uses it
    (pano_joint_predictor.py:22-71): d/dtheta and d/dx of a loss on the input gradient vs autograd through the oracle."""
    from perf_amd import tcnn
    cfg = {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16,
           "per_level_scale": 1.5, "interpolation": "Smoothstep"}
    enc = tcnn.Encoding(3, cfg, dtype='fp32')      # fp32 output: a half cast would overflow the second-order gradients
    with torch.no_grad():
        enc.params.mul_(1e4)
    lv = O.grid_levels(8, 2, 15, 16, 1.5)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(300, 3, generator=g) * 0.9 + 0.05)
    proj = torch.randn(16, generator=g)
    # fast path (no input grad): kernel forward equals the composed forward
    y_fast = enc(x.cuda()).float().cpu()
    xg = x.clone().cuda().requires_grad_(True)
    y = enc(xg).float()
    assert (y.detach().cpu() - y_fast).abs().max() < 1e-5 * float(y_fast.abs().max())
    out = torch.tanh(y @ proj.cuda())
    (gx,) = torch.autograd.grad(out.sum(), xg, create_graph=True)
    loss2 = (gx ** 2).sum() + out.sum()
    loss2.backward()
    # oracle
    table = enc.params.detach().cpu().view(-1, 2).clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = O.hashgrid_encode(xr, table, lv, interpolation='Smoothstep')
    outr = torch.tanh(yr @ proj)
    (gxr,) = torch.autograd.grad(outr.sum(), xr, create_graph=True)
    l2r = (gxr ** 2).sum() + outr.sum()
    l2r.backward()
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))
    assert rel(gx.detach().cpu(), gxr.detach()) < 1e-4
    assert rel(enc.params.grad.cpu(), table.grad.reshape(-1)) < 1e-4
    assert rel(xg.grad.cpu(), xr.grad) < 1e-4


def test_dp_overlap_path_world1():
    """The data-parallel step (async RCCL all-reduce overlapped with the next step's prefetch) on a world-size-1
    process group: same parameters after 3 steps as the plain single-process path with the same seeds."""
    import os
    import torch.distributed as dist
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')

    def run(use_dist):
        torch.manual_seed(0)
        scene = NeRFScene(dtype='bf16')
        rays = gen_pano_rays(torch.eye(4), 64, 128)
        d_, rgb = synthetic.room(rays.d)
        pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, d_)
        scene.train_conf.pixel_loss_batch_size = 1024
        scene.set_train()
        scene.estimator.set_binaries(torch.ones(256 ** 3, dtype=torch.uint8, device='cuda'))
        r = scene.renderer
        r.render_step_size = 0.99 / 32; r.far_plane = 10.0; r.early_stop_eps = 0.0; r.max_steps = 32
        scene.nerf.reset_geo()
        opt = scene.make_optimizer(scene.nerf.geo_mlp, 1e-3)
        gen = torch.Generator(device='cuda'); gen.manual_seed(7)
        torch.manual_seed(1)
        for i in range(3):
            scene.train_one_step_geo(opt, pool, progress=0.3, generator=gen)
        return scene.nerf.geo_mlp.params.detach().clone()

    ref = run(False)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        got = run(True)
    finally:
        dist.destroy_process_group()
    # identical batches (same generator); the jitter / noise draws come from the default generator in a different
    # order when the next batch is prefetched early, so compare loosely: parameters moved the same way
    assert float((got - ref).abs().max()) < 5e-3 and float((got - ref).abs().mean()) < 1e-4


def test_pano_visibility_mask_same_pose_is_visible():
    """a12: surface points rendered from the pose of a registered panorama are visible in that panorama."""
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    from perf_amd.visibility import geo_check, pano_visibility_mask
    rays = gen_pano_rays(torch.eye(4), 64, 128)
    dist, rgb = synthetic.room(rays.d)
    pool = SupInfoPool()
    pool.register_sup_info(torch.eye(4), torch.ones(64, 128, 1, device='cuda'), rgb, dist)
    # (PanoSupInfo drops the pixels on depth edges -- the room's wall-to-wall creases at this resolution)
    assert 0.5 * 64 * 128 < len(pool) <= 64 * 128 and len(pool.sup_infos) == 1
    assert len(pool) == int(pool.sup_infos[0]['mask'].sum())
    vis = pano_visibility_mask(rays.o, rays.d, dist[..., 0], pool.sup_infos)
    assert vis.shape == (64, 128) and float(vis[pool.sup_infos[0]['mask'][..., 0]].mean()) > 0.9
    # points pushed well behind the observed surface are occluded; points in front conflict with the geo check
    vis_far = pano_visibility_mask(rays.o, rays.d, dist[..., 0] * 1.5, pool.sup_infos)
    assert float(vis_far.mean()) < 0.01
    ok_behind = geo_check(rays.o, rays.d, dist * 1.2, pool.sup_infos)
    ok_front = geo_check(rays.o, rays.d, dist * 0.5, pool.sup_infos)
    assert float(ok_behind.mean()) > 0.99 and float(ok_front.mean()) < 0.01


def test_checkpoint_format_and_render_dense(tmp_path):
    """next-4 / config 4: the scene state_dict has the reference's checkpoint layout (nerf.py:374-380,
    core_exp_runner.py:248-256) and round-trips through torch.save; render_dense walks a dense trajectory."""
    from perf_amd import synthetic
    from perf_amd.pose_sampler import CirclePoseSampler
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    from perf_amd.traverse import render_dense
    torch.manual_seed(0)
    scene = NeRFScene(dtype='fp16')
    rays = gen_pano_rays(torch.eye(4), 64, 128)
    dist, rgb = synthetic.room(rays.d)
    pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
    scene.set_train(); scene.prepare_occupancy(pool)
    sd = scene.state_dict()
    # (+ one private top-level key: the fixed-point headroom state; the reference's loader reads only its own three, nerf.py:368-380)
    assert set(sd.keys()) == {'render', 'nerf', 'estimator', '_perf_amd'}
    assert set(sd['_perf_amd'].keys()) == {'geo_headroom', 'app_headroom'}
    assert set(sd['nerf'].keys()) == {'aabb', 'geo_mlp.params', 'app_mlp.params'}
    assert set(sd['estimator'].keys()) == {'resolution', 'aabbs', 'occs', 'binaries'}
    assert sd['nerf']['geo_mlp.params'].dtype == torch.float32 and sd['nerf']['geo_mlp.params'].numel() == 3072 + 6641216
    assert sd['estimator']['binaries'].shape == (1, 256, 256, 256) and sd['estimator']['binaries'].dtype == torch.bool
    path = tmp_path / 'ckpt.pth'
    torch.save({'scene': sd, 'phase': 3}, path)
    before = scene.render(rays, ['rgb', 'distance'])
    scene2 = NeRFScene(dtype='fp16')
    with torch.no_grad():
        scene2.nerf.geo_mlp.params.add_(1.0)                       # make sure loading really restores the weights
    ck = torch.load(path, map_location='cuda')
    scene2.load_state_dict(ck['scene'])
    after = scene2.render(rays, ['rgb', 'distance'])
    assert torch.equal(before['rgb'], after['rgb']) and torch.equal(before['distance'], after['distance'])
    sampler = CirclePoseSampler(dist[..., 0].cpu(), [.2, .4, .6], [8, 8, 8])
    import numpy as np
    np.random.seed(0)
    frames = render_dense(scene2, sampler, n_poses=12, height=32, width=64, max_frames=3)
    assert len(frames) == 3 and frames[0]['rgb'].shape == (32, 64, 3) and torch.isfinite(frames[2]['distance']).all()


def test_training_is_bit_reproducible():
    """Default path (packed fixed-point grid gradient, two-stage MLP weight-gradient reduction, atomic-free compositing):
    two runs with the same seeds give bit-identical parameters."""
    from perf_amd import synthetic, tcnn
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    if tcnn.GRID_GRAD_ACCUM != 'fixed':
        pytest.fail('an earlier test overflowed the fixed-point grid gradient (fell back to order-dependent fp32 atomics)')

    def run():
        torch.manual_seed(0)
        scene = NeRFScene(dtype='bf16')
        rays = gen_pano_rays(torch.eye(4), 64, 128)
        d_, rgb = synthetic.room(rays.d)
        pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, d_)
        scene.train_conf.pixel_loss_batch_size = 1024
        scene.set_train(); scene.prepare_occupancy(pool); scene.nerf.reset_geo()
        opt = scene.make_optimizer(scene.nerf.geo_mlp, 1e-3)
        torch.manual_seed(1)
        for i in range(6):
            scene.train_one_step_geo(opt, pool, progress=0.3)
        opt2 = scene.make_optimizer(scene.nerf.app_mlp, 1e-3)
        for i in range(4):
            scene.train_one_step_app(opt2, pool, progress=0.3)
        return scene.nerf.geo_mlp.params.detach().clone(), scene.nerf.app_mlp.params.detach().clone()

    g1, a1 = run(); g2, a2 = run()
    assert torch.equal(g1, g2) and torch.equal(a1, a2)


def test_fused_steps_equal_autograd_steps():
    """The explicit kernel chains of the two training steps give the same parameter updates as the autograd formulation
    (same seeds, default bit-reproducible accumulation): gradients agree to fp32 rounding of the loss-scale products."""
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays

    def run(fused, faithful):
        torch.manual_seed(0)
        scene = NeRFScene(dtype='fp16')
        scene.fused_steps = fused
        rays = gen_pano_rays(torch.eye(4), 64, 128)
        d_, rgb = synthetic.room(rays.d)
        pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, d_)
        scene.train_conf.pixel_loss_batch_size = 1024
        scene.set_train()
        if faithful:
            scene.prepare_occupancy(pool)
        else:
            scene.estimator.set_binaries(torch.ones(256 ** 3, dtype=torch.uint8, device='cuda'))
            r = scene.renderer
            r.render_step_size = 0.99 / 32; r.far_plane = 10.0; r.early_stop_eps = 0.0; r.max_steps = 32
        scene.nerf.reset_geo()
        gen = torch.Generator(device='cuda'); gen.manual_seed(5)
        g = torch.Generator(device='cuda'); g.manual_seed(9)
        opt = scene.make_optimizer(scene.nerf.geo_mlp, 1e-3)
        for i in range(3):
            rand = {'jitter': torch.rand(1024, device='cuda', generator=g), 'bg': torch.rand(1024, 3, device='cuda', generator=g),
                    'noise': torch.rand(1024, 1, device='cuda', generator=g)}
            scene.train_one_step_geo(opt, pool, progress=0.3, rand=rand, generator=gen)
        dl = float(scene.last_losses['depth_loss']); dd = float(scene.last_losses['dist_loss'])
        opt2 = scene.make_optimizer(scene.nerf.app_mlp, 1e-3)
        for i in range(2):
            rand = {'jitter': torch.rand(1024, device='cuda', generator=g), 'bg': torch.rand(1024, 3, device='cuda', generator=g),
                    'noise': torch.rand(1024, 1, device='cuda', generator=g)}
            scene.train_one_step_app(opt2, pool, progress=0.3, rand=rand, generator=gen)
        return scene.nerf.geo_mlp.params.detach().clone(), scene.nerf.app_mlp.params.detach().clone(), dl, dd, float(scene.last_losses['color_loss'])

    for faithful in (False, True):
        g0, a0, dl0, dd0, cl0 = run(False, faithful)
        g1, a1, dl1, dd1, cl1 = run(True, faithful)
        assert abs(dl0 - dl1) < 1e-5 * max(1, abs(dl0)) and abs(dd0 - dd1) < 1e-4 * max(1e-3, abs(dd0)) and abs(cl0 - cl1) < 1e-5
        # Adam divides by sqrt(v): tiny gradient differences can flip the sign of a near-zero update, so compare loosely
        assert float((g0 - g1).abs().max()) < 2.5e-3 and float((g0 - g1).abs().mean()) < 2e-5
        assert float((a0 - a1).abs().max()) < 2.5e-3 and float((a0 - a1).abs().mean()) < 2e-5


def test_skipping_the_unused_colour_render_changes_nothing():
    """The geometry step's colour render feeds no loss (nerf.py:197-252): dropping it leaves the parameters bit-identical."""
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays

    def run(skip):
        torch.manual_seed(0)
        scene = NeRFScene(dtype='bf16')
        scene.skip_unused_color = skip
        rays = gen_pano_rays(torch.eye(4), 64, 128)
        d_, rgb = synthetic.room(rays.d)
        pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, d_)
        scene.train_conf.pixel_loss_batch_size = 1024
        scene.set_train(); scene.prepare_occupancy(pool); scene.nerf.reset_geo()
        gen = torch.Generator(device='cuda'); gen.manual_seed(5)
        g = torch.Generator(device='cuda'); g.manual_seed(9)
        opt = scene.make_optimizer(scene.nerf.geo_mlp, 1e-3)
        for i in range(4):
            rand = {'jitter': torch.rand(1024, device='cuda', generator=g), 'noise': torch.rand(1024, 1, device='cuda', generator=g)}
            scene.train_one_step_geo(opt, pool, progress=0.3, rand=rand, generator=gen)
        return scene.nerf.geo_mlp.params.detach().clone()

    assert torch.equal(run(False), run(True))


def test_reprojection_and_morphology_kernels_match_torch_formulation():
    """next-3: perf_pano_reproject / perf_morph_binary against the torch formulation of nerf.py:321-358 and
    sup_info.py:261-302 (grid_sample border bilinear + conv morphology).  Morphology is bit-exact; the depth test may differ
    only on pixels whose |distance - looked-up distance| is within float rounding of the threshold."""
    from perf_amd import ops, synthetic
    from perf_amd import visibility as V
    from perf_amd.scene import SupInfoPool, gen_pano_rays
    g = torch.Generator(device='cpu').manual_seed(5)
    for shape, p in (((37, 53), 0.15), ((64, 128), 0.6), ((512, 1024), 0.97)):
        m = (torch.rand(shape, generator=g) < p).float().cuda()
        for rows, cols in ((3, 3), (5, 5), (9, 9), (5, 9)):
            k = V.ellipse_kernel(rows, cols)
            assert torch.equal(ops.morph_binary(m, k.flip(0, 1), 'dilate'), V.dilate(m[None, None], k.cuda())[0, 0])
            assert torch.equal(ops.morph_binary(m, k, 'erode'), V.erode(m[None, None], k.cuda())[0, 0])
    # three registered panoramas at different poses with partly masked distance maps
    pool = SupInfoPool()
    poses = []
    for i, t in enumerate(((0., 0., 0.), (0.25, -0.1, 0.05), (-0.2, 0.3, -0.1))):
        pose = torch.eye(4)
        a = 0.4 * i
        pose[:3, :3] = torch.tensor([[math.cos(a), -math.sin(a), 0.], [math.sin(a), math.cos(a), 0.], [0., 0., 1.]])
        pose[:3, 3] = torch.tensor(t)
        rays = gen_pano_rays(pose, 128, 256)
        dist, rgb = synthetic.room(rays.d)     # not a consistent scene across poses: only the two formulations are compared
        msk = torch.ones(128, 256, 1, device='cuda')
        msk[40:60, 30 * i:30 * i + 50] = 0
        pool.register_sup_info(pose, msk, rgb, dist)
        poses.append(pose)
    view = torch.eye(4); view[:3, 3] = torch.tensor([0.1, 0.05, -0.05])
    rays = gen_pano_rays(view, 192, 384)
    dist, _ = synthetic.room(rays.d)
    for scale in (0.6, 1.0, 1.3):
        d = dist[..., 0] * scale
        pts = rays.o + rays.d * d[..., None]
        # raw depth tests (before morphology)
        flat = pts.reshape(-1, 3).contiguous()
        vis = torch.zeros(flat.shape[0], device='cuda'); geo = torch.ones(flat.shape[0], device='cuda')
        vis_t = torch.zeros(192, 384, 1, device='cuda'); geo_t = torch.ones(192, 384, 1, device='cuda')
        margin = torch.full((192, 384, 1), 1e9, device='cuda')
        for info in pool.sup_infos:
            dm = (info['distance_map'] * info['mask'].float())[..., 0].contiguous()
            ops.pano_reproject(flat, info['pose'], dm, vis, 0)
            ops.pano_reproject(flat, info['pose'], dm, geo, 1)
            dd, proj = V._lookup_distance(pts, info)
            vis_t = torch.maximum(vis_t, (dd < proj + 1 / 256.).float())
            geo_t = torch.minimum(geo_t, (proj < dd).float())
            margin = torch.minimum(margin, torch.minimum((dd - proj - 1 / 256.).abs(), (dd - proj).abs()))
        near = margin[..., 0].reshape(-1) < 2e-5
        assert float(near.float().mean()) < 0.02
        assert torch.equal(vis[~near], vis_t.reshape(-1)[~near]) and torch.equal(geo[~near], geo_t.reshape(-1)[~near])
        # end to end through both public functions
        a = V.pano_visibility_mask(rays.o, rays.d, d, pool.sup_infos)
        b = V.pano_visibility_mask(rays.o, rays.d, d, pool.sup_infos, use_kernels=False)
        assert float((a != b).float().mean()) < 0.01
        a = V.geo_check(rays.o, rays.d, d[..., None], pool.sup_infos)
        b = V.geo_check(rays.o, rays.d, d[..., None], pool.sup_infos, use_kernels=False)
        assert float((a != b).float().mean()) < 0.01


@pytest.mark.parametrize('nh,n_out,act', [(1, 1, 'None'), (2, 3, 'Sigmoid')])
def test_network_with_20_level_grid_forward_and_gradient(nh, n_out, act):
    """BASELINE config 5's field shape (L = 20 levels -> 40 input features) through tcnn.NetworkWithInputEncoding: output and
    the flat [network | grid] gradient against the oracle's 16-bit emulation -- the MLP backward's second 32-feature block
    and the grid backward at 20 levels."""
    from perf_amd import tcnn
    dtype = 'fp16'
    enc = {"otype": "HashGrid", "n_levels": 20, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16,
           "per_level_scale": 1.3}
    net_cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act, "n_neurons": 64, "n_hidden_layers": nh}
    net = tcnn.NetworkWithInputEncoding(3, n_out, enc, net_cfg, dtype=dtype)
    spec = O.FieldSpec(O.grid_levels(n_levels=20, log2_hashmap_size=15, base_resolution=16, per_level_scale=1.3), nh, n_out, act)
    assert net.params.numel() == spec.n_params
    g = torch.Generator().manual_seed(21)
    p0 = O.init_field_params(spec, seed=3)
    p0[spec.n_net:] *= 3000.0                                    # features of order 0.3: every level matters to the output
    with torch.no_grad():
        net.params.copy_(p0.cuda())
    n = 1500
    x = torch.rand(n, 3, generator=g) * 0.98 + 0.01
    dout = torch.randn(n, n_out, generator=g)
    y = net(x.cuda(), out_fp32=True)
    (y * dout.cuda()).sum().backward()
    pr = p0.clone().requires_grad_(True)
    yr = O.network_with_encoding(x, pr, spec, quant=dtype)
    (yr * dout).sum().backward()
    ulp = 2.0 ** -10
    assert float((y.detach().cpu() - yr.detach()).abs().max()) < 16 * ulp * max(1.0, float(yr.abs().max()))
    _assert_field_gradient_close(net.params.grad.cpu(), pr.grad, spec)


def test_supervision_sampling_modes():
    """sup_info.py:236-259: 'by_all_pixels' draws over every registered ray, 'only_first' / 'only_last' over one panorama's."""
    from perf_amd.scene import SupInfoPool, gen_pano_rays
    pool = SupInfoPool()
    for k, (h, w) in enumerate(((8, 16), (4, 8), (6, 12))):
        rays = gen_pano_rays(torch.eye(4), h, w)
        pool.register_rays(rays.o, rays.d, torch.full((h * w, 3), float(k), device='cuda'), torch.ones(h * w, 1, device='cuda'))
    assert len(pool) == 128 + 32 + 72
    g = torch.Generator(device='cuda')
    for mode, want in (('only_first', {0.0}), ('only_last', {2.0}), ('by_all_pixels', {0.0, 1.0, 2.0})):
        g.manual_seed(3)
        _, col, _, _ = pool.rand_ray_color_data(4096, rand_mode=mode, generator=g)
        assert set(col[:, 0].unique().tolist()) == want, mode
    # the index stream of a mode is torch.randint(0, n_mode) of the same generator state, shifted to the panorama's range
    g.manual_seed(3)
    _, col_last, _, _ = pool.rand_ray_color_data(64, rand_mode='only_last', generator=g)
    g.manual_seed(3)
    idx = torch.randint(0, 72, (64,), device='cuda', generator=g) + 160
    assert torch.equal(col_last, pool.all_sup_colors[idx])


def test_pano_sup_info_validity_rules_and_pool_state_dict(tmp_path):
    """PanoSupInfo (sup_info.py:27-120): supervision skips masked pixels, zero distances, depth edges (normalised 3x3
    Laplacian >= 0.01, cleaned by a 3x3 erosion + dilation) and -- with normals -- grazing surfaces; SupInfoPool.state_dict
    keeps the reference's keys and load_state_dict restores the rays."""
    from perf_amd.scene import SupInfoPool, gen_pano_rays, _edge_free
    h, w = 32, 64
    dist = torch.full((h, w, 1), 0.5, device='cuda')
    dist[:, 40:] = 0.9                                   # a depth step between columns 39 | 40
    dist[3, 5] = 0.0                                     # a hole
    mask = torch.ones(h, w, 1, device='cuda'); mask[20:24, 10:14] = 0
    rgb = torch.rand(h, w, 3, device='cuda')
    ef = _edge_free(dist)[..., 0]
    # restated by hand: |lap| with reflect padding, then opening by a 3x3 box
    x = torch.nn.functional.pad(dist[..., 0][None, None], (1, 1, 1, 1), mode='reflect')
    lap = (torch.nn.functional.avg_pool2d(x, 3, 1) * 9 - 9 * dist[..., 0][None, None]) / 16
    smooth = (lap.abs() < 0.01)[0, 0]
    er = ~(torch.nn.functional.max_pool2d((~smooth).float()[None, None], 3, 1, 1)[0, 0] > 0.5)       # erosion: borders do not erode
    op = torch.nn.functional.max_pool2d(er.float()[None, None], 3, 1, 1)[0, 0] > 0.5
    assert torch.equal(ef, op)
    assert not bool(ef[:, 39:41].any()) and not bool(ef[3, 5]) and bool(ef[8:, 10:36].all())
    pool = SupInfoPool()
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.1, 0.0, -0.05])
    pool.register_sup_info(pose, mask, rgb, dist)
    info = pool.sup_infos[0]
    want = (mask[..., 0] > .5) & (dist[..., 0] > 1e-5) & ef
    assert torch.equal(info['mask'][..., 0], want) and len(pool) == int(want.sum())
    rays = gen_pano_rays(pose, h, w)
    assert torch.equal(pool.all_sup_rays.d, rays.d[torch.where(want)]) and torch.equal(pool.all_sup_colors, rgb[torch.where(want)])
    # normals: a surface facing the camera passes, a grazing one does not
    local = gen_pano_rays(torch.eye(4), h, w)
    facing = -local.d
    tilted = torch.nn.functional.normalize(torch.cross(local.d, torch.tensor([0., 0., 1.], device='cuda').expand_as(local.d), dim=-1), dim=-1)
    flat = torch.full((h, w, 1), 0.7, device='cuda')
    p2 = SupInfoPool(); p2.register_sup_info(torch.eye(4), torch.ones(h, w, 1, device='cuda'), rgb, flat, facing)
    p3 = SupInfoPool(); p3.register_sup_info(torch.eye(4), torch.ones(h, w, 1, device='cuda'), rgb, flat, tilted)
    assert len(p2) == h * w and len(p3) == 0
    # checkpoint round trip with the reference's keys
    pool.register_sup_info(torch.eye(4), torch.ones(h, w, 1, device='cuda'), rgb, flat)
    sd = pool.state_dict()
    assert sd['n_sup_infos'] == 2 and set(sd['sup_info_0'].keys()) == set(SupInfoPool._INFO_KEYS)
    assert 'sup_info_{}_height' in sd and 'sup_info_{}_width' in sd
    torch.save(sd, tmp_path / 'pool.pth')
    q = SupInfoPool(); q.load_state_dict(torch.load(tmp_path / 'pool.pth'))
    assert len(q) == len(pool) and torch.equal(q.all_sup_rays.o, pool.all_sup_rays.o) and torch.equal(q.all_sup_distances, pool.all_sup_distances)
    assert q._ranges == pool._ranges and len(q.sup_infos) == 2


# ---- the HIP path against what the REFERENCE's own functions produced (tests/golden/train_glue.npz, visibility.npz) -----------
def _glue_scene(g, tag, dtype):
    """NeRFScene set up like the recorded step of tests/golden/make_fixtures.py:fx_train_glue."""
    from perf_amd.nerfacc_impl import OccGridEstimator
    from perf_amd.scene import NeRFScene, Rays, SupInfoPool
    h, w, res = int(g['h']), int(g['w']), int(g['res'])
    o_all, d_all = O.pano_rays(torch.eye(4), h, w)
    o_all = o_all.reshape(-1, 3); d_all = d_all.reshape(-1, 3)
    dist_all, rgb_all = O.synthetic_room(d_all)
    occ = O.gen_occ_grid(o_all, d_all, dist_all, res)
    idx = torch.from_numpy(g[f'{tag}_idx'])
    gs, as_ = O.geo_spec(), O.app_spec()
    geo = O.init_field_params(gs, int(g['geo_seed'])); app = O.init_field_params(as_, int(g['app_seed']))
    geo[gs.n_net:] *= float(g['grid_gain']); app[as_.n_net:] *= float(g['grid_gain'])
    geo[:gs.n_net] *= 3.0
    scene = NeRFScene(dtype=dtype)
    scene.renderer.render_step_size = float(g['step'])
    scene.train_conf.pixel_loss_batch_size = len(idx)
    scene.set_train()
    scene.estimator = OccGridEstimator(AABB, resolution=res).cuda(); scene.estimator.train()
    scene.estimator.set_binaries(occ.cuda())
    with torch.no_grad():
        scene.nerf.geo_mlp.params.copy_(geo.cuda()); scene.nerf.app_mlp.params.copy_(app.cuda())
    pool = SupInfoPool()
    pool.register_rays(o_all[idx].cuda(), d_all[idx].cuda(), rgb_all.reshape(-1, 3)[idx].cuda(), dist_all.reshape(-1, 1)[idx].cuda())
    pool.rand_ray_color_data = lambda bs, **kw: (Rays(pool.all_sup_rays.o, pool.all_sup_rays.d), pool.all_sup_colors,
                                                  pool.all_sup_distances, pool.all_sup_normals)
    rand = {k: torch.from_numpy(g[f'{tag}_{k}']).cuda() for k in ('jitter', 'bg', 'noise')}
    return scene, pool, rand


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_training_steps_match_the_reference_glue(golden_dir, dtype):
    """NeRFScene.train_one_step_geo / train_one_step_app on the HIP kernels against the gradient and the loss terms the
    REFERENCE's own train_one_step_geo / _app produced (fp32 oracle operators underneath, nerf.py:186-297): same batch, same
    random draws.  The HIP side evaluates 16-bit fields, so the comparison is at 16-bit accuracy: loss terms to a few per
    cent, the optimizer's gradient by direction and norm (a sample at the early-stop threshold may be kept on one side only).
    Together with tests/test_cpu_oracle.py::test_train_step_glue_matches_reference (oracle == reference glue, 2e-5) and the
    oracle-level gradient tests above (HIP == oracle's 16-bit emulation per grid level) this closes the chain
    reference glue -> oracle -> HIP for the training steps."""
    g = np.load(f'{golden_dir}/train_glue.npz')
    for tag in ('geo_p2', 'app_p5'):
        kind = tag[:3]
        scene, pool, rand = _glue_scene(g, tag, dtype)
        captured = {}

        class _Catch:
            param_groups = [{'lr': 0.0}]

            def step(self):
                pass
        scene.fused_adam = False

        def apply(net, grad, optimizer, dist_info, overlap, **kw):
            captured['grad'] = grad[:net.params.numel()].detach().clone(); captured['net'] = net
        scene._apply_grad = apply
        if kind == 'geo':
            scene.train_one_step_geo(_Catch(), pool, progress=float(g[f'{tag}_progress']), rand=rand, prefetch_next=False)
        else:
            scene.train_one_step_app(_Catch(), pool, progress=float(g[f'{tag}_progress']), rand=rand)
        grad = captured['grad'].cpu().double()
        ref = torch.zeros(int(g[f'{tag}_grad_numel']), dtype=torch.float64)
        ref[torch.from_numpy(g[f'{tag}_grad_idx']).long()] = torch.from_numpy(g[f'{tag}_grad_val']).double()
        cos = float((grad @ ref) / (grad.norm() * ref.norm()))
        ratio = float(grad.norm() / ref.norm())
        losses = {k: float(v) for k, v in scene.last_losses.items()}
        print(f'[{tag} {dtype}] gradient cosine {cos:.5f}, norm ratio {ratio:.4f}, losses {losses}')
        lim = (0.9999, 0.002) if dtype == 'fp16' else (0.999, 0.005)      # measured: 1.00000 / 0.9998 and 0.99977 / 1.0008
        assert cos > lim[0] and abs(ratio - 1.0) < lim[1], (tag, cos, ratio)
        if kind == 'geo':
            assert abs(losses['depth_loss'] - float(g[f'{tag}_depth_loss'])) < 5e-3 * float(g[f'{tag}_depth_loss'])
            assert abs(losses['dist_loss'] - float(g[f'{tag}_dist_loss'])) < 1e-2 * float(g[f'{tag}_dist_loss'])
        else:
            assert abs(losses['color_loss'] - float(g[f'{tag}_color_loss'])) < 2e-3 * float(g[f'{tag}_color_loss'])


def test_sup_info_and_visibility_match_the_reference(golden_dir):
    """SupInfoPool.register_sup_info == PanoSupInfo.__init__ / update_sup_info (sup_info.py:27-120) and the reprojection
    kernels (perf_pano_reproject + perf_morph_binary) == the reference's get_pano_visibility_mask (nerf.py:321-358) and
    SupInfoPool.geo_check (sup_info.py:261-302) as recorded in tests/golden/visibility.npz."""
    from perf_amd.scene import NeRFScene, Rays, SupInfoPool, gen_pano_rays
    g = np.load(f'{golden_dir}/visibility.npz')
    h, w = int(g['h']), int(g['w'])
    pool = SupInfoPool()
    for i in range(2):
        f = lambda k: torch.from_numpy(g[f'pano{i}_{k}']).cuda()
        pool.register_sup_info(f('pose'), f('mask_in'), f('rgb'), f('distance'), f('normal'))
        info = pool.sup_infos[i]
        assert np.array_equal(info['mask_raw'].cpu().numpy(), g[f'pano{i}_mask_raw'])
        assert np.array_equal(info['mask'].cpu().numpy(), g[f'pano{i}_mask'])
        assert np.array_equal(info['sup_distances'].cpu().numpy(), g[f'pano{i}_sup_distances'])
        assert np.array_equal(info['sup_colors'].cpu().numpy(), g[f'pano{i}_sup_colors'])
        assert np.array_equal(info['sup_positions'].cpu().numpy(), g[f'pano{i}_sup_positions'])
        assert np.abs(info['sup_dirs'].cpu().numpy() - g[f'pano{i}_sup_dirs']).max() < 2e-6
        assert np.array_equal(info['sup_normals'].cpu().numpy(), g[f'pano{i}_sup_normals'])
    assert len(pool) == len(g['pano0_sup_distances']) + len(g['pano1_sup_distances'])
    rays = gen_pano_rays(torch.from_numpy(g['probe_pose']), h, w)
    dist = torch.from_numpy(g['probe_distance']).cuda()
    from perf_amd.visibility import geo_check, pano_visibility_mask
    vis = pano_visibility_mask(rays.o, rays.d, dist, pool.sup_infos).cpu().numpy()
    chk = geo_check(rays.o, rays.d, dist[..., None], pool.sup_infos).cpu().numpy()
    # a point may sit on the depth threshold (two fp32 distances are compared); the morphology spreads such a pixel
    bad_v, bad_c = float((vis != g['visibility_mask']).mean()), float((chk != g['geo_check']).mean())
    print(f'[visibility] mismatching pixels: visibility {bad_v:.4f}, geo_check {bad_c:.4f}')
    assert bad_v < 0.01 and bad_c < 0.01


def test_unmodified_runner_imports_reach_the_mirrors(tmp_path):
    """perf_amd.install_shims(scene=True): `from modules.scene.nerf import NeRFScene` / `from modules.dataset.sup_info import
    SupInfoPool` -- the import statements of core_exp_runner.py:20,24 -- resolve to this package's mirrors even though a
    `modules` tree with files of those names is on sys.path (here a decoy tree whose files raise when executed; in the build
    container tools/check_reference_imports.py does the same against the real reference tree), everything else of that tree
    imports as it is, and the runner's call sequence (core_exp_runner.py:64,77-83,111-113,137-139,174-175,220) runs through
    those names: construct with the runner's keywords, register, fit, render, visibility mask, geo_check, register again,
    fit, checkpoint round trip."""
    import importlib
    import sys
    from types import SimpleNamespace
    import perf_amd
    from perf_amd import synthetic
    root = tmp_path / 'tree'
    for sub in ('modules', 'modules/scene', 'modules/dataset'):
        (root / sub).mkdir(parents=True)
    (root / 'modules' / '__init__.py').write_text('')
    boom = "raise ImportError('the decoy file was executed: the finder did not win')\n"
    for f in ('scene/nerf.py', 'scene/nerf_renderer.py', 'dataset/sup_info.py'):
        (root / 'modules' / f).write_text(boom)
    (root / 'modules' / 'dataset' / 'dataset.py').write_text("MARK = 'decoy dataset module'\n")
    saved = {k: v for k, v in sys.modules.items() if k == 'modules' or k.startswith('modules.')}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, str(root))
    try:
        perf_amd.install_shims(scene=True)
        from modules.scene.nerf import NeRFScene                      # core_exp_runner.py:24
        from modules.dataset.sup_info import SupInfoPool               # :20
        from modules.scene.nerf_renderer import NeRFOCCRenderer
        import perf_amd.scene as mirror
        assert NeRFScene is mirror.NeRFScene and SupInfoPool is mirror.SupInfoPool
        assert importlib.import_module('modules.dataset.dataset').MARK == 'decoy dataset module'     # the rest of the tree is untouched
        opt = lambda: SimpleNamespace(init_lr=0.0, peak_lr=1e-2, peak_at=0.2, lr_alpha=1e-2)
        train_conf = SimpleNamespace(raw_phase_iter_geo=60, raw_phase_iter_app=40, geo_optimizer=opt(), app_optimizer=opt(),
                                     color_loss_weight=1., depth_loss_weight=1., density_loss_weight=0., distortion_loss_weight=0.1,
                                     pixel_loss_batch_size=2048)
        torch.manual_seed(0)
        scene = NeRFScene(str(tmp_path / 'exp'), train_conf=train_conf, estimator_type='occ',
                          renderer_conf={'max_radius': 2, 'bg_color': 'rand_noise'})       # :64 with configs/nerf.yaml:24-29
        assert isinstance(scene.renderer, NeRFOCCRenderer)
        H, W = 64, 128
        pose0 = torch.eye(4)
        rays0 = mirror.gen_pano_rays(pose0, H, W)
        dist0, rgb0 = synthetic.room_with_box(rays0.o, rays0.d)
        pool = SupInfoPool()
        pool.register_sup_info(pose=pose0, mask=torch.ones([H, W], device='cuda'), rgb=rgb0, distance=dist0, normal=None)   # :77-82
        pool.gen_occ_grid(256)                                                                # :83
        scene.fit(pool)                                                                       # :111
        out = scene.render(mirror.gen_pano_rays(pose0, 32, 64), query_keys=['rgb', 'distance'])   # :113
        assert out['rgb'].shape == (32, 64, 3) and torch.isfinite(out['distance']).all()
        pose1 = torch.eye(4); pose1[:3, 3] = torch.tensor([0.2, 0.1, 0.0])
        rays1 = mirror.gen_pano_rays(pose1, H, W)
        visi = scene.get_pano_visibility_mask(pool, rays1)                                    # :137
        res = scene.render(rays1, query_keys=['rgb', 'distance'])                              # :139
        d1, c1 = synthetic.room_with_box(rays1.o, rays1.d)
        ok = pool.geo_check(rays1, d1)                                                        # :155
        # (geo_check passes only points strictly in FRONT of every registered surface, sup_info.py:261-302: ground truth that lies
        #  on surfaces the first panorama saw is a conflict -- a small fraction here, as in tools/mini_perf_loop.py)
        assert visi.shape[:2] == (H, W) and 0.0 < float(visi.float().mean()) <= 1.0 and ok.shape[:2] == (H, W) and 0.0 <= float(ok.float().mean()) <= 1.0
        sup_mask = (1. - visi.reshape(H, W).float())
        n_before = len(pool)
        pool.register_sup_info(pose=pose1, mask=sup_mask, rgb=c1, distance=d1, normal=None)   # :174
        assert len(pool) > n_before and res['rgb'].shape == (H, W, 3)
        scene.fit(pool)                                                                       # :175
        sd = scene.state_dict()                                                               # :250
        torch.save({'scene': sd, 'phase': 1}, tmp_path / 'ckpt.pth')
        scene2 = NeRFScene(str(tmp_path / 'exp2'), train_conf=train_conf, estimator_type='occ',
                           renderer_conf={'max_radius': 2, 'bg_color': 'rand_noise'})
        scene2.load_state_dict(torch.load(tmp_path / 'ckpt.pth', map_location='cuda')['scene'])   # :220
        a = scene.render(rays0, query_keys=['rgb']); b = scene2.render(rays0, query_keys=['rgb'])
        assert torch.equal(a['rgb'], b['rgb'])
        assert mirror.psnr(a['rgb'], rgb0) > 20.0                       # (100 iterations: it trains)
    finally:
        perf_amd.uninstall_scene_shims()
        sys.path.remove(str(root))
        for k in [k for k in sys.modules if k == 'modules' or k.startswith('modules.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_occupancy_update_kernels_follow_nerfacc_rule():
    """OccGridEstimator.update_every_n_steps in its warm-up phase (what the reference's NeRFScene calls 256 times per episode
    with a look-up closure, nerf.py:147-168): the three launches around the closure (perf_occ_jitter_points,
    perf_occ_ema_update, perf_occ_threshold) against the rule restated in torch on the SAME points -- occs = max(occs *
    decay, occ), binaries = occs > min(mean(occs), occ_thre) --, jitter inside its cell, chunking invisible, and the
    reference's 256-call warm-up leaves binaries == pre_grid up to a handful of cells at occupied boundaries."""
    from perf_amd import ops
    from perf_amd.nerfacc_impl import OccGridEstimator
    res = 64
    est = OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=res, levels=1).cuda()
    est.train()
    est.UPDATE_CHUNK = 50000                                   # several ragged chunks
    seen = []

    def closure(x):
        seen.append(x.clone())
        return (x.norm(dim=-1) < 0.6).float() * 0.7

    torch.manual_seed(3)
    est.update_every_n_steps(step=0, occ_eval_fn=closure, occ_thre=1e-2, ema_decay=0.1, warmup_steps=256, n=1)
    x = torch.cat(seen)
    assert x.shape == (res ** 3, 3)
    idx = torch.arange(res ** 3, device='cuda')
    cell = torch.stack([idx // (res * res), (idx // res) % res, idx % res], -1).float()
    u = (x + 1.0) * 0.5 * res - cell                          # position inside the cell
    assert float(u.min()) >= -1e-4 and float(u.max()) <= 1.0 + 1e-4 and 0.45 < float(u.mean()) < 0.55 and float(u.std()) > 0.25
    occ = (x.norm(dim=-1) < 0.6).float() * 0.7
    occs_ref = torch.maximum(torch.zeros_like(occ) * 0.1, occ)
    assert torch.equal(est.occs, occs_ref)
    thre = min(float(occs_ref.double().mean()), 1e-2)
    assert torch.equal(est.binaries.reshape(-1), occs_ref > thre)
    # a second call: the moving maximum decays what is not confirmed; another draw of the jitter
    seen.clear()
    est.update_every_n_steps(step=1, occ_eval_fn=lambda p: torch.zeros(p.shape[0], device=p.device), occ_thre=1e-2, ema_decay=0.1,
                             warmup_steps=256, n=1)
    assert torch.allclose(est.occs, occs_ref * 0.1)
    # the reference's warm-up on a pre-grid (ema 0.1, 0/1 results): binaries == pre_grid but for boundary cells the jitter can reach
    pre = (torch.rand(res ** 3, device='cuda') < 0.02)
    est2 = OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=res, levels=1).cuda(); est2.train()

    def occ_eval_fn(p):                                       # nerf.py:149-158
        q = ((p.clip(-0.999, 0.999) * .5 + .5) * res).to(torch.int64)
        return pre[q[..., 0] * res * res + q[..., 1] * res + q[..., 2]].float()
    for i in range(8):
        est2.update_every_n_steps(step=i, occ_eval_fn=occ_eval_fn, occ_thre=1e-2, ema_decay=0.1, warmup_steps=256, n=1)
    diff = int((est2.binaries.reshape(-1) != pre).sum())
    assert int((est2.binaries.reshape(-1) & ~pre).sum()) == diff and diff <= 0.02 * int(pre.sum()) + 8, diff     # only additions, a handful
