"""Generate golden vectors by IMPORTING the reference (PeRF) in the build container.

Run only where /root/reference exists (never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py

The reference's Python never travels; only the small .npz files written next to this
script do.  sys.modules stubs satisfy *import statements* of packages that are not
installed here (cv2, kornia, trimesh, icecream, tensorboard, tinycudann, nerfacc,
torch_efficient_distloss); no reference logic is replaced.  For the renderer-glue
fixture the stubbed `nerfacc` functions and the fake field are backed by this repo's
oracle, so what is pinned is the reference's *glue* (nerf_renderer.py:112-209) around
operators whose own arithmetic is third-party and unpinned (see oracle header).
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import perf_oracle as O  # noqa: E402

REF = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    _stub('trimesh'); _stub('trimesh.creation', icosphere=None)
    # cv2 / kornia are absent: the two operators the pinned functions take from them are restated HERE (not imported from
    # perf_amd): OpenCV's MORPH_ELLIPSE rasterisation, kornia's flat binary dilation / erosion with geodesic borders
    # (scipy.ndimage) and kornia.filters.laplacian (normalised 3x3 kernel, reflect border).  What the fixtures pin is the
    # reference's own composition around them (nerf.py:321-358, sup_info.py:27-120).
    _stub('cv2', COLORMAP_JET=2, MORPH_ELLIPSE=2, getStructuringElement=_cv_structuring_element)
    _stub('kornia'); _stub('kornia.morphology', erosion=_kornia_erosion, dilation=_kornia_dilation)
    sys.modules['kornia'].morphology = sys.modules['kornia.morphology']
    _stub('kornia.filters', laplacian=_kornia_laplacian)
    sys.modules['kornia'].filters = sys.modules['kornia.filters']
    _stub('icecream', ic=print)
    _stub('tinycudann')
    _stub('torch_efficient_distloss', flatten_eff_distloss=O.flatten_eff_distloss, eff_distloss=None)
    tb = _stub('torch.utils.tensorboard', SummaryWriter=object)
    # nerfacc surface, backed by the oracle (used only by the renderer-glue fixture)
    na = _stub('nerfacc')
    _stub('nerfacc.estimators')
    _stub('nerfacc.estimators.prop_net', PropNetEstimator=object)
    _stub('nerfacc.estimators.occ_grid', OccGridEstimator=object)

    def render_weight_from_density(t_starts, t_ends, sigmas, ray_indices=None, n_rays=None):
        packed = O.packed_info_from_ray_indices(ray_indices.numpy(), _CTX['n_rays'])
        return O.render_weight_from_density(t_starts, t_ends, sigmas, packed)

    def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
        return O.accumulate_along_rays(weights, values, ray_indices, n_rays)

    na.render_weight_from_density = render_weight_from_density
    na.accumulate_along_rays = accumulate_along_rays
    na.render_transmittance_from_alpha = None
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


_CTX = {}


def _cv_structuring_element(shape, ksize):
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (cols, rows)) as OpenCV rasterises it: row i spans c - dx .. c + dx with
    dx = cvRound(c * sqrt((r^2 - dy^2) / r^2))."""
    assert shape == 2
    cols, rows = ksize
    r, c = rows // 2, cols // 2
    k = np.zeros((rows, cols), np.uint8)
    inv_r2 = 1.0 / (r * r) if r else 0.0
    for i in range(rows):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            k[i, max(c - dx, 0):min(c + dx + 1, cols)] = 1
    return k


def _kornia_dilation(x, kernel):
    """kornia.morphology.dilation of a [1,C,H,W] 0/1 image with a flat kernel: max over the support, outside = background."""
    import scipy.ndimage as ndi
    fp = kernel.numpy() > 0.5
    out = np.stack([ndi.grey_dilation(ch, footprint=fp, mode='constant', cval=0.0) for ch in x[0].numpy()])
    return torch.from_numpy(out)[None].float()


def _kornia_erosion(x, kernel):
    """kornia.morphology.erosion: min over the support, outside = foreground (geodesic border)."""
    import scipy.ndimage as ndi
    fp = kernel.numpy() > 0.5
    out = np.stack([ndi.grey_erosion(ch, footprint=fp, mode='constant', cval=1.0) for ch in x[0].numpy()])
    return torch.from_numpy(out)[None].float()


def _kornia_laplacian(x, kernel_size=3):
    """kornia.filters.laplacian(x, 3): [[1,1,1],[1,-8,1],[1,1,1]] / 16, reflect border."""
    assert kernel_size == 3
    k = torch.ones(3, 3); k[1, 1] = -8.0
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1), mode='reflect')
    return torch.nn.functional.conv2d(xp, (k / 16.0)[None, None].repeat(x.shape[1], 1, 1, 1), groups=x.shape[1])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fx_rays(cu):
    out = {}
    poses = {'eye': torch.eye(4)}
    ang = 0.7
    R = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=torch.float32)
    R2 = torch.tensor([[1, 0, 0], [0, np.cos(.3), -np.sin(.3)], [0, np.sin(.3), np.cos(.3)]], dtype=torch.float32)
    P = torch.eye(4); P[:3, :3] = R @ R2; P[:3, 3] = torch.tensor([0.2, -0.1, 0.05])
    poses['rt'] = P
    for name, pose in poses.items():
        out[f'pose_{name}'] = pose.numpy()
        r = cu.gen_pano_rays(pose, 32, 64)
        out[f'pano_{name}_32x64_o'] = r.o.numpy(); out[f'pano_{name}_32x64_d'] = r.d.numpy()
        for (h, w) in [(256, 512), (1024, 2048)]:
            r = cu.gen_pano_rays(pose, h, w)
            d = r.d.numpy()
            g = np.random.RandomState(0)
            ii = g.randint(0, h, 64); jj = g.randint(0, w, 64)
            out[f'pano_{name}_{h}x{w}_ij'] = np.stack([ii, jj], -1)
            out[f'pano_{name}_{h}x{w}_d'] = d[ii, jj]
            out[f'pano_{name}_{h}x{w}_sha'] = np.frombuffer(bytes.fromhex(sha(d)), np.uint8)
        r = cu.gen_pers_rays(pose, np.deg2rad(75.), 64)
        out[f'pers_{name}_o'] = r.o.numpy(); out[f'pers_{name}_d'] = r.d.numpy()
    # round trip direction -> image coordinate
    r = cu.gen_pano_rays(torch.eye(4), 32, 64)
    out['pano_eye_32x64_imgcoord'] = cu.direction_to_img_coord(r.d).numpy()
    np.savez_compressed(os.path.join(HERE, 'rays.npz'), **out)


def fx_field_bits(ngp):
    x = torch.tensor([0., 1., 15., 20., -3.], requires_grad=True)
    y = ngp.trunc_exp(x)
    y.sum().backward()
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(1000, 3, generator=g) * 6 - 3)
    aabb = torch.tensor([-1., -1, -1, 1, 1, 1])
    c = ngp.contract_to_unisphere(pts.clone(), aabb)
    np.savez_compressed(os.path.join(HERE, 'field_bits.npz'), te_x=x.detach().numpy(), te_y=y.detach().numpy(),
                        te_g=x.grad.numpy(), ct_x=pts.numpy(), ct_y=c.numpy())


def _room_pool(cu, h, w):
    rays = cu.gen_pano_rays(torch.eye(4), h, w)
    dist, rgb = O.synthetic_room(rays.d)
    return rays, dist, rgb


def fx_sup(cu, si):
    out = {}
    for (h, w, res) in [(16, 32, 64), (64, 128, 256)]:
        rays, dist, rgb = _room_pool(cu, h, w)
        pool = si.SupInfoPool()
        pool.all_sup_rays = cu.Rays(rays.o.reshape(-1, 3), rays.d.reshape(-1, 3))
        pool.all_sup_distances = dist.reshape(-1, 1)
        pool.all_sup_colors = rgb.reshape(-1, 3)
        pool.all_sup_normals = torch.zeros(h * w, 3)
        occ, pts = pool.gen_occ_grid(res)
        out[f'occ_{h}x{w}_r{res}_idx'] = torch.where(occ > 0)[0].numpy().astype(np.int32)
        if res == 64:
            torch.manual_seed(0)
            r, c, dd, nn = pool.rand_ray_color_data(8192)
            torch.manual_seed(0)
            idx = torch.randint(0, h * w, (8192,))
            assert torch.equal(r.d, pool.all_sup_rays.d[idx])
            out['rand_idx_seed0_n512_b8192'] = idx.numpy().astype(np.int32)
            out['rand_dist_first16'] = dd[:16].numpy()
    np.savez_compressed(os.path.join(HERE, 'sup.npz'), **out)


def fx_lr(nerf_mod):
    conf = types.SimpleNamespace(init_lr=0.0, peak_lr=1e-2, peak_at=0.2, lr_alpha=1e-2)
    conf2 = types.SimpleNamespace(init_lr=1e-4, peak_lr=1e-3, peak_at=0.1, lr_alpha=0.1)
    prog = np.linspace(0, 0.9999, 41)
    vals = []
    for c in (conf, conf2):
        row = []
        for p in prog:
            opt = types.SimpleNamespace(param_groups=[{'lr': None}])
            nerf_mod.NeRFScene.update_lr(None, opt, c, float(p))
            row.append(opt.param_groups[0]['lr'])
        vals.append(row)
    np.savez_compressed(os.path.join(HERE, 'lr.npz'), progress=prog, lr=np.array(vals, np.float64),
                        conf=np.array([[0.0, 1e-2, 0.2, 1e-2], [1e-4, 1e-3, 0.1, 0.1]]))


def fx_pose(ps, cu):
    g = torch.Generator().manual_seed(0)
    dm = 0.3 + 0.2 * torch.rand(64, 128, generator=g)
    s = ps.CirclePoseSampler(dm, [.2, .4, .6], [8, 8, 8])
    anchors = torch.stack([s.sample_pose(i) for i in range(s.n_poses)]).numpy()
    np.random.seed(0)
    dense = ps.DenseTravelPoseSampler(s, 180)
    dposes = torch.stack([dense.sample_pose(i) for i in range(dense.n_poses)]).numpy()
    np.savez_compressed(os.path.join(HERE, 'poses.npz'), distance_map=dm.numpy(), anchors=anchors, dense=dposes)


def fx_render_glue(rend_mod, cu):
    """Run the reference's NeRFOCCRenderer.render (nerf_renderer.py:112-209) on oracle operators."""
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    gs, as_ = O.geo_spec(), O.app_spec()
    geo = O.init_field_params(gs, 1337); app = O.init_field_params(as_, 4242)
    # make the fields non-trivial: grid entries U(-1,1) instead of U(-1e-4,1e-4)
    geo[gs.n_net:] *= 1e4; app[as_.n_net:] *= 1e4
    res = 32
    rays = cu.gen_pano_rays(torch.eye(4), 8, 16)
    o = rays.o.reshape(-1, 3).contiguous(); d = rays.d.reshape(-1, 3).contiguous()
    dist, _ = O.synthetic_room(d)
    occ = O.gen_occ_grid(o, d, dist, res).reshape(res, res, res).bool().numpy()
    R = o.shape[0]
    step = 4e-3
    out = {'o': o.numpy(), 'd': d.numpy(), 'binaries': np.packbits(occ.reshape(-1)), 'res': res, 'step': step,
           'geo_seed': 1337, 'app_seed': 4242, 'grid_gain': 1e4}

    class FakeNerf:
        training = True

        def query_density(self, x):
            return O.query_density(x, geo, gs, torch.from_numpy(aabb))

        def query_rgb(self, x):
            return O.query_rgb(x, app, as_, torch.from_numpy(aabb))

    class FakeEstimator:
        def sampling(self, rays_o, rays_d, sigma_fn=None, near_plane=0., far_plane=1e10, render_step_size=1e-3,
                     stratified=False, cone_angle=0., alpha_thre=0.):
            # PeRF passes step 5e-4 / far 1.5; the fixture shrinks the problem, same code path.
            t0 = np.full(R, near_plane, np.float32)
            if stratified:
                t0 = (t0 + _CTX['jitter'] * np.float32(step)).astype(np.float32)
            ri, ts, te, packed = O.occ_march(rays_o.numpy(), rays_d.numpy(), occ, aabb, near_plane, 1.5, step, t0)
            ri_t, ts_t, te_t = torch.from_numpy(ri), torch.from_numpy(ts), torch.from_numpy(te)
            with torch.no_grad():
                sig = sigma_fn(ts_t, te_t, ri_t)
            keep, _ = O.visibility_keep_mask(sig.numpy(), ts, te, packed, 1e-4)
            keep = torch.from_numpy(keep)
            return ri_t[keep], ts_t[keep], te_t[keep]

    rend = rend_mod.NeRFOCCRenderer(max_radius=2, bg_color='rand_noise')
    _CTX['n_rays'] = R
    for mode in ('train', 'eval'):
        nerf = FakeNerf(); nerf.training = (mode == 'train')
        g = torch.Generator().manual_seed(7)
        _CTX['jitter'] = torch.rand(R, generator=g).numpy()
        torch.manual_seed(11)
        res_d = rend.render(nerf, FakeEstimator(), o, d, torch.zeros(R, 1), torch.ones(R, 1),
                            geo_inference=False, app_inference=True)
        torch.manual_seed(11)
        bg = torch.rand(R, 3); noise = torch.rand(R, 1)
        out[f'{mode}_jitter'] = _CTX['jitter']; out[f'{mode}_bg'] = bg.numpy(); out[f'{mode}_noise'] = noise.numpy()
        for k in ('rgb', 'distance', 'weights', 'opacities', 'trans', 't_starts', 't_ends', 'ray_indices'):
            out[f'{mode}_{k}'] = res_d[k].detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'render_glue.npz'), **out)


def fx_train_glue(nerf_mod, rend_mod, si, cu):
    """The reference's own NeRFScene.train_one_step_geo / train_one_step_app (modules/scene/nerf.py:186-297) executed --
    unbound, on a stand-in `self` -- over oracle-backed operators: its batch draw (the reference's SupInfoPool), its
    render_once / NeRFOCCRenderer.render, its loss assembly, its GradScaler.scale(loss).backward(), a recording optimizer.
    Pins what the oracle's geo_step_loss / app_step_loss restate: loss values and the gradient the optimizer sees."""
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    gs, as_ = O.geo_spec(), O.app_spec()
    res, step, B = 32, 4e-3, 16
    h, w = 16, 32
    rays, dist_map, rgb_map = _room_pool(cu, h, w)
    pool = si.SupInfoPool()
    pool.all_sup_rays = cu.Rays(rays.o.reshape(-1, 3), rays.d.reshape(-1, 3))
    pool.all_sup_distances = dist_map.reshape(-1, 1)
    pool.all_sup_colors = rgb_map.reshape(-1, 3)
    pool.all_sup_normals = torch.zeros(h * w, 3)
    occ_u8, _ = pool.gen_occ_grid(res)
    occ = occ_u8.reshape(res, res, res).bool().numpy()
    out = {'h': h, 'w': w, 'res': res, 'step': step, 'batch': B, 'geo_seed': 1337, 'app_seed': 4242, 'grid_gain': 1e4,
           'loss_scale': 128.0}

    class Recorder:                                   # torch.optim.Adam's place: sees what optimizer.step() would see
        param_groups = [{'lr': 0.0}]

        def __init__(self, p): self.p = p; self.grad = None
        def zero_grad(self): self.p.grad = None
        def step(self): self.grad = self.p.grad.detach().clone()

    class Scaler:                                     # torch.cuda.amp.GradScaler(2**7).scale(): outputs * scale (nerf.py:139)
        def scale(self, loss): return loss * 128.0

    class Writer:
        def __init__(self): self.rec = {}
        def add_scalar(self, tag, val, step): self.rec[tag] = float(val)

    for kind, progress in (('geo', 0.2), ('geo', 0.8), ('app', 0.5)):
        geo = O.init_field_params(gs, 1337); app = O.init_field_params(as_, 4242)
        geo[gs.n_net:] *= 1e4; app[as_.n_net:] *= 1e4
        # a density field with some opacity, so that the early stop prunes and the distance loss has a gradient
        geo[:gs.n_net] *= 3.0
        geo.requires_grad_(kind == 'geo'); app.requires_grad_(kind == 'app')

        class FakeNerf:
            training = True
            def query_density(self, x): return O.query_density(x, geo, gs, torch.from_numpy(aabb))
            def query_rgb(self, x): return O.query_rgb(x, app, as_, torch.from_numpy(aabb))

        class FakeEstimator:
            def sampling(self, rays_o, rays_d, sigma_fn=None, near_plane=0., far_plane=1e10, render_step_size=1e-3,
                         stratified=False, cone_angle=0., alpha_thre=0.):
                R = rays_o.shape[0]
                t0 = (np.full(R, near_plane, np.float32) + _CTX['jitter'] * np.float32(step)).astype(np.float32)
                ri, ts, te, packed = O.occ_march(rays_o.numpy(), rays_d.numpy(), occ, aabb, near_plane, 1.5, step, t0)
                ri_t, ts_t, te_t = torch.from_numpy(ri), torch.from_numpy(ts), torch.from_numpy(te)
                with torch.no_grad():
                    sig = sigma_fn(ts_t, te_t, ri_t)
                keep, _ = O.visibility_keep_mask(sig.numpy(), ts, te, packed, 1e-4)
                keep = torch.from_numpy(keep)
                return ri_t[keep], ts_t[keep], te_t[keep]

        me = types.SimpleNamespace(
            train_conf=types.SimpleNamespace(pixel_loss_batch_size=B, depth_loss_weight=1.0, distortion_loss_weight=0.1,
                                             density_loss_weight=0.0, color_loss_weight=1.0),
            renderer=rend_mod.NeRFOCCRenderer(max_radius=2, bg_color='rand_noise'), nerf=FakeNerf(), estimator=FakeEstimator(),
            writer=Writer(), global_iter_step_geo=0, global_iter_step_app=0)
        for name in ('render_once', 'to_bounded_rays', 'need_to_update_occ'):
            setattr(me, name, types.MethodType(getattr(nerf_mod.NeRFScene, name), me))
        _CTX['n_rays'] = B
        g = torch.Generator().manual_seed(21)
        _CTX['jitter'] = torch.rand(B, generator=g).numpy()
        rec = Recorder(geo if kind == 'geo' else app)
        torch.manual_seed(5)
        fn = nerf_mod.NeRFScene.train_one_step_geo if kind == 'geo' else nerf_mod.NeRFScene.train_one_step_app
        fn(me, rec, pool, 'by_all_pixels', progress, Scaler())
        # the draws the step made, in its order: batch indices (sup_info.py:247), background colour and distance noise
        # (nerf_renderer.py:185,193)
        torch.manual_seed(5)
        idx = torch.randint(0, h * w, (B,)); bg = torch.rand(B, 3); noise = torch.rand(B, 1)
        tag = f'{kind}_p{int(progress * 10)}'
        out[f'{tag}_idx'] = idx.numpy(); out[f'{tag}_bg'] = bg.numpy(); out[f'{tag}_noise'] = noise.numpy()
        out[f'{tag}_jitter'] = _CTX['jitter']; out[f'{tag}_progress'] = progress
        gr = rec.grad.numpy()                          # sparse: a 16-ray batch touches a fraction of the 6.6 M table entries
        out[f'{tag}_grad_numel'] = gr.size
        out[f'{tag}_grad_norm'] = float(np.sqrt((gr.astype(np.float64) ** 2).sum()))
        pg = np.random.RandomState(123).standard_normal((3, gr.size)).astype(np.float32)
        out[f'{tag}_grad_proj'] = (pg.astype(np.float64) @ gr.astype(np.float64))      # three fixed random projections
        if tag != 'geo_p8':                            # (the second geometry case differs by the distortion ramp only)
            nz = np.nonzero(gr)[0]
            out[f'{tag}_grad_idx'] = nz.astype(np.int32); out[f'{tag}_grad_val'] = gr[nz]
        for k, v in me.writer.rec.items():
            out[f'{tag}_{k.split("/")[-1]}'] = v
        assert (me.global_iter_step_geo, me.global_iter_step_app) == ((1, 0) if kind == 'geo' else (0, 1))
    np.savez_compressed(os.path.join(HERE, 'train_glue.npz'), **out)


def fx_visibility(nerf_mod, si, cu):
    """The reference's own NeRFScene.get_pano_visibility_mask (nerf.py:321-358) and SupInfoPool.geo_check
    (sup_info.py:261-302) on two registered panoramas and a probe panorama whose rendered distance is given; and
    PanoSupInfo.__init__ / update_sup_info (sup_info.py:27-120): validity rules and the supervision rays they select."""
    h, w = 48, 96
    out = {'h': h, 'w': w}
    poses = []
    for t in ([0.0, 0.0, 0.0], [0.25, -0.1, 0.05]):
        P = torch.eye(4); P[:3, 3] = torch.tensor(t); poses.append(P)
    ang = 0.4
    poses[1][:3, :3] = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=torch.float32)
    infos = []
    g = torch.Generator().manual_seed(9)
    for i, P in enumerate(poses):
        rays = cu.gen_pano_rays(P, h, w)
        # distance to the walls of the room from this pose (analytic), a depth edge (a floating slab), a hole in the mask
        half = torch.tensor([0.9, 0.7, 0.5]) / 1.05
        o, d = rays.o, rays.d
        tt = torch.where(d > 0, (half - o) / d.clamp_min(1e-9), (-half - o) / d.clamp_max(-1e-9))
        dist = tt.min(-1, keepdim=True).values
        dist[10:16, 20:30] *= 0.6
        rgb = torch.rand(h, w, 3, generator=g)
        mask = torch.ones(h, w, 1); mask[2:6, 40:50] = 0.0
        # normals of the walls: the axis whose plane is hit, pointing inwards
        ax = tt.argmin(-1)
        normal = -torch.nn.functional.one_hot(ax, 3).float() * torch.sign(d.gather(-1, ax[..., None]))
        out[f'pano{i}_pose'] = P.numpy(); out[f'pano{i}_distance'] = dist.numpy(); out[f'pano{i}_rgb'] = rgb.numpy()
        out[f'pano{i}_mask_in'] = mask.numpy(); out[f'pano{i}_normal'] = normal.numpy()
        info = si.PanoSupInfo(P, mask, rgb, dist, normal)
        infos.append(info)
        out[f'pano{i}_mask_raw'] = info.mask_raw.numpy(); out[f'pano{i}_mask'] = info.mask.numpy()
        out[f'pano{i}_sup_colors'] = info.sup_colors.numpy(); out[f'pano{i}_sup_distances'] = info.sup_distances.numpy()
        out[f'pano{i}_sup_dirs'] = info.sup_dirs.numpy(); out[f'pano{i}_sup_positions'] = info.sup_positions.numpy()
        out[f'pano{i}_sup_normals'] = info.sup_normals.numpy()
    pool = si.SupInfoPool()
    pool.sup_infos = infos
    # probe panorama: a third pose; "rendered" distance = analytic distance with a bump that hides part of it
    P = torch.eye(4); P[:3, 3] = torch.tensor([-0.2, 0.15, -0.05])
    rays = cu.gen_pano_rays(P, h, w)
    half = torch.tensor([0.9, 0.7, 0.5]) / 1.05
    tt = torch.where(rays.d > 0, (half - rays.o) / rays.d.clamp_min(1e-9), (-half - rays.o) / rays.d.clamp_max(-1e-9))
    dist = tt.min(-1).values * 0.98  # a rendered surface sits a little in front of the true wall
    dist[4:22, 8:36] *= 1.3          # behind the wall: occluded for every registered panorama (and no conflict for geo_check)
    dist[28:40, 50:74] *= 0.5        # in free space in front of the wall
    me = types.SimpleNamespace(render=lambda r, query_keys=None: {'distance': dist[..., None].clone()})
    vis = nerf_mod.NeRFScene.get_pano_visibility_mask(me, pool, rays)
    chk = pool.geo_check(rays, dist[..., None].clone())
    out['probe_pose'] = P.numpy(); out['probe_distance'] = dist.numpy()
    out['visibility_mask'] = vis.numpy(); out['geo_check'] = chk.numpy()
    np.savez_compressed(os.path.join(HERE, 'visibility.npz'), **out)


def main():
    assert os.path.isdir(REF), 'reference tree not present: fixtures can only be made in the build container'
    install_stubs()
    from utils import camera_utils as cu
    from modules.fields import ngp_nerf as ngp
    from modules.dataset import sup_info as si
    from modules import pose_sampler as ps
    from modules.scene import nerf_renderer as rend_mod
    from modules.scene import nerf as nerf_mod
    fx_rays(cu)
    fx_field_bits(ngp)
    fx_sup(cu, si)
    fx_lr(nerf_mod)
    fx_pose(ps, cu)
    fx_render_glue(rend_mod, cu)
    fx_train_glue(nerf_mod, rend_mod, si, cu)
    fx_visibility(nerf_mod, si, cu)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == '__main__':
    main()
