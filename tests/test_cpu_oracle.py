"""CPU tests (no GPU): the oracle against the golden vectors made from the reference, and the
C-ABI library's loadability / exported symbols."""
import ctypes
import os
import re
import sys

import numpy as np
import torch

from oracle import perf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')


def test_rays_match_reference():
    g = np.load(f'{G}/rays.npz')
    for name in ('eye', 'rt'):
        pose = torch.from_numpy(g[f'pose_{name}'])
        o, d = O.pano_rays(pose, 32, 64)
        assert np.abs(d.numpy() - g[f'pano_{name}_32x64_d']).max() < 1e-6
        assert np.array_equal(o.numpy(), g[f'pano_{name}_32x64_o'])
        for (h, w) in ((256, 512), (1024, 2048)):
            _, d = O.pano_rays(pose, h, w)
            ij = g[f'pano_{name}_{h}x{w}_ij']
            assert np.abs(d.numpy()[ij[:, 0], ij[:, 1]] - g[f'pano_{name}_{h}x{w}_d']).max() < 1e-6
        o, d = O.pers_rays(pose, np.deg2rad(75.), 64)
        assert np.abs(d.numpy() - g[f'pers_{name}_d']).max() < 1e-6
        assert np.abs(o.numpy() - g[f'pers_{name}_o']).max() == 0


def test_trunc_exp_and_contract():
    g = np.load(f'{G}/field_bits.npz')
    x = torch.from_numpy(g['te_x']).requires_grad_(True)
    y = O.trunc_exp(x)
    y.sum().backward()
    assert np.allclose(y.detach().numpy(), g['te_y'], rtol=1e-6) and np.allclose(x.grad.numpy(), g['te_g'], rtol=1e-6)
    c = O.contract_to_unisphere(torch.from_numpy(g['ct_x']), torch.tensor([-1., -1, -1, 1, 1, 1]))
    assert np.abs(c.numpy() - g['ct_y']).max() < 1e-6


def test_occ_grid_and_batch_sampler():
    g = np.load(f'{G}/sup.npz')
    for (h, w, res) in ((16, 32, 64), (64, 128, 256)):
        o, d = O.pano_rays(torch.eye(4), h, w)
        dist, _ = O.synthetic_room(d)
        occ = O.gen_occ_grid(o.reshape(-1, 3), d.reshape(-1, 3), dist.reshape(-1, 1), res)
        assert np.array_equal(torch.where(occ > 0)[0].numpy(), g[f'occ_{h}x{w}_r{res}_idx'])
    torch.manual_seed(0)
    idx = O.rand_ray_indices(16 * 32, 8192)
    assert np.array_equal(idx.numpy(), g['rand_idx_seed0_n512_b8192'])


def test_lr_schedule():
    g = np.load(f'{G}/lr.npz')
    for row, conf in zip(g['lr'], g['conf']):
        got = [O.lr_schedule(float(p), *conf) for p in g['progress']]
        assert np.allclose(got, row, rtol=1e-12, atol=0)


def test_renderer_glue_matches_reference():
    """oracle.occ_render == the reference's NeRFOCCRenderer.render run over the same operators."""
    g = np.load(f'{G}/render_glue.npz')
    res = int(g['res'])
    occ = np.unpackbits(g['binaries'])[:res ** 3].reshape(res, res, res).astype(bool)
    gs, as_ = O.geo_spec(), O.app_spec()
    geo = O.init_field_params(gs, int(g['geo_seed'])); app = O.init_field_params(as_, int(g['app_seed']))
    geo[gs.n_net:] *= float(g['grid_gain']); app[as_.n_net:] *= float(g['grid_gain'])
    o = torch.from_numpy(g['o']); d = torch.from_numpy(g['d'])
    R = o.shape[0]
    step = float(g['step'])
    for mode in ('train', 'eval'):
        t0 = (np.zeros(R, np.float32) + g[f'{mode}_jitter'] * np.float32(step)).astype(np.float32) if mode == 'train' else None
        out = O.occ_render(o, d, geo, app, occ, [-1, -1, -1, 1, 1, 1], training=(mode == 'train'), t0=t0,
                           bg_color=torch.from_numpy(g[f'{mode}_bg']), dist_noise=torch.from_numpy(g[f'{mode}_noise']),
                           near=0.0, far=1.5, step=step)
        assert np.array_equal(out['ray_indices'].numpy(), g[f'{mode}_ray_indices'])
        assert np.array_equal(out['t_starts'].numpy(), g[f'{mode}_t_starts'])
        for k in ('rgb', 'distance', 'weights', 'opacities', 'trans'):
            assert np.abs(out[k].detach().numpy() - g[f'{mode}_{k}']).max() < 1e-6, k


def test_canonical_scan_is_a_prefix_sum():
    rng = np.random.RandomState(0)
    counts = rng.randint(0, 200, 50); counts[0] = 0
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    packed = np.stack([starts, counts], -1).astype(np.int32)
    v = rng.rand(counts.sum()).astype(np.float32)
    ex = O.packed_exclusive_sum_canonical(v, packed)
    ref = O.packed_exclusive_sum(torch.from_numpy(v), packed).numpy()
    assert np.abs(ex - ref).max() < 1e-4


def test_library_exports_every_declared_symbol():
    from perf_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'perf_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(perf_[a-z0-9_]+)\s*\(', header)))
    assert declared == _lib.exported_symbols()
    lib = _lib.load()                      # raises if the .so is missing: no CPU fallback
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.perf_version() == _lib.ABI_VERSION == int(re.search(r'#define\s+PERF_ABI_VERSION\s+(\d+)', header).group(1))
    assert lib.perf_sizeof_grid_desc() == ctypes.sizeof(_lib.GridDesc)
    assert lib.perf_sizeof_mlp_desc() == ctypes.sizeof(_lib.MlpDesc)


def test_abi_version_is_bumped_with_every_signature_change():
    """include/perf_hip.abi.json records (PERF_ABI_VERSION, digest of the header's declarations).  A header whose
    declarations changed must carry a NEW version (then re-record with `python tools/abi_digest.py --write`): the
    load-time version check of the binding is what catches a stale libperf_hip.so."""
    import json
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import abi_digest
    now = abi_digest.digest()
    rec = json.load(open(abi_digest.RECORD))
    if now['digest'] != rec['digest']:
        assert now['version'] > rec['version'], ('include/perf_hip.h changed but PERF_ABI_VERSION is still '
                                                 f"{rec['version']}: bump it, then `python tools/abi_digest.py --write`")
        raise AssertionError('include/perf_hip.abi.json is out of date: run `python tools/abi_digest.py --write`')
    assert now['version'] == rec['version']
    from perf_amd import _lib
    assert _lib.ABI_VERSION == now['version']


def test_graft_entry_build_runs():
    """__graft_entry__.build() compiles (or finds) the library and checks header / binding / library agreement."""
    import __graft_entry__ as G
    G.build()


def test_integration_md_binding_matches_the_header():
    """The ctypes snippet INTEGRATION.md shows a maintainer is executed as written: its perf_grid_desc mirror must have the
    size the built library reports, and its argtypes for perf_hashgrid_fwd the arity of the header's declaration."""
    from perf_amd import _lib
    _lib.load()
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    code = re.search(r'```python\nimport ctypes, torch\n(.*?)```', text, re.S).group(1)
    code = code.replace('ctypes.CDLL("perf_amd/libperf_hip.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec('import ctypes, torch\n' + code, ns)                     # runs the snippet's own load-time asserts too
    assert ctypes.sizeof(ns['GridDesc']) == ctypes.sizeof(_lib.GridDesc) == ns['lib'].perf_sizeof_grid_desc()
    for (name, ctype), (name2, ctype2) in zip(ns['GridDesc']._fields_, _lib.GridDesc._fields_):
        assert name == name2 and ctypes.sizeof(ctype) == ctypes.sizeof(ctype2), name
    header = open(os.path.join(ROOT, 'include', 'perf_hip.h')).read()
    decl = re.search(r'int perf_hashgrid_fwd\((.*?)\);', header, re.S).group(1)
    assert len(ns['lib'].perf_hashgrid_fwd.argtypes) == len(decl.split(',')) == len(_lib._SIGS['perf_hashgrid_fwd'][1])


def test_binding_arity_matches_the_header():
    """Every entry point: the number of ctypes argtypes in perf_amd/_lib.py equals the number of parameters declared in
    include/perf_hip.h (a drifted signature would otherwise only show up as garbage arguments on the GPU)."""
    from perf_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'perf_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    for name, (_, args) in _lib._SIGS.items():
        m = re.search(r'\b' + name + r'\s*\((.*?)\)\s*;', header, re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ('', 'void') else len(params.split(','))
        assert n == len(args), (name, n, len(args))


def test_ops_refuse_cpu_tensors():
    import pytest
    from perf_amd import ops, _lib
    from perf_amd.grid import GridConfig
    with pytest.raises(_lib.PerfError):
        ops.hashgrid_fwd(GridConfig(), torch.rand(4, 3), torch.zeros(8, dtype=torch.bfloat16))


def test_pose_samplers_match_reference():
    """a11: the host-side pose samplers reproduce the reference's anchors and dense trajectory (golden poses.npz)."""
    from perf_amd.pose_sampler import CirclePoseSampler, DenseTravelPoseSampler
    g = np.load(f'{G}/poses.npz')
    s = CirclePoseSampler(torch.from_numpy(g['distance_map']), [.2, .4, .6], [8, 8, 8])
    anchors = torch.stack([s.sample_pose(i) for i in range(s.n_poses)]).numpy()
    assert anchors.shape == g['anchors'].shape == (24, 4, 4)
    assert np.abs(anchors - g['anchors']).max() < 1e-6
    np.random.seed(0)
    dense = DenseTravelPoseSampler(s, 180)
    poses = torch.stack([dense.sample_pose(i) for i in range(dense.n_poses)]).numpy()
    assert poses.shape == g['dense'].shape
    assert np.abs(poses - g['dense']).max() < 1e-5


def test_direction_to_img_coord_and_morphology():
    """a12 helpers: the equirect inverse mapping against the reference's golden vector; ellipse kernels / morphology
    through their defining properties (plain torch, runs on CPU)."""
    from perf_amd import visibility as V
    g = np.load(f'{G}/rays.npz')
    d = torch.from_numpy(g['pano_eye_32x64_d'])
    assert np.abs(V.direction_to_img_coord(d).numpy() - g['pano_eye_32x64_imgcoord']).max() < 1e-6
    k5 = V.ellipse_kernel(5, 5)
    assert k5.tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    m = torch.zeros(1, 1, 21, 21); m[0, 0, 10, 10] = 1
    assert torch.equal(V.dilate(m, k5)[0, 0, 8:13, 8:13], k5)                 # dilation of a point = the kernel
    full = torch.ones(1, 1, 21, 21)
    assert torch.equal(V.erode(full, V.ellipse_kernel(9, 9)), full)           # borders do not erode
    hole = full.clone(); hole[0, 0, 10, 10] = 0
    assert float(V.erode(hole, k5).sum()) == 21 * 21 - float(k5.sum())        # erosion grows a hole by the kernel


def test_config1_cpu_plumbing():
    """BASELINE config 1 (pure-PyTorch field on CPU, 32 samples/ray, no GPU): the oracle's full geometry training
    step -- sampling, both fields, compositing, losses, backward, Adam -- runs and reduces the loss."""
    torch.manual_seed(0)
    gs, as_ = O.geo_spec(), O.app_spec()
    geo = O.init_field_params(gs).requires_grad_(True); app = O.init_field_params(as_)
    o, d = O.pano_rays(torch.eye(4), 8, 16)
    o = o.reshape(-1, 3); d = d.reshape(-1, 3)
    gt, _ = O.synthetic_room(d)
    occ = np.ones((8, 8, 8), bool)
    m = torch.zeros_like(geo); v = torch.zeros_like(geo)
    losses = []
    for step in range(1, 4):
        out = O.occ_render(o, d, geo, app, occ, [-1, -1, -1, 1, 1, 1], training=True, t0=np.zeros(len(o), np.float32),
                           bg_color=torch.rand(len(o), 3), dist_noise=torch.full((len(o), 1), 0.5), near=0.0, far=10.0,
                           step=0.99 / 32, early_stop_eps=0.0, max_steps=32)
        assert out['ray_indices'].numel() == len(o) * 32
        loss, dl, _ = O.geo_step_loss(out, gt, 0.25)
        geo.grad = None
        loss.backward()
        with torch.no_grad():
            p, m, v = O.adam_step(geo, geo.grad, m, v, step, 1e-2)
            geo.copy_(p)
        losses.append(float(dl))
    assert losses[-1] < losses[0]


def test_morphology_matches_scipy():
    """next-3: ellipse elements and the border rules of dilate / erode (background outside for dilation, foreground
    outside for erosion) against scipy.ndimage on random masks — the definition the HIP morphology kernel is tested
    against on the GPU."""
    from scipy import ndimage
    from perf_amd import visibility as V
    rng = np.random.default_rng(3)
    for shape, p in (((37, 53), 0.15), ((16, 128), 0.6), ((64, 64), 0.95)):
        m = (rng.random(shape) < p)
        mt = torch.from_numpy(m.astype(np.float32))[None, None]
        for rows, cols in ((3, 3), (5, 5), (9, 9), (5, 9)):
            k = V.ellipse_kernel(rows, cols)
            kb = k.numpy() > 0.5
            assert kb[rows // 2].all() and kb[:, cols // 2].all() and (kb == kb[::-1, ::-1]).all()
            d = ndimage.binary_dilation(m, structure=kb, border_value=0)
            e = ndimage.binary_erosion(m, structure=kb, border_value=1)
            assert np.array_equal(V.dilate(mt, k)[0, 0].numpy() > 0.5, d)
            assert np.array_equal(V.erode(mt, k)[0, 0].numpy() > 0.5, e)


# ---- the reference's training-step and visibility glue (tests/golden/train_glue.npz, visibility.npz: made by running the
#      reference's own NeRFScene.train_one_step_geo / _app, get_pano_visibility_mask, SupInfoPool.geo_check and PanoSupInfo on
#      oracle-backed operators, tests/golden/make_fixtures.py) -----------------------------------------------------------------
def _train_glue_case(g, tag):
    """Inputs of one recorded step -> (o, d, gt_dist, gt_rgb, occ, geo, app, t0, bg, noise)."""
    h, w, res = int(g['h']), int(g['w']), int(g['res'])
    o_all, d_all = O.pano_rays(torch.eye(4), h, w)
    o_all = o_all.reshape(-1, 3); d_all = d_all.reshape(-1, 3)
    dist_all, rgb_all = O.synthetic_room(d_all)
    occ = O.gen_occ_grid(o_all, d_all, dist_all, res).reshape(res, res, res).bool().numpy()
    idx = torch.from_numpy(g[f'{tag}_idx'])
    gs, as_ = O.geo_spec(), O.app_spec()
    geo = O.init_field_params(gs, int(g['geo_seed'])); app = O.init_field_params(as_, int(g['app_seed']))
    geo[gs.n_net:] *= float(g['grid_gain']); app[as_.n_net:] *= float(g['grid_gain'])
    geo[:gs.n_net] *= 3.0
    t0 = (np.zeros(len(idx), np.float32) + g[f'{tag}_jitter'] * np.float32(float(g['step']))).astype(np.float32)
    return (o_all[idx].contiguous(), d_all[idx].contiguous(), dist_all.reshape(-1, 1)[idx], rgb_all.reshape(-1, 3)[idx], occ, geo, app, t0,
            torch.from_numpy(g[f'{tag}_bg']), torch.from_numpy(g[f'{tag}_noise']))


def _dense_grad(g, tag):
    out = np.zeros(int(g[f'{tag}_grad_numel']), np.float32)
    out[g[f'{tag}_grad_idx']] = g[f'{tag}_grad_val']
    return out


def test_train_step_glue_matches_reference():
    """oracle.occ_render + geo_step_loss / app_step_loss == the reference's train_one_step_geo / train_one_step_app
    (nerf.py:186-297): the loss terms it logs and the gradient its optimizer.step() sees (128x, never unscaled)."""
    g = np.load(f'{G}/train_glue.npz')
    step = float(g['step'])
    for tag in ('geo_p2', 'geo_p8', 'app_p5'):
        o, d, gt_d, gt_c, occ, geo, app, t0, bg, noise = _train_glue_case(g, tag)
        kind = tag[:3]
        geo.requires_grad_(kind == 'geo'); app.requires_grad_(kind == 'app')
        out = O.occ_render(o, d, geo, app, occ, [-1, -1, -1, 1, 1, 1], training=True, t0=t0, bg_color=bg, dist_noise=noise,
                           near=0.0, far=1.5, step=step, geo_grad=(kind == 'geo'), app_grad=(kind == 'app'))
        if kind == 'geo':
            loss, dl, distl = O.geo_step_loss(out, gt_d, float(g[f'{tag}_progress']))
            assert abs(float(dl) - float(g[f'{tag}_depth_loss'])) < 1e-6 * max(1.0, abs(float(g[f'{tag}_depth_loss'])))
            assert abs(float(distl) - float(g[f'{tag}_dist_loss'])) < 1e-6 * max(1.0, abs(float(g[f'{tag}_dist_loss'])))
        else:
            loss, cl = O.app_step_loss(out, gt_c)
            assert abs(float(cl) - float(g[f'{tag}_color_loss'])) < 1e-6
        loss.backward()
        grad = (geo if kind == 'geo' else app).grad.numpy()
        assert grad.size == int(g[f'{tag}_grad_numel'])
        ref_norm = float(g[f'{tag}_grad_norm'])
        assert abs(float(np.sqrt((grad.astype(np.float64) ** 2).sum())) - ref_norm) < 1e-5 * ref_norm
        pg = np.random.RandomState(123).standard_normal((3, grad.size)).astype(np.float32)
        proj = pg.astype(np.float64) @ grad.astype(np.float64)
        assert np.abs(proj - g[f'{tag}_grad_proj']).max() < 1e-4 * ref_norm
        if f'{tag}_grad_idx' in g.files:
            ref = _dense_grad(g, tag)
            assert np.abs(grad - ref).max() <= 2e-5 * np.abs(ref).max(), tag


def _vis_infos(g):
    infos = []
    for i in range(2):
        infos.append({'pose': torch.from_numpy(g[f'pano{i}_pose']), 'distance_map': torch.from_numpy(g[f'pano{i}_distance']),
                      'mask': torch.from_numpy(g[f'pano{i}_mask'])})
    return infos


def test_visibility_and_geo_check_glue_match_reference():
    """perf_amd.visibility's torch formulation (the test reference of the HIP kernels) == the reference's own
    get_pano_visibility_mask (nerf.py:321-358) and SupInfoPool.geo_check (sup_info.py:261-302)."""
    from perf_amd import visibility as V
    g = np.load(f'{G}/visibility.npz')
    h, w = int(g['h']), int(g['w'])
    o, d = O.pano_rays(torch.from_numpy(g['probe_pose']), h, w)
    dist = torch.from_numpy(g['probe_distance'])
    infos = _vis_infos(g)
    vis = V.pano_visibility_mask(o, d, dist, infos, use_kernels=False)
    chk = V.geo_check(o, d, dist[..., None], infos, use_kernels=False)
    # the depth test compares two fp32 distances: a pixel may sit on the threshold; the morphology then spreads it
    assert float((vis.numpy() != g['visibility_mask']).mean()) < 0.01, float((vis.numpy() != g['visibility_mask']).mean())
    assert float((chk.numpy() != g['geo_check']).mean()) < 0.01
    assert 0.05 < float(g['visibility_mask'].mean()) < 0.98 and 0.02 < float(g['geo_check'].mean()) < 0.98        # non-trivial masks


def test_pano_sup_info_validity_rules_match_reference():
    """The validity rules SupInfoPool.register_sup_info applies (perf_amd/scene.py:_edge_free + mask / distance / normal tests) ==
    PanoSupInfo.__init__ (sup_info.py:27-97), and the supervision rays they select == update_sup_info (:99-120)."""
    from perf_amd.scene import _edge_free
    g = np.load(f'{G}/visibility.npz')
    h, w = int(g['h']), int(g['w'])
    for i in range(2):
        dist = torch.from_numpy(g[f'pano{i}_distance']); mask_in = torch.from_numpy(g[f'pano{i}_mask_in'])
        normal = torch.from_numpy(g[f'pano{i}_normal'])
        mask_raw = (mask_in > .5) & (dist > 1e-5)
        assert np.array_equal(mask_raw.numpy(), g[f'pano{i}_mask_raw'])
        valid = mask_raw & _edge_free(dist)
        _, local_d = O.pano_rays(torch.eye(4), h, w)
        valid = valid & (((-local_d) * normal).sum(-1, True).clip(0., 1.) > 0.15)
        assert np.array_equal(valid.numpy(), g[f'pano{i}_mask']), int((valid.numpy() != g[f'pano{i}_mask']).sum())
        assert 0.3 < float(valid.float().mean()) < 0.99
        # the rays of the valid pixels, in row-major order
        o, d = O.pano_rays(torch.from_numpy(g[f'pano{i}_pose']), h, w)
        idx = torch.where(valid[..., 0])
        assert np.abs(d[idx].numpy() - g[f'pano{i}_sup_dirs']).max() < 2e-6
        assert np.array_equal(o[idx].numpy(), g[f'pano{i}_sup_positions'])
        assert np.array_equal(dist[idx].numpy(), g[f'pano{i}_sup_distances'])


def test_dense_pose_sampler_in_a_worker_process_equals_the_in_line_call():
    """DenseTravelPoseSampler.start (forked worker, off the critical path of render_dense) == the in-line construction that
    is pinned on the reference's golden poses: same trajectory, same numpy RNG state afterwards; a second request with the
    same inputs is served from the memo."""
    import time
    from perf_amd import pose_sampler as PS
    g = np.load(f'{G}/poses.npz')
    s = PS.CirclePoseSampler(torch.from_numpy(g['distance_map']), [.2, .4, .6], [8, 8, 8])
    PS._DENSE_CACHE.clear()
    np.random.seed(0)
    fut = PS.DenseTravelPoseSampler.start(s, 180)
    dense = fut.result()
    after_async = np.random.rand()
    assert np.abs(dense.sample_poses.numpy() - g['dense']).max() < 1e-5
    PS._DENSE_CACHE.clear()
    np.random.seed(0)
    ref = PS.DenseTravelPoseSampler(s, 180)
    assert torch.equal(ref.sample_poses, dense.sample_poses) and np.random.rand() == after_async
    np.random.seed(0)
    t0 = time.perf_counter()
    again = PS.DenseTravelPoseSampler.start(s, 180)
    assert again.done() and torch.equal(again.result().sample_poses, ref.sample_poses) and time.perf_counter() - t0 < 0.05


def test_dense_pose_sampler_falls_back_when_its_worker_dies():
    """A worker that dies (or does not answer in time) must not hang render_dense: the future builds the trajectory in line
    from the RNG state saved at start() -- the same poses, the same numpy RNG state afterwards."""
    import warnings
    from perf_amd import pose_sampler as PS
    g = np.load(f'{G}/poses.npz')
    s = PS.CirclePoseSampler(torch.from_numpy(g['distance_map']), [.2, .4, .6], [8, 8, 8])
    PS._DENSE_CACHE.clear()
    np.random.seed(0)
    fut = PS.DenseTravelPoseSampler.start(s, 180)
    fut.proc.kill(); fut.proc.join()                      # the worker is gone before it could answer
    np.random.rand(3)                                      # ... and somebody drew from the global RNG in between
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        dense = fut.result(timeout=5.0)
    assert any('in line' in str(x.message) for x in w)
    after = np.random.rand()
    assert np.abs(dense.sample_poses.numpy() - g['dense']).max() < 1e-5
    PS._DENSE_CACHE.clear()
    np.random.seed(0)
    ref = PS.DenseTravelPoseSampler(s, 180)
    assert torch.equal(ref.sample_poses, dense.sample_poses) and np.random.rand() == after


def test_repeated_addition_lattice_closed_form():
    """The marching kernels evaluate the repeated-addition lattice t_{k+1} = fl(t_k + step) for an arbitrary k in closed form
    (perf_amd/csrc/march.hip:lattice_repeated: per binade two real additions, then a constant number of ulps per step).  The
    algorithm, restated here in numpy, equals the oracle's sequential accumulation for every k -- ties (dyadic steps) included."""
    F = np.float32

    def bits(x):
        return int(np.array([x], F).view(np.uint32)[0])

    def closed_form(t0, k, step):
        t, left = F(t0), k
        while left > 0:
            t1 = F(t + step); left -= 1
            if left == 0:
                return t1
            if (bits(t1) >> 23) != (bits(t) >> 23):
                t = t1; continue
            t2 = F(t1 + step); left -= 1
            if left == 0:
                return t2
            b1, b2 = bits(t1), bits(t2)
            if (b2 >> 23) != (b1 >> 23):
                t = t2; continue
            d = b2 - b1
            if d == 0:
                return t2
            j = min(((b2 & 0xff800000) + 0x00800000 - 1 - b2) // d, left)
            t = np.array([b2 + j * d], np.uint32).view(F)[0]; left -= j
        return t

    rng = np.random.RandomState(0)
    for step in (F(5e-4), F(0.99 / 128), F(1 / 3.), F(0.001953125), F(3e-4)):
        for t0 in list((rng.rand(4) * step).astype(F)) + [F(0), F(1e-9)]:
            K = 3100
            table = O.lattice_table_repeated(np.array([t0], F), K, step)[0]
            for k in list(range(0, 24)) + list(rng.randint(0, K + 1, 40)) + [K]:
                assert closed_form(t0, int(k), step) == table[k], (step, t0, k)


def test_repeated_addition_lattice_run_table():
    """perf_amd/csrc/march.hip:lattice_runs_build / LatticeRuns, restated in numpy: ONE walk per ray leaves a table of runs
    (first index, first bit pattern, ulps per step); t_k is read off the run that holds k.  Equal to the sequential
    accumulation for every k, including the reciprocal-estimate division (div_u24) the kernel uses."""
    F = np.float32

    def bits(x):
        return int(np.array([x], F).view(np.uint32)[0])

    def flt(b):
        return np.array([b & 0xffffffff], np.uint32).view(F)[0]

    def div_u24(n, d):
        q = int(np.uint32(F(F(n) * (F(1.0) / F(d)))))            # (uint32)((float)n * __frcp_rn((float)d)): truncation
        p = q * d
        if p > n:
            q -= 1
        elif n - p >= d:
            q += 1
        assert q == n // d, (n, d, q)
        return q

    def build(t0, step, k_need, max_runs=64):
        runs = [(0, bits(t0), 0)]
        tb, K = bits(t0), 0
        while K < k_need:
            if len(runs) + 2 > max_runs:
                return None
            b1 = bits(F(flt(tb) + step))
            if (b1 >> 23) != (tb >> 23):
                runs.append((K + 1, b1, 0)); tb = b1; K += 1; continue
            b2 = bits(F(flt(b1) + step))
            if (b2 >> 23) != (b1 >> 23):
                runs.append((K + 1, b1, 0)); runs.append((K + 2, b2, 0)); tb = b2; K += 2; continue
            d = b2 - b1
            runs.append((K + 1, b1, d))
            if d == 0:
                return runs
            j = div_u24((b2 & 0xff800000) + 0x00800000 - 1 - b2, d)
            tb = b2 + j * d; K += 2 + j
        return runs

    def evaluate(runs, k):
        lo, hi = 0, len(runs) - 1
        while lo < hi:
            mid = (lo + hi + 1) >> 1
            if runs[mid][0] <= k:
                lo = mid
            else:
                hi = mid - 1
        ks, bs, dd = runs[lo]
        return flt(bs + (k - ks) * dd)

    rng = np.random.RandomState(1)
    longest = 0
    for step in (F(5e-4), F(0.99 / 128), F(1 / 3.), F(0.001953125), F(3e-4), F(4e-3), F(1e-8)):
        for t0 in list((rng.rand(4) * step).astype(F)) + [F(0), F(1e-9), F(1e-2)]:
            K = 3100
            runs = build(t0, step, K + 64)
            assert runs is not None
            longest = max(longest, len(runs))
            table = O.lattice_table_repeated(np.array([t0], F), K + 64, step)[0]
            for k in list(range(0, 70)) + list(rng.randint(0, K + 65, 120)) + [K, K + 64]:
                assert evaluate(runs, int(k)) == table[k], (step, t0, k)
    assert longest <= 48, longest                                   # (the kernel's table holds 64 runs)


def test_synthetic_room_with_box_on_cpu():
    """perf_amd/synthetic.py (pure torch): without the box and from the centre room_with_box is room() up to the normalisation
    constant; the box hides part of the walls; from another position every ray still ends on a surface inside the room."""
    import math
    from perf_amd import synthetic

    def pano(pose, H, W):
        i = (torch.arange(H) + .5) / H; j = (torch.arange(W) + .5) / W
        y, x = torch.meshgrid(i, j, indexing='ij')
        beta = -(y - .5) * math.pi; alpha = -(x - .5) * 2 * math.pi
        d = torch.stack([torch.cos(alpha) * torch.cos(beta), torch.sin(alpha) * torch.cos(beta), torch.sin(beta)], -1)
        return pose[:3, 3].expand_as(d), d @ pose[:3, :3].T
    o, d = pano(torch.eye(4), 64, 128)
    d0, c0 = synthetic.room(d)
    d1, c1 = synthetic.room_with_box(o, d, box_h=(0., 0., 0.))
    ratio = d1 / d0
    assert float(ratio.max() - ratio.min()) < 1e-5 and 0.98 < float(ratio.mean()) < 1.0
    assert float((c0 - c1).abs().max()) < 1e-5
    d2, _ = synthetic.room_with_box(o, d)
    assert 0.01 < float((d2 < d1 - 1e-6).float().mean()) < 0.2
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.2, 0.1, 0.0])
    o3, dd3 = pano(pose, 64, 128)
    d3, c3 = synthetic.room_with_box(o3, dd3)
    p = (o3 + dd3 * d3) * synthetic.ROOM_SCALE
    half = torch.tensor([0.9, 0.7, 0.5])
    assert torch.isfinite(d3).all() and float(d3.min()) > 0
    assert bool(((p.abs() <= half + 1e-4).all(-1)).all())                       # every hit lies inside the room
    assert float(c3.min()) >= 0.0 and float(c3.max()) <= 1.0


def _upstream_files(tmp_path):
    """The committed vectors of the real packages when a maintainer has produced them (tools/pin_upstream.py), else vectors
    generated right here by the same script over the oracle-backed stand-in (tests/upstream_standin.py): the format and the
    checkers are exercised either way; only the former is a pin."""
    import os
    paths = {k: os.path.join(G, f'upstream_{k}.npz') for k in ('tcnn', 'nerfacc', 'distloss')}
    if all(os.path.exists(p) for p in paths.values()):
        return paths, True
    from tests import upstream_standin
    from tools import pin_upstream
    files = pin_upstream.main(modules=upstream_standin.modules(), out_dir=str(tmp_path), device='cpu', backend='oracle stand-in (NOT a pin)')
    return {k: files[f'upstream_{k}'] for k in paths}, False


def test_oracle_against_upstream_vectors(tmp_path):
    """tools/pin_upstream.py records what tinycudann / nerfacc / torch_efficient_distloss compute on seeded inputs; the oracle is
    evaluated on the same inputs.  With committed upstream files this test IS the pin of the third-party arithmetic (and of
    the marching lattice); without them it runs the script over the stand-in and must get its own numbers back -- which
    proves the file format, the seeded-input rules and the checkers a maintainer's run will meet."""
    from tests import upstream_check as C
    paths, pinned = _upstream_files(tmp_path)
    t = np.load(paths['tcnn'], allow_pickle=False); n = np.load(paths['nerfacc'], allow_pickle=False); dl = np.load(paths['distloss'], allow_pickle=False)
    assert int(t['format']) == 1 and C.is_upstream(t) == pinned
    rep = {'tcnn': C.oracle_vs_tcnn(t), 'nerfacc': C.oracle_vs_nerfacc(n), 'distloss': C.oracle_vs_distloss(dl)}
    assert rep['nerfacc']['lattice']['perf'][O.DEFAULT_LATTICE]
    if not pinned:
        # the two lattices really are different walks at PeRF's step: only one of them reproduces the recorded samples
        assert not rep['nerfacc']['lattice']['perf']['single']
    print('upstream vectors:', 'PINNED' if pinned else 'stand-in (parity of the third-party arithmetic stays unpinned)', rep)


def test_scene_shims_against_the_reference_tree():
    """Build-container only (the reference never travels): tools/check_reference_imports.py -- over the REAL reference tree,
    install_shims(scene=True) makes `modules.scene.nerf` / `modules.scene.nerf_renderer` / `modules.dataset.sup_info` resolve to
    the mirrors while the rest of the tree imports as it is; the mirrors bind the runner's constructor call with the reference's
    own configs/nerf.yaml and offer every method core_exp_runner.py calls on the scene and the pool."""
    import os
    import subprocess
    import sys
    import pytest
    if not os.path.isdir('/root/reference/modules'):
        pytest.skip('needs /root/reference (build container)')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_reference_imports.py')], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert r.returncode == 0 and 'install_shims(scene=True): OK' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_synthetic_scene_families_are_what_their_oracle_curves_were_made_on():
    """perf_amd/synthetic.py SCENES (pure torch): the doorway family shows depth discontinuities of ~3x along the door frame and rays
    that run on through the opening; the pillar family hides a sizeable part of the walls behind fourteen thin occluders; all
    distances are normalised into (0, 1/1.05], colours into [0, 1] -- and the committed oracle curves name their family."""
    import json
    import math
    from perf_amd import synthetic
    H, W = 128, 256
    i = (torch.arange(H) + .5) / H; j = (torch.arange(W) + .5) / W
    y, x = torch.meshgrid(i, j, indexing='ij')
    beta = -(y - .5) * math.pi; alpha = -(x - .5) * 2 * math.pi
    d = torch.stack([torch.cos(alpha) * torch.cos(beta), torch.sin(alpha) * torch.cos(beta), torch.sin(beta)], -1)
    room_d, _ = synthetic.room(d)
    for name, fn in synthetic.SCENES.items():
        dist, rgb = fn(d)
        assert dist.shape == (H, W, 1) and rgb.shape == (H, W, 3) and torch.isfinite(dist).all()
        assert float(dist.min()) > 0 and abs(float(dist.max()) - 1 / 1.05) < 1e-5 and 0.0 <= float(rgb.min()) and float(rgb.max()) <= 1.0
    door, _ = synthetic.doorway(d)
    ratio = door[:, 1:, 0] / door[:, :-1, 0]
    assert float(torch.maximum(ratio, 1 / ratio).max()) > 2.5                    # the door frame: neighbouring pixels 3x apart in depth
    row = door[H // 2 + 8, :, 0]                                                 # a row below the horizon, through the opening
    assert float(row.max()) > 2.0 * float(row.min())
    pil, _ = synthetic.pillars(d)
    hidden = float((pil[..., 0] * float(pil.max() / room_d.max()) < room_d[..., 0] * 0.9).float().mean())
    assert 0.1 < hidden < 0.6, hidden
    jumps = ((pil[H // 2, 1:, 0] / pil[H // 2, :-1, 0] - 1).abs() > 0.2).sum()
    assert int(jumps) >= 16                                                      # the horizon row crosses many pillar edges
    for name in ('doorway', 'pillars'):
        cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', f'psnr_curve_{name}.json')))
        assert cfg['config']['scene'] == name and len(cfg['seeds']) == 3
