"""GPU parity tests: every C-ABI kernel against the CPU oracle on the same seeded inputs.

Integer / index bookkeeping is compared bit-exactly; floating point within the tolerance
written next to each assertion.  All calls go through libperf_hip.so (perf_amd.ops)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import perf_oracle as O  # noqa: E402


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    from perf_amd import ops as _ops
    return _ops


def _grid_cfg(**kw):
    from perf_amd.grid import GridConfig
    return GridConfig(**kw)


def _lv_of(cfg):
    return O.grid_levels(cfg.n_levels, 2, cfg.log2_hashmap_size, cfg.base_resolution, cfg.per_level_scale, layout=cfg.layout, sb_shift=cfg.sb_shift,
                         local_min_res=cfg.local_min_res)


DT = {'bf16': (torch.bfloat16, 2.0 ** -8), 'fp16': (torch.float16, 2.0 ** -11)}


def test_level_table_matches_oracle():
    cfg = _grid_cfg()
    lv = O.grid_levels()
    assert cfg.total == lv.total == 3320608
    assert np.array_equal(cfg.scale, lv.scale) and np.array_equal(cfg.res, lv.res)
    assert np.array_equal(cfg.size, lv.size) and np.array_equal(cfg.offset, lv.offset)


def test_pano_raygen(ops, golden_dir):
    g = np.load(f'{golden_dir}/rays.npz')
    for name in ('eye', 'rt'):
        pose = torch.from_numpy(g[f'pose_{name}'])
        o, d = ops.pano_raygen(pose, 32, 64)
        # fp32 tolerance: GPU sinf/cosf vs the reference's CPU libm, unit vectors -> 2e-6 absolute
        assert np.abs(d.cpu().numpy() - g[f'pano_{name}_32x64_d']).max() < 2e-6
        assert np.array_equal(o.cpu().numpy(), g[f'pano_{name}_32x64_o'])
        for (h, w) in ((256, 512), (1024, 2048)):
            o, d = ops.pano_raygen(pose, h, w)
            ij = g[f'pano_{name}_{h}x{w}_ij']
            got = d.cpu().numpy()[ij[:, 0], ij[:, 1]]
            assert np.abs(got - g[f'pano_{name}_{h}x{w}_d']).max() < 2e-6
        # row-sharded generation is identical to the full one (multi-GPU eval partitioning)
        o2, d2 = ops.pano_raygen(pose, 256, 512, row0=64, nrows=32)
        o1, d1 = ops.pano_raygen(pose, 256, 512)
        assert torch.equal(d1[64:96], d2) and torch.equal(o1[64:96], o2)
    # full-size size-independent property: unit norm and direction -> pixel round trip
    o, d = ops.pano_raygen(torch.eye(4), 1024, 2048)
    assert (d.norm(dim=-1) - 1).abs().max() < 1e-6


def test_points_from_rays_bit_exact(ops):
    g = torch.Generator().manual_seed(0)
    R, S = 257, 5000
    o = torch.rand(R, 3, generator=g) * 0.4 - 0.2
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    ri = torch.sort(torch.randint(0, R, (S,), generator=g)).values
    ts = torch.rand(S, generator=g) * 1.5
    te = ts + 5e-4
    aabb = torch.tensor([-1., -1, -1, 1, 1, 1])
    x01, sel = ops.points_from_rays(o.cuda(), d.cuda(), ri.cuda(), ts.cuda(), te.cuda(), aabb)
    pos = o[ri] + d[ri] * (ts + te)[:, None] / 2.0
    ref = (pos - aabb[:3]) / (aabb[3:] - aabb[:3])
    assert torch.equal(x01.cpu(), ref)
    assert torch.equal(sel.cpu().bool(), ((ref > 0) & (ref < 1)).all(-1))


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('interp', ['Linear', 'Smoothstep'])
def test_hashgrid_fwd(ops, dt, interp):
    tdt, ulp = DT[dt]
    cfg = _grid_cfg(interpolation=interp)
    lv = _lv_of(cfg)
    g = torch.Generator().manual_seed(1)
    table = (torch.rand(cfg.total, 2, generator=g) * 2 - 1)
    n = 3001                                   # ragged: not a multiple of the block or tile size
    x = torch.rand(n, 3, generator=g)
    x[:7] = torch.tensor([[0., 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1e-7, 1 - 1e-7, 0.25], [0.999999, 0, 1],
                          [1.0, 0.0, 0.3], [0.3, 1.0, 0.0]])
    t16 = table.to(tdt)
    feat = ops.hashgrid_fwd(cfg, x.cuda(), t16.reshape(-1).cuda())
    got = feat.float().cpu().permute(1, 0, 2).reshape(n, -1)
    ref = O.hashgrid_encode(x, table, lv, interpolation=interp, quant=dt)
    # the kernel rounds the fp32 interpolation to 16 bits: 1 ulp of the storage type (+ fp32 noise)
    err = (got - ref).abs()
    assert (err <= ulp * ref.abs() * 1.01 + 1e-6).all(), err.max()
    # empty input is legal
    assert ops.hashgrid_fwd(cfg, x[:0].cuda(), t16.reshape(-1).cuda()).shape == (cfg.n_levels, 0, 2)


def test_hashgrid_fwd_f32_and_small_grid(ops):
    cfg = _grid_cfg(n_levels=5, log2_hashmap_size=17, base_resolution=16,
                    per_level_scale=float(np.exp((np.log(128) - np.log(16)) / 4)))
    lv = _lv_of(cfg)
    g = torch.Generator().manual_seed(2)
    table = torch.rand(cfg.total, 2, generator=g) * 2 - 1
    x = torch.rand(1000, 3, generator=g)
    feat = ops.hashgrid_fwd_f32(cfg, x.cuda(), table.reshape(-1).cuda())
    got = feat.cpu().permute(1, 0, 2).reshape(1000, -1)
    ref = O.hashgrid_encode(x, table, lv)
    assert (got - ref).abs().max() < 2e-6       # fp32 accumulation-order noise only


def test_hashgrid_bwd_indices_and_weights(ops):
    """The scatter touches exactly the oracle's corner indices (bit-exact bookkeeping) with the
    oracle's trilinear weights (fp32 atomics: summation-order tolerance)."""
    cfg = _grid_cfg()
    lv = _lv_of(cfg)
    g = torch.Generator().manual_seed(3)
    n = 777
    x = torch.rand(n, 3, generator=g)
    dfeat = torch.randn(cfg.n_levels, n, 2, generator=g)
    grad = ops.hashgrid_bwd(cfg, x.cuda(), dfeat.cuda()).cpu().numpy().reshape(-1, 2)
    ref = np.zeros((cfg.total, 2), np.float64)
    xn = x.numpy()
    for l in range(cfg.n_levels):
        idx, f = O.grid_corner_indices(xn, lv, l)
        for c in range(8):
            w = np.ones(n, np.float32)
            for a in range(3):
                w = w * (f[:, a] if (c >> a) & 1 else (np.float32(1) - f[:, a]))
            np.add.at(ref, idx[:, c].astype(np.int64) + int(lv.offset[l]), w[:, None].astype(np.float64) * dfeat[l].numpy())
    assert np.array_equal(grad != 0, ref != 0)            # same set of touched entries
    assert np.abs(grad - ref).max() < 1e-5


def test_hashgrid_bwd_input(ops):
    for interp in ('Linear', 'Smoothstep'):
        cfg = _grid_cfg(n_levels=8, log2_hashmap_size=15, interpolation=interp)
        lv = _lv_of(cfg)
        g = torch.Generator().manual_seed(4)
        table = torch.rand(cfg.total, 2, generator=g) * 2 - 1
        x = (torch.rand(500, 3, generator=g) * 0.98 + 0.01).requires_grad_(True)
        dfeat = torch.randn(cfg.n_levels, 500, 2, generator=g)
        ref_feat = O.hashgrid_encode(x, table, lv, interpolation=interp)
        (ref_feat * dfeat.permute(1, 0, 2).reshape(500, -1)).sum().backward()
        dx = ops.hashgrid_bwd_input(cfg, x.detach().cuda(), dfeat.cuda(), table.reshape(-1).cuda()).cpu()
        scale = x.grad.abs().max()
        assert (dx - x.grad).abs().max() < 1e-4 * scale


def _mlp_case(ops, dt, nh, n_out, act, n, n_levels=16, seed=0):
    from perf_amd.grid import MlpConfig
    tdt, ulp = DT[dt]
    cfgm = MlpConfig(n_levels=n_levels, n_hidden_layers=nh, n_output_dims=n_out, output_activation=act,
                     exp_shift=1.0 if act == 'Exponential' else 0.0)
    g = torch.Generator().manual_seed(seed)
    w = torch.cat([(torch.rand(o * i, generator=g) * 2 - 1) * math.sqrt(6.0 / (i + o)) * 1.5 for o, i in cfgm.shapes])
    feat = (torch.rand(n_levels, n, 2, generator=g) * 2 - 1)
    sel = (torch.rand(n, generator=g) > 0.2).to(torch.uint8)
    return cfgm, w, feat, sel, tdt, ulp


def _mlp_oracle(cfgm, w, feat, sel, dt):
    n = feat.shape[1]
    x = feat.permute(1, 0, 2).reshape(n, -1)
    if cfgm.n_in_pad > x.shape[1]:
        x = torch.cat([x, torch.zeros(n, cfgm.n_in_pad - x.shape[1])], 1)
    y = O.mlp_forward(x, w, cfgm.n_in_pad, cfgm.n_hidden_layers, cfgm.n_output_dims, 'None', quant=dt)
    if cfgm.output_activation == 'Sigmoid':
        y = torch.sigmoid(y)
    elif cfgm.output_activation == 'Exponential':
        y = O.trunc_exp(y - cfgm.exp_shift)
    return y * sel[:, None].float()


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('nh,n_out,act', [(1, 1, 'None'), (1, 1, 'Exponential'), (2, 3, 'Sigmoid'), (2, 16, 'None')])
def test_mlp_fwd(ops, dt, nh, n_out, act):
    cfgm, w, feat, sel, tdt, ulp = _mlp_case(ops, dt, nh, n_out, act, n=1000 + 13)
    w16 = w.to(tdt)
    f16 = feat.to(tdt)
    out = ops.mlp_fwd(cfgm, w16.cuda(), f16.cuda(), sel.cuda()).cpu()
    ref = _mlp_oracle(cfgm, w16.float(), f16.float(), sel, dt)
    # operands are identical 16-bit values on both sides; differences come from fp32 accumulation order
    # and from hidden activations that round differently at 16-bit ties: a few 16-bit ulps of the largest activation
    tol = 8 * ulp * max(1.0, float(ref.abs().max()))
    assert (out - ref).abs().max() < tol, (out - ref).abs().max()


def test_mlp_fwd_small_input(ops):
    cfgm, w, feat, sel, tdt, ulp = _mlp_case(ops, 'fp16', 1, 1, 'Exponential', n=300, n_levels=5, seed=5)
    out = ops.mlp_fwd(cfgm, w.to(tdt).cuda(), feat.to(tdt).cuda(), None).cpu()
    ref = _mlp_oracle(cfgm, w.to(tdt).float(), feat.to(tdt).float(), torch.ones(300, dtype=torch.uint8), 'fp16')
    assert (out - ref).abs().max() < 8 * ulp * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('nh,n_out,act', [(1, 1, 'Exponential'), (2, 3, 'Sigmoid'), (1, 1, 'None')])
def test_mlp_bwd(ops, dt, nh, n_out, act):
    cfgm, w, feat, sel, tdt, ulp = _mlp_case(ops, dt, nh, n_out, act, n=2000 + 7, seed=7)
    w16 = w.to(tdt); f16 = feat.to(tdt)
    g = torch.Generator().manual_seed(8)
    dout = torch.randn(feat.shape[1], n_out, generator=g)
    dfeat, dw = ops.mlp_bwd(cfgm, w16.cuda(), f16.cuda(), dout.cuda(), sel.cuda())
    wr = w16.float().requires_grad_(True)
    fr = f16.float().requires_grad_(True)
    y = _mlp_oracle(cfgm, wr, fr, sel, dt)
    (y * dout).sum().backward()
    # the kernel rounds dY and every dH to 16 bits before the MFMA chain (tcnn does the same in fp16):
    # a few 16-bit ulps relative to the largest gradient of the tensor
    def close(a, b, k):
        return (a - b).abs().max() <= k * ulp * float(b.abs().max()) + 1e-6
    assert close(dfeat.cpu(), fr.grad, 16), ((dfeat.cpu() - fr.grad).abs().max(), fr.grad.abs().max())
    assert close(dw.cpu(), wr.grad, 16), ((dw.cpu() - wr.grad).abs().max(), wr.grad.abs().max())


@pytest.mark.parametrize('nh,n_out,act,n_levels,n,with_sel', [
    (1, 16, 'None', 16, 300_000 + 17, True),     # the geometry step's network; every wave goes round its two input sets > 4 times
    (2, 3, 'Sigmoid', 16, 150_000 + 1, False),   # the colour network, no selector (the request reads a dummy byte)
    (1, 16, 'None', 8, 4000 + 3, True),          # one k-step: scalar-base loads, per-level stores
    (1, 1, 'Exponential', 16, 40, False),        # two tiles: most waves have none
    (2, 16, 'None', 16, 1, True)])               # one sample
def test_mlp_tiles_ahead(ops, nh, n_out, act, n_levels, n, with_sel):
    """mlp_fwd / mlp_bwd request a tile's inputs one tile ahead on alternating register sets, re-reading the last tile (and its last
    sample) where there is nothing to request: same bars as test_mlp_fwd / test_mlp_bwd at the sizes that exercise that."""
    dt = 'bf16'
    cfgm, w, feat, sel, tdt, ulp = _mlp_case(ops, dt, nh, n_out, act, n=n, n_levels=n_levels, seed=21)
    if not with_sel:
        sel = torch.ones(n, dtype=torch.uint8)
    sel_dev = sel.cuda() if with_sel else None
    w16 = w.to(tdt); f16 = feat.to(tdt)
    out = ops.mlp_fwd(cfgm, w16.cuda(), f16.cuda(), sel_dev).cpu()
    ref = _mlp_oracle(cfgm, w16.float(), f16.float(), sel, dt)
    assert (out - ref).abs().max() < 8 * ulp * max(1.0, float(ref.abs().max()))
    g = torch.Generator().manual_seed(22)
    dout = torch.randn(n, n_out, generator=g)
    dfeat, dw, amax = ops.mlp_bwd(cfgm, w16.cuda(), f16.cuda(), dout.cuda(), sel_dev, want_absmax=True)
    wr = w16.float().requires_grad_(True)
    fr = f16.float().requires_grad_(True)
    (_mlp_oracle(cfgm, wr, fr, sel, dt) * dout).sum().backward()

    def close(a, b, k):
        return (a - b).abs().max() <= k * ulp * float(b.abs().max()) + 1e-6
    assert close(dfeat.cpu(), fr.grad, 16), ((dfeat.cpu() - fr.grad).abs().max(), fr.grad.abs().max())
    # dW sums n products rounded to 16 bits each: the bound grows like sqrt(n) rounding errors of the largest product
    k = 16 * max(1.0, math.sqrt(n / 2000.0))
    assert close(dw.cpu(), wr.grad, k), ((dw.cpu() - wr.grad).abs().max(), wr.grad.abs().max())
    assert float(amax.max()) == float(dfeat.abs().max())


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('nh,n_out,act,n_levels', [(1, 1, 'Exponential', 20), (2, 3, 'Sigmoid', 20), (1, 1, 'None', 24), (2, 16, 'None', 17)])
def test_mlp_more_than_16_levels(ops, dt, nh, n_out, act, n_levels):
    """BASELINE config 5 trains L = 20 grids: 40 (up to 48) input features = three k-steps forward and two 32-row blocks of
    dX / dW1 backward.  Same bars as test_mlp_fwd / test_mlp_bwd, plus the per-half level maxima the grid backward scales by."""
    cfgm, w, feat, sel, tdt, ulp = _mlp_case(ops, dt, nh, n_out, act, n=3000 + 5, n_levels=n_levels, seed=11)
    w16 = w.to(tdt); f16 = feat.to(tdt)
    out = ops.mlp_fwd(cfgm, w16.cuda(), f16.cuda(), sel.cuda()).cpu()
    ref = _mlp_oracle(cfgm, w16.float(), f16.float(), sel, dt)
    assert (out - ref).abs().max() < 8 * ulp * max(1.0, float(ref.abs().max()))
    g = torch.Generator().manual_seed(12)
    dout = torch.randn(feat.shape[1], n_out, generator=g)
    dfeat, dw, amax = ops.mlp_bwd(cfgm, w16.cuda(), f16.cuda(), dout.cuda(), sel.cuda(), want_absmax=True)
    wr = w16.float().requires_grad_(True)
    fr = f16.float().requires_grad_(True)
    (_mlp_oracle(cfgm, wr, fr, sel, dt) * dout).sum().backward()

    def close(a, b, k):
        return (a - b).abs().max() <= k * ulp * float(b.abs().max()) + 1e-6
    assert dfeat.shape == (n_levels, feat.shape[1], 2) and dw.shape == wr.grad.shape
    assert close(dfeat.cpu(), fr.grad, 16), ((dfeat.cpu() - fr.grad).abs().max(), fr.grad.abs().max())
    assert close(dw.cpu(), wr.grad, 16), ((dw.cpu() - wr.grad).abs().max(), wr.grad.abs().max())
    # level l's bound is the maximum over the levels its half-wave owns ((l >> 1) & 1): never below the level's own maximum
    own = dfeat.abs().amax(dim=(1, 2))
    assert bool((amax[:n_levels] >= own).all()) and float(amax[n_levels:].abs().sum()) == 0.0
    assert float(amax.max()) == float(own.max())


def test_cast_and_adam(ops):
    g = torch.Generator().manual_seed(9)
    n = 100003
    p = torch.randn(n, generator=g)
    assert torch.equal(ops.cast_params(p.cuda(), 'bf16').cpu(), p.to(torch.bfloat16))
    assert torch.equal(ops.cast_params(p.cuda(), 'fp16').cpu(), p.to(torch.float16))
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2)
    pd = p.cuda().clone(); m = torch.zeros_like(pd); v = torch.zeros_like(pd)
    w16 = torch.empty(n, dtype=torch.bfloat16, device='cuda')
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * 128
        ref.grad = gr.clone()
        opt.step()
        gd = gr.cuda()
        ops.adam_step(pd, m, v, gd, step, 1e-2, w16=w16)
        assert float(gd.abs().max()) == 0.0                       # gradient cleared in the same pass
    assert (pd.cpu() - ref.detach()).abs().max() < 2e-6            # fp32, 1-2 ulp of |p|~1
    assert torch.equal(w16.cpu(), pd.cpu().to(torch.bfloat16))


def _room(R_h=24, R_w=48, res=64, seed=0):
    o, d = O.pano_rays(torch.eye(4), R_h, R_w)
    o = o.reshape(-1, 3).contiguous(); d = d.reshape(-1, 3).contiguous()
    dist, rgb = O.synthetic_room(d)
    occ = O.gen_occ_grid(o, d, dist, res)
    return o, d, dist, rgb, occ


def test_occ_splat_and_pack(ops, golden_dir):
    o, d, dist, rgb, occ = _room(16, 32, 64)
    got = ops.occ_splat(o.cuda(), d.cuda(), dist.cuda(), 64)
    assert torch.equal(got.cpu(), occ)
    gold = np.load(f'{golden_dir}/sup.npz')['occ_16x32_r64_idx']
    assert np.array_equal(torch.where(got.cpu() > 0)[0].numpy(), gold)
    bits = ops.occ_pack_bits(got).cpu().numpy().view(np.uint32)
    ref_bits = np.packbits(occ.numpy().astype(bool), bitorder='little').view(np.uint32)
    assert np.array_equal(bits, ref_bits)


@pytest.mark.parametrize('stratified', [False, True])
def test_occ_march_bit_exact(ops, stratified):
    o, d, dist, rgb, occ = _room(24, 48, 64)
    # move the camera off-centre and add rays that miss / graze the box
    o = o + torch.tensor([0.2, -0.1, 0.05])
    o[:5] = torch.tensor([3.0, 0.0, 0.0]); d[5] = torch.tensor([1.0, 0.0, 0.0]); d[6] = torch.tensor([0.0, 0.0, -1.0])
    R = o.shape[0]
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    step, far, near = 2e-3, 1.5, 0.0
    max_steps = int(math.ceil((far - near) / step)) + 1
    g = torch.Generator().manual_seed(5)
    t0 = torch.full((R,), near) + (torch.rand(R, generator=g) * step if stratified else 0.0)
    binaries = occ.reshape(64, 64, 64).bool()
    ri, ts, te, packed = O.occ_march(o.numpy(), d.numpy(), binaries.numpy(), aabb, near, far, step, t0.numpy(), max_steps)
    bits = ops.occ_pack_bits(occ.cuda())
    assert ri.size > 1000
    coarse = ops.occ_build_coarse(bits, 64)
    for cz in (None, coarse):            # exhaustive lattice test, and with the conservative empty-space skip
        gri, gts, gte, gpacked = ops.occ_march(o.cuda(), d.cuda(), t0.cuda(), bits, 64, aabb, far, step, max_steps, occ_coarse=cz)
        assert np.array_equal(gri.cpu().numpy(), ri)
        assert np.array_equal(gts.cpu().numpy(), ts) and np.array_equal(gte.cpu().numpy(), te)   # bit-exact floats
        assert np.array_equal(gpacked.cpu().numpy(), packed)


@pytest.mark.parametrize('K', [1, 4, 16])
def test_occ_march_count_head_equals_the_three_pass_head(ops, K):
    """perf_occ_march_count_head: the counting pass writes the first K samples of every ray itself (rows r*K.., padding rows
    with selector 0) -- same masks / counts as perf_occ_march_count and, per ray, bit for bit the samples and positions that
    count clamp + scan + perf_occ_march_write_points produce for ranks [0, K); rays that miss the box included."""
    o, d, dist, rgb, occ = _room(24, 48, 64)
    o = o + torch.tensor([0.2, -0.1, 0.05])
    o[:5] = torch.tensor([3.0, 0.0, 0.0]); d[5] = torch.tensor([1.0, 0.0, 0.0]); d[6] = torch.tensor([0.0, 0.0, -1.0])
    R = o.shape[0]
    aabb = [-1., -1, -1, 1, 1, 1]
    step, far = 2e-3, 1.5
    max_steps = int(math.ceil(far / step)) + 1
    g = torch.Generator().manual_seed(5)
    t0 = (torch.rand(R, generator=g) * step).cuda()
    bits = ops.occ_pack_bits(occ.cuda())
    coarse = ops.occ_build_coarse(bits, 64)
    oc, dc = o.cuda(), d.cuda()
    masks, counts = ops.occ_march_count(oc, dc, t0, bits, 64, aabb, far, step, max_steps, coarse)
    ch = ops.head_tail_counts(counts, K)
    oh, total = ops.exclusive_scan_i32(ch)
    ri, ts, te, pk, x01, sel = ops.occ_march_write(t0, masks, ch, oh, R * K, step, max_steps, oc, dc, aabb)
    m2, c2, (ri2, ts2, te2, pk2, x2, s2) = ops.occ_march_count_head(oc, dc, t0, bits, 64, aabb, far, step, max_steps, coarse, K, aabb)
    assert torch.equal(c2, counts) and int((counts == 0).sum()) >= 5 and int((counts >= K).sum()) > 100
    have = torch.clamp(counts, max=K)
    assert torch.equal(pk2[:, 1], have) and torch.equal(pk2[:, 0], torch.arange(R, device='cuda', dtype=torch.int32) * K)
    # strided rows -> packed order
    row = torch.arange(R * K, device='cuda').view(R, K)
    live = (torch.arange(K, device='cuda')[None, :] < have[:, None])
    idx = row[live]
    n = int(total.item())
    assert idx.numel() == n
    for a, b in ((ri2, ri), (ts2, ts), (te2, te), (x2, x01), (s2, sel)):
        assert torch.equal(a[idx], b[:n])
    pad = row[~live]
    assert int(s2[pad].sum()) == 0 and bool((x2[pad] == 0.5).all())
    # the keep masks of the live chunks are the same records
    r0 = int(torch.nonzero(counts > 0)[0])
    mw = masks.numel() // R
    assert torch.equal(m2.view(R, mw)[r0, :1], masks.view(R, mw)[r0, :1])


def test_scan_and_empty(ops):
    g = torch.Generator().manual_seed(6)
    # (beyond 2,097,152 elements the block sums get a scan launch of their own: the last two sizes)
    for n in (1, 5, 1024, 1025, 8192, 65536, 65537, 100000, 2097152, 2097153 + 1024 * 1500 + 7, 8388608):
        c = torch.randint(0, 100 if n > 2 ** 20 else 300, (n,), generator=g, dtype=torch.int32)
        out, total = ops.exclusive_scan_i32(c.cuda())
        ref = torch.cumsum(c.long(), 0) - c.long()
        assert torch.equal(out.cpu().long(), ref) and int(total.item()) == int(c.sum())
    # a batch with no occupied cell returns zero samples (nerf_renderer.py:156-162 guard)
    occ = torch.zeros(32 ** 3, dtype=torch.uint8)
    o = torch.zeros(10, 3); d = torch.nn.functional.normalize(torch.randn(10, 3, generator=g), dim=-1)
    bits = ops.occ_pack_bits(occ.cuda())
    ri, ts, te, packed = ops.occ_march(o.cuda(), d.cuda(), torch.zeros(10).cuda(), bits, 32, [-1, -1, -1, 1, 1, 1], 1.5, 1e-2, 151)
    assert ri.numel() == 0 and int(packed[:, 1].sum()) == 0


def _packed_case(seed=0, R=300, maxc=200):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, maxc, (R,), generator=g)
    counts[3] = 0; counts[10] = 64; counts[11] = 65; counts[12] = 1
    starts = torch.cumsum(counts, 0) - counts
    packed = torch.stack([starts, counts], -1).to(torch.int32)
    S = int(counts.sum())
    ri = torch.repeat_interleave(torch.arange(R), counts)
    ts = torch.rand(S, generator=g) * 1.5
    te = ts + 5e-3 * (1 + torch.rand(S, generator=g))
    sig = torch.exp(torch.randn(S, generator=g) * 2 + 2)
    rgb = torch.rand(S, 3, generator=g)
    return packed, ri, ts, te, sig, rgb


def test_visibility_and_compaction_bit_exact(ops):
    packed, ri, ts, te, sig, rgb = _packed_case(1)
    keep, ex = O.visibility_keep_mask(sig.numpy(), ts.numpy(), te.numpy(), packed.numpy(), 1e-4)
    nc, gex = ops.visibility_count(sig.cuda(), ts.cuda(), te.cuda(), packed.cuda(), 1e-4, want_exsum=True)
    assert np.array_equal(gex.cpu().numpy(), ex)            # canonical scan order: bit-exact fp32
    ref_counts = np.bincount(ri.numpy()[keep], minlength=packed.shape[0])
    assert np.array_equal(nc.cpu().numpy(), ref_counts)
    assert 0 < keep.sum() < keep.size
    gri, gts, gte, gsig, gpacked = ops.compact_prefix(packed.cuda(), nc, ts.cuda(), te.cuda(), sig.cuda())
    assert np.array_equal(gri.cpu().numpy(), ri.numpy()[keep])
    assert np.array_equal(gts.cpu().numpy(), ts.numpy()[keep]) and np.array_equal(gsig.cpu().numpy(), sig.numpy()[keep])
    assert np.array_equal(gpacked.cpu().numpy(), O.packed_info_from_ray_indices(ri.numpy()[keep], packed.shape[0]))


def test_team_kernels_give_the_same_bits_in_every_launch_shape(ops):
    """The ray-team kernels (visibility, compaction, compositing) pick their launch shape from the number of rays: one wave per ray,
    four rays per wave, and -- from 524,288 rays on: a 512 x 1024 eval frame as one batch -- sixteen rays per wave (4-lane teams when all
    sixteen hold <= 4 samples, else their four groups of four one after the other).  A frame-sized launch whose stretches of rays are short
    (0-4 samples), medium (<= 16), long (up to 150) and mixed gives, in two halves (four rays per wave) and ray by ray in small pieces (one
    wave per ray), the same bits."""
    g = torch.Generator().manual_seed(77)
    R = 530000
    counts = torch.randint(0, 5, (R,), generator=g)                       # short rays ...
    seg = torch.arange(R) // 4096
    counts = torch.where(seg % 5 == 1, torch.randint(0, 17, (R,), generator=g), counts)            # ... stretches of medium ones
    counts = torch.where(seg % 10 == 2, torch.randint(0, 150, (R,), generator=g), counts)          # ... of long ones
    lone = torch.randint(0, R, (300,), generator=g)
    counts[lone] = torch.randint(5, 90, (300,), generator=g)              # ... and single heavier rays among the short ones
    starts = torch.cumsum(counts, 0) - counts
    packed = torch.stack([starts, counts], -1).to(torch.int32).cuda()
    S = int(counts.sum())
    ts = (torch.rand(S, generator=g) * 1.5).cuda(); te = ts + 5e-3
    sig = torch.exp(torch.randn(S, generator=g) * 2 + 2).cuda()
    rgb = torch.rand(S, 3, generator=g).cuda()
    nc_a, ex_a = ops.visibility_count(sig, ts, te, packed, 1e-4, want_exsum=True)
    w_a, T_a, al_a, op_a, d_a, col_a = ops.composite_fwd(sig, rgb, ts, te, packed)
    cp_a = ops.compact_prefix(packed, nc_a, ts, te, sig)
    assert 0 < int(nc_a.sum()) < S
    pieces = [(0, R // 2), (R // 2, R)] + [(lo, lo + 3000) for lo in (0, 4096 + 17, 2 * 4096 + 5, 5 * 4096 - 1500, R - 3000)]
    kept_before = torch.cumsum(nc_a.long(), 0) - nc_a.long()
    for lo, hi in pieces:
        s0, s1 = int(starts[lo]), int(starts[hi - 1] + counts[hi - 1])
        pk = packed[lo:hi].clone(); pk[:, 0] -= s0
        nc, ex = ops.visibility_count(sig[s0:s1].contiguous(), ts[s0:s1].contiguous(), te[s0:s1].contiguous(), pk, 1e-4, want_exsum=True)
        assert torch.equal(nc, nc_a[lo:hi]) and torch.equal(ex, ex_a[s0:s1]), (lo, hi)
        w, T, al, op, dd, col = ops.composite_fwd(sig[s0:s1].contiguous(), rgb[s0:s1].contiguous(), ts[s0:s1].contiguous(), te[s0:s1].contiguous(), pk)
        for a, b, name in ((w, w_a[s0:s1], 'w'), (T, T_a[s0:s1], 'T'), (al, al_a[s0:s1], 'alpha'), (op, op_a[lo:hi], 'op'), (dd, d_a[lo:hi], 'dist'), (col, col_a[lo:hi], 'col')):
            assert torch.equal(a, b), (name, lo, hi)
        cp = ops.compact_prefix(pk, nc, ts[s0:s1].contiguous(), te[s0:s1].contiguous(), sig[s0:s1].contiguous())
        k0 = int(kept_before[lo]); k1 = k0 + int(nc.sum())
        assert torch.equal(cp[0] + lo, cp_a[0][k0:k1]) and torch.equal(cp[1], cp_a[1][k0:k1]) and torch.equal(cp[3], cp_a[3][k0:k1])
        assert torch.equal(cp[4][:, 1], cp_a[4][lo:hi, 1])


def test_compaction_hands_out_rows_instead_of_a_feature_copy(ops):
    """perf_compact_prefix(src_index_out) + perf_mlp_bwd(feat_index): the kept samples' features are read where the sampler's
    density pass wrote them.  The rows are the kept positions of the input order; the MLP backward through them is bit-identical
    to the one on the compacted copy -- with a device-side live count below the capacity, too."""
    packed, ri, ts, te, sig, rgb = _packed_case(5, R=700, maxc=150)
    S = ts.shape[0]
    g = torch.Generator().manual_seed(6)
    feat = ((torch.rand(16, S, 2, generator=g) * 2 - 1)).to(torch.bfloat16).cuda()
    x01 = torch.rand(S, 3, generator=g).cuda(); sel = (torch.rand(S, generator=g) > 0.1).to(torch.uint8).cuda()
    keep, _ = O.visibility_keep_mask(sig.numpy(), ts.numpy(), te.numpy(), packed.numpy(), 1e-4)
    nc = ops.visibility_count(sig.cuda(), ts.cuda(), te.cuda(), packed.cuda(), 1e-4)
    cap = S + 77
    args = (packed.cuda(), nc, ts.cuda(), te.cuda(), sig.cuda())
    copy = ops.compact_prefix(*args, capacity=cap, x01=x01, sel=sel, feat=feat, index_features=False)
    rows = ops.compact_prefix(*args, capacity=cap, x01=x01, sel=sel, feat=feat, index_features=True)
    n_kept = int(copy[5].item())
    assert n_kept == int(keep.sum()) and int(rows[5].item()) == n_kept
    fi = rows[8]
    assert isinstance(fi, ops.IndexedFeat) and fi.feat is feat and fi.index.dtype == torch.int32 and fi.index.shape == (cap,)
    assert np.array_equal(fi.index[:n_kept].cpu().numpy(), np.nonzero(keep)[0])
    assert torch.equal(fi.materialize()[:, :n_kept], copy[8][:, :n_kept])
    for a, b in zip(copy[:8], rows[:8]):                       # everything else is compacted as before
        assert torch.equal(a[:n_kept] if a.shape[0] == cap else a, b[:n_kept] if b.shape[0] == cap else b)
    from perf_amd.grid import MlpConfig
    for nh, n_out, act in ((1, 1, 'Exponential'), (1, 16, 'None'), (2, 3, 'Sigmoid')):
        cfgm = MlpConfig(n_levels=16, n_hidden_layers=nh, n_output_dims=n_out, output_activation=act, exp_shift=1.0 if act == 'Exponential' else 0.0)
        w = torch.cat([(torch.rand(o * i, generator=g) * 2 - 1) * math.sqrt(6.0 / (i + o)) for o, i in cfgm.shapes]).to(torch.bfloat16).cuda()
        dout = torch.randn(cap, n_out, generator=g).cuda()
        a = ops.mlp_bwd(cfgm, w, copy[8], dout, copy[7], want_absmax=True, n_dev=copy[5])
        b = ops.mlp_bwd(cfgm, w, fi, dout, rows[7], want_absmax=True, n_dev=rows[5])
        assert torch.equal(a[0][:, :n_kept], b[0][:, :n_kept]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), (nh, n_out)
    # (a grid of another depth takes the kernel without the scalar-base addressing: same contract)
    cfgm = MlpConfig(n_levels=12, n_hidden_layers=1, n_output_dims=1, output_activation='None', exp_shift=0.0)
    w = torch.cat([(torch.rand(o * i, generator=g) * 2 - 1) * math.sqrt(6.0 / (i + o)) for o, i in cfgm.shapes]).to(torch.bfloat16).cuda()
    f12 = feat[:12].contiguous()
    idx = fi.index[:n_kept].contiguous()
    dout = torch.randn(n_kept, 1, generator=g).cuda()
    a = ops.mlp_bwd(cfgm, w, f12.index_select(1, idx.long()), dout)
    b = ops.mlp_bwd(cfgm, w, ops.IndexedFeat(f12, idx), dout)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_composite_fwd_bwd(ops):
    packed, ri, ts, te, sig, rgb = _packed_case(2)
    sig = sig * 0.05
    R = packed.shape[0]
    sr = sig.clone().requires_grad_(True); cr = rgb.clone().requires_grad_(True)
    w, T, al = O.render_weight_from_density(ts, te, sr, packed.numpy())
    op = O.accumulate_along_rays(w, None, ri, R)
    dist = O.accumulate_along_rays(w, ((ts + te) / 2)[:, None], ri, R)
    col = O.accumulate_along_rays(w.detach(), cr, ri, R)
    gw, gT, gal, gop, gdist, gcol = ops.composite_fwd(sig.cuda(), rgb.cuda(), ts.cuda(), te.cuda(), packed.cuda())
    # fp32, sums of <=200 terms in a different association order: 1e-5 absolute on O(1) quantities
    for a, b in ((gw, w), (gT, T), (gal, al), (gop, op), (gdist, dist), (gcol, col)):
        assert (a.cpu() - b.detach()).abs().max() < 1e-5
    g = torch.Generator().manual_seed(3)
    g_w = torch.randn(sig.numel(), generator=g); g_op = torch.randn(R, 1, generator=g)
    g_d = torch.randn(R, 1, generator=g); g_c = torch.randn(R, 3, generator=g)
    ((w * g_w).sum() + (op * g_op).sum() + (dist * g_d).sum() + (col * g_c).sum()).backward(retain_graph=True)
    g_T = torch.randn(sig.numel(), generator=g); g_a = torch.randn(sig.numel(), generator=g)
    ((T * g_T).sum() + (al * g_a).sum()).backward()
    ds, dr = ops.composite_bwd(sig.cuda(), ts.cuda(), te.cuda(), packed.cuda(), gw, gT, g_w.cuda(), g_op.cuda(), g_d.cuda(),
                               g_c.cuda(), g_trans=g_T.cuda(), g_alphas=g_a.cuda(), want_drgb=True)
    assert (ds.cpu() - sr.grad).abs().max() < 2e-5 * max(1.0, float(sr.grad.abs().max()))
    assert (dr.cpu() - cr.grad).abs().max() < 1e-5


def test_distloss(ops):
    packed, ri, ts, te, sig, rgb = _packed_case(4)
    R = packed.shape[0]
    w = (torch.rand(sig.numel(), generator=torch.Generator().manual_seed(1)) * 0.02).requires_grad_(True)
    loss = O.flatten_eff_distloss(w, (ts + te) * .5, te - ts, ri)
    loss.backward()
    n_rays = int(ri.max()) + 1
    per_ray = ops.distloss_fwd(w.detach().cuda(), ts.cuda(), te.cuda(), packed.cuda())
    assert abs(float(per_ray.sum()) / n_rays - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    gw = ops.distloss_bwd(w.detach().cuda(), ts.cuda(), te.cuda(), packed.cuda(), 1.0 / n_rays)
    assert (gw.cpu() - w.grad).abs().max() < 1e-5 * max(1.0, float(w.grad.abs().max()))


def test_hashgrid_bwd_accumulates(ops):
    cfg = _grid_cfg()
    g = torch.Generator().manual_seed(11)
    n = 5000
    x = torch.rand(n, 3, generator=g).cuda()
    dfeat = torch.randn(cfg.n_levels, n, 2, generator=g).cuda()
    g1 = ops.hashgrid_bwd(cfg, x, dfeat)
    acc = g1.clone()
    ops.hashgrid_bwd(cfg, x, dfeat, out=acc, accumulate=True)
    assert (acc - 2 * g1).abs().max() < 1e-4 * float(g1.abs().max())
    # masked samples (zero incoming gradient) contribute nothing; n == 0 writes an all-zero table
    z = ops.hashgrid_bwd(cfg, x, torch.zeros_like(dfeat))
    assert float(z.abs().max()) == 0.0
    z0 = ops.hashgrid_bwd(cfg, x[:0], dfeat[:, :0])
    assert z0.numel() == cfg.n_params and float(z0.abs().max()) == 0.0


def test_accumulate_and_pack_info(ops):
    packed, ri, ts, te, sig, rgb = _packed_case(6)
    R = packed.shape[0]
    w = torch.rand(sig.numel(), generator=torch.Generator().manual_seed(2))
    assert torch.equal(ops.pack_info(ri.cuda(), R).cpu(), packed)
    assert torch.equal(ops.pack_info(ri[:0].cuda(), R).cpu()[:, 1], torch.zeros(R, dtype=torch.int32))
    for vals in (None, rgb):
        ref = O.accumulate_along_rays(w, vals, ri, R)
        got = ops.accumulate_fwd(w.cuda(), None if vals is None else vals.cuda(), packed.cuda())
        assert (got.cpu() - ref).abs().max() < 1e-5 * max(1.0, float(ref.abs().max()))


def test_hashgrid_bwd_fixed_point_mode(ops):
    """Packed fixed-point accumulation (integer LDS atomics) against the fp32 mode: unit = 2^(h-31) of the level's
    max |dfeat| (rounded up to a power of two), so entry sums agree to ~sqrt(fan-in) units."""
    cfg = _grid_cfg()
    g = torch.Generator().manual_seed(21)
    n = 20000
    x = torch.rand(n, 3, generator=g).cuda()
    dfeat = (torch.randn(cfg.n_levels, n, 2, generator=g) * torch.logspace(-3, 1, cfg.n_levels)[:, None, None]).cuda()
    amax = torch.zeros(16, device='cuda'); amax[:cfg.n_levels] = dfeat.abs().amax(dim=(1, 2))
    ref = ops.hashgrid_bwd(cfg, x, dfeat)
    ops.overflow_flag(x.device).zero_()
    got = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax)
    assert int(ops.overflow_flag(x.device).item()) == 0
    for l in range(cfg.n_levels):
        lo, hi = 2 * int(cfg.offset[l]), 2 * (int(cfg.offset[l]) + int(cfg.size[l]))
        h = min(24, max(12, math.ceil(math.log2(max(8.0 * n / int(cfg.size[l]), 1.0))) + 6))      # headroom rule of perf_hashgrid_bwd
        unit = float(2.0 ** torch.ceil(torch.log2(amax[l])) * 2.0 ** (h - 31))
        fan = 8.0 * n / int(cfg.size[l]) + 8
        err = float((got[lo:hi] - ref[lo:hi]).abs().max())
        assert err <= unit * (4 * fan ** 0.5 + 4), (l, err, unit)
    # the per-level max the MLP backward reports is what the fixed-point scale is derived from
    from perf_amd.grid import MlpConfig
    mlp = MlpConfig(16, 1, 1, 'Exponential')
    w = (torch.randn(mlp.n_params, generator=g) * 0.3).to(torch.bfloat16).cuda()
    feat = torch.rand(16, n, 2, generator=g).to(torch.bfloat16).cuda()
    dfe, dw, am = ops.mlp_bwd(mlp, w, feat, torch.randn(n, 1, generator=g).cuda(), want_absmax=True)
    true = dfe.abs().amax(dim=(1, 2))
    am = am[:16]
    assert bool((am >= true).all()) and float(am.max()) == float(true.max())      # a bound per level, tight overall


def test_hashgrid_fwd2_equals_two_single_passes(ops):
    cfg = _grid_cfg()
    g = torch.Generator().manual_seed(31)
    ta = (torch.rand(cfg.n_params, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    tb = (torch.rand(cfg.n_params, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    x = torch.rand(4099, 3, generator=g).cuda()
    fa, fb = ops.hashgrid_fwd2(cfg, x, ta, tb)
    assert torch.equal(fa, ops.hashgrid_fwd(cfg, x, ta)) and torch.equal(fb, ops.hashgrid_fwd(cfg, x, tb))


def test_pdf_resample_bit_exact(ops):
    """Hierarchical resampling (a7): inverse-CDF edges, bit-exact against the oracle's unfused fp32 definition."""
    rng = np.random.RandomState(0)
    for (R, n_in, n_out, strat) in ((37, 128, 64, False), (500, 64, 64, True), (3, 1, 5, True)):
        w = rng.rand(R, n_in).astype(np.float32) ** 4 + 1e-6
        w[0, : n_in // 2] = 0                                  # empty leading intervals (flat CDF)
        cdf = np.concatenate([np.zeros((R, 1), np.float32), np.cumsum(w, 1) / np.sum(w, 1, keepdims=True)], 1).astype(np.float32)
        cdf[:, -1] = 1.0
        s = np.sort(rng.rand(R, n_in + 1).astype(np.float32), 1)
        tau = rng.rand(R).astype(np.float32) if strat else None
        ref = O.pdf_resample(s, cdf, n_out, tau)
        got = ops.pdf_resample(torch.from_numpy(s).cuda(), torch.from_numpy(cdf).cuda(), n_out,
                               None if tau is None else torch.from_numpy(tau).cuda()).cpu().numpy()
        assert np.array_equal(got, ref)
        assert (np.diff(got, axis=1) >= 0).all()               # sorted edges: a size-independent property


def test_hashgrid_bwd_coded_owners_equal_streaming_owners(ops):
    """The hashed owners that stream 4-byte tile codes (default) and the ones that stream positions compute the same
    fixed-point sums: bit-identical tables, also for ragged sizes, ray-coherent bursts and positions outside the unit
    cube (which send a level back to the generic owners through the escape flag)."""
    cfg = _grid_cfg()
    g = torch.Generator().manual_seed(21)
    for n, kind in ((4099, 'uniform'), (65536 + 3, 'rays'), (30001, 'outside')):
        if kind == 'rays':
            R = n // 128 + 1
            d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
            t = (torch.arange(128) + 0.5) / 128
            x = ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5)[:n].contiguous()
        else:
            x = torch.rand(n, 3, generator=g)
        if kind == 'outside':
            x[::7, 0] = -0.25
            x[5::11, 0] = 9.5
            x[3::13, 1] = -3.0
        x = x.cuda()
        dfeat = torch.randn(cfg.n_levels, n, 2, generator=g).cuda()
        amax = dfeat.abs().amax(dim=(1, 2)).contiguous()
        amax = torch.cat([amax, torch.zeros(16 - amax.numel(), device='cuda')])
        a_fix = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax)
        a_f32 = ops.hashgrid_bwd(cfg, x, dfeat)
        # a workspace without room for the tile codes selects the position-streaming owners (include/perf_hip.h)
        b_fix = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax, use_codes=False)
        b_f32 = ops.hashgrid_bwd(cfg, x, dfeat, use_codes=False)
        assert torch.equal(a_fix, b_fix), (kind, float((a_fix - b_fix).abs().max()))
        assert float((a_f32 - b_f32).abs().max()) <= 1e-4 * float(b_f32.abs().max()), kind


def test_hashgrid_bwd_run_merging_owners_equal_streaming_owners(ops):
    """The coarse owners that sum runs of samples sharing a cell in registers before they touch LDS compute the same
    fixed-point sums whatever the runs are -- the batch in ray order (long runs) and in a random order (no runs) -- and as the
    position-streaming owners: bit-identical tables -- ragged sizes,
    ray-coherent runs, positions outside the unit cube (index wrap), a live count below the capacity, and the full
    1 M-sample batch of the benchmark."""
    cfg = _grid_cfg()
    g = torch.Generator().manual_seed(22)
    for n, kind, live in ((4099, 'uniform', None), (65536 + 3, 'rays', None), (30001, 'outside', None), (50000, 'rays', 31111),
                          (1 << 20, 'rays', None), (1 << 20, 'rays', 700001)):
        if kind == 'rays':
            R = n // 128 + 1
            d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
            t = (torch.arange(128) + 0.5) / 128
            x = ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5)[:n].contiguous()
        else:
            x = torch.rand(n, 3, generator=g)
        if kind == 'outside':
            x[::7, 0] = -0.25
            x[5::11, 0] = 9.5
            x[3::13, 1] = -3.0
        x = x.cuda()
        dfeat = torch.randn(cfg.n_levels, n, 2, generator=g).cuda()
        amax = dfeat.abs().amax(dim=(1, 2)).contiguous()
        amax = torch.cat([amax, torch.zeros(16 - amax.numel(), device='cuda')])
        n_dev = None if live is None else torch.tensor([live], dtype=torch.int64, device='cuda')
        a_fix = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax, n_dev=n_dev)
        a_f32 = ops.hashgrid_bwd(cfg, x, dfeat, n_dev=n_dev)
        # the same live samples in a RANDOM order: runs of samples that share a cell fall apart (nothing is merged in registers),
        # the replicas of the coarse levels get other samples, the queues fill differently -- integer sums must not notice
        m = n if live is None else live
        perm = torch.randperm(m, generator=g).cuda()
        xp = x.clone(); xp[:m] = x[perm]
        dp = dfeat.clone(); dp[:, :m] = dfeat[:, perm]
        c_fix = ops.hashgrid_bwd(cfg, xp, dp, level_absmax=amax, n_dev=n_dev)
        b_fix = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax, use_codes=False, n_dev=n_dev)
        b_f32 = ops.hashgrid_bwd(cfg, x, dfeat, use_codes=False, n_dev=n_dev)
        assert torch.equal(a_fix, b_fix), (n, kind, float((a_fix - b_fix).abs().max()))
        assert torch.equal(c_fix, b_fix), (n, kind, float((c_fix - b_fix).abs().max()))
        assert float((a_f32 - b_f32).abs().max()) <= 1e-4 * float(b_f32.abs().max()), (n, kind)


# ---- deep / large grids (BASELINE config 5: L = 20, tables beyond 2^32 entries; inference only) ----------------------
def test_deep_grid_forward_20_levels(ops):
    """L = 20 (three k-steps of MLP input, a third encode pass per level group) against the oracle."""
    cfg = _grid_cfg(n_levels=20, log2_hashmap_size=15, base_resolution=16, per_level_scale=1.3819)
    lv = O.grid_levels(20, 2, 15, 16, 1.3819)
    assert cfg.total == lv.total and np.array_equal(cfg.offset, lv.offset)
    g = torch.Generator().manual_seed(31)
    table = torch.rand(cfg.total, 2, generator=g) * 2 - 1
    n = 2500
    x = torch.rand(n, 3, generator=g)
    t16 = table.to(torch.bfloat16)
    feat = ops.hashgrid_fwd(cfg, x.cuda(), t16.reshape(-1).cuda())
    assert feat.shape == (20, n, 2)
    got = feat.float().cpu().permute(1, 0, 2).reshape(n, -1)
    ref = O.hashgrid_encode(x, table, lv, quant='bf16')
    assert ((got - ref).abs() <= 2.0 ** -8 * ref.abs() * 1.01 + 1e-6).all()
    # MLP forward on 40 (padded to 48) inputs, both depths
    for nh, n_out, act in ((1, 1, 'Exponential'), (2, 3, 'Sigmoid')):
        cfgm, w, f, sel, tdt, ulp = _mlp_case(ops, 'bf16', nh, n_out, act, n=1500, n_levels=20, seed=3)
        assert cfgm.n_in_pad == 48
        out = ops.mlp_fwd(cfgm, w.to(tdt).cuda(), f.to(tdt).cuda(), sel.cuda()).cpu()
        ref = _mlp_oracle(cfgm, w.to(tdt).float(), f.to(tdt).float(), sel, 'bf16')
        assert (out - ref).abs().max() < 8 * ulp * max(1.0, float(ref.abs().max()))
    # (the backward of such a field: test_mlp_more_than_16_levels, test_network_with_20_level_grid_forward_and_gradient)


@pytest.mark.parametrize('layout', ['tcnn', 'line_local', 'line_overlap'])
@pytest.mark.parametrize('log2_t,sb_shift,min_res', [(15, (2, 2, 1), 16), (20, (3, 3, 2), 64), (24, (5, 6, 8), 64), (24, (7, 5, 7), 64)])
def test_deep_grid_forward_both_layouts(ops, layout, log2_t, sb_shift, min_res):
    """The deep-grid forward kernel (one level per workgroup, XCD-stable balanced; 16-byte x-runs on line-local levels) against
    the oracle -- corner indices bit-exact, features within an ulp of the storage type -- for tcnn's layout and for the opt-in
    line-local ones (oracle/perf_oracle.py:grid_levels: 'line_local', and 'line_overlap' whose x runs overlap by one vertex), at table sizes where the line-local levels are all hashed (2^15, super-blocks
    of one block), mixed dense / hashed (2^20) and with the shipped 32 x 64 x 256 / 128 x 32 x 128 super-blocks (2^24); ragged n, points on cell and
    block boundaries, a device-side live count."""
    L, b = 20, 1.3819
    cfg = _grid_cfg(n_levels=L, log2_hashmap_size=log2_t, base_resolution=16, per_level_scale=b, layout=layout, sb_shift=sb_shift,
                    local_min_res=min_res)
    lv = _lv_of(cfg)
    assert cfg.total == lv.total and np.array_equal(cfg.offset, lv.offset) and np.array_equal(cfg.size, lv.size)
    if layout != 'tcnn':
        assert int(cfg.local.sum()) == int((cfg.res >= min_res).sum()) > 0
        assert log2_t == 15 or (int(((cfg.local == 1) & (cfg.hashed == 0)).sum()) > 0 and int(((cfg.local == 1) & (cfg.hashed == 1)).sum()) > 0)
    g = torch.Generator().manual_seed(41 + log2_t)
    n = 3001
    x = torch.rand(n, 3, generator=g)
    # points ON vertices of several levels (fraction 0: the cell's first vertex; x-runs that start at a block's last vertex)
    for k, l in enumerate((5, 9, 14, 19)):
        v = torch.randint(0, int(cfg.res[l]) - 1, (40, 3), generator=g).float()
        v[:20, 0] = (v[:20, 0] // 4) * 4 + 3                                   # first vertex = last of its block along x
        cells_per_row = 3 << (sb_shift[0] - 2)                                  # (line_overlap: the last cell of a super-block row and the cell behind it)
        v[20:30, 0] = (v[20:30, 0] // cells_per_row) * cells_per_row + cells_per_row - 1
        v[30:40, 0] = (v[30:40, 0] // cells_per_row) * cells_per_row
        v[:, 0] = v[:, 0].clamp(0, int(cfg.res[l]) - 2)
        x[100 * k: 100 * k + 40] = ((v - 0.5) / float(cfg.scale[l])).clamp(0.0, 0.999999)
    x[-1] = torch.tensor([0.9999999, 0.9999999, 0.9999999])                      # the last cell of every level
    xd = x.cuda()
    idx = ops.hashgrid_corners(cfg, xd).cpu().numpy()                            # [L, n, 8] absolute entries
    xn = x.numpy()
    for l in range(L):
        ref, _ = O.grid_corner_indices(xn, lv, l)
        assert np.array_equal(idx[l].astype(np.int64), ref.astype(np.int64) + int(lv.offset[l])), (layout, l)
        assert ref.max() < lv.size[l]
    for dt in ('fp16', 'bf16'):
        tdt, ulp = DT[dt]
        table = (torch.rand(cfg.total, 2, generator=g) * 2 - 1)
        feat = ops.hashgrid_fwd(cfg, xd, table.to(tdt).reshape(-1).cuda())
        assert feat.shape == (L, n, 2)
        got = feat.float().cpu().permute(1, 0, 2).reshape(n, -1)
        ref = O.hashgrid_encode(x, table, lv, quant=dt)
        assert ((got - ref).abs() <= ulp * ref.abs() * 1.01 + 1e-6).all(), (layout, dt, float((got - ref).abs().max()))
        # capacity-sized launch with a device-side live count: the live rows are the same bits
        live = 1777
        t16 = table.to(tdt).reshape(-1).cuda()
        part = ops.hashgrid_fwd(cfg, xd, t16, n_dev=torch.tensor([live], dtype=torch.int64, device='cuda'))
        assert torch.equal(part[:, :live], feat[:, :live])
        # tiny and ragged launches (one sample; less than a 16-sample group; one group short of a wave; around the ends of a wave's
        # step of 64, of its four steps and of a workgroup's 1,024 samples), live counts that end on those boundaries, an empty live count
        for m in (1, 15, 70, 255, 256, 257, 1023, 1024, 1025):
            assert torch.equal(ops.hashgrid_fwd(cfg, xd[:m].contiguous(), t16), feat[:, :m]), (layout, dt, m)
        for live2 in (64, 1024, 1088, 2047):
            part = ops.hashgrid_fwd(cfg, xd, t16, n_dev=torch.tensor([live2], dtype=torch.int64, device='cuda'))
            assert torch.equal(part[:, :live2], feat[:, :live2]), (layout, dt, live2)
        canary = ops.hashgrid_fwd(cfg, xd[:300].contiguous(), t16, n_dev=torch.tensor([0], dtype=torch.int64, device='cuda'))
        assert canary.shape == (L, 300, 2)
        # points OUTSIDE the unit cube (a caller's positions beyond its box: their selector is 0, their features are never used) read
        # inside the table -- densely addressed line-local levels clamp the entry -- and leave the other rows alone
        xo = xd[:512].clone()
        xo[::7] = torch.tensor([[-0.31, 0.5, 1.7], [5.0, -2.0, 0.2], [1.0001, 1.0001, 1.0001], [-1e-4, 0.3, 0.9]], device='cuda').repeat(19, 1)[:xo[::7].shape[0]]
        fo = ops.hashgrid_fwd(cfg, xo, t16)
        keep = torch.ones(512, dtype=torch.bool, device='cuda'); keep[::7] = False
        assert torch.equal(fo[:, keep], feat[:, :512][:, keep]) and bool(torch.isfinite(fo.float()).all())
    if layout != 'tcnn':
        # inference only: the gradient entry points refuse the layout instead of scattering into the wrong entries
        from perf_amd._lib import PerfError
        with pytest.raises(PerfError):
            ops.hashgrid_bwd(cfg, xd, torch.zeros(L, n, 2, device='cuda'))


def test_overlapping_runs_hold_one_field(ops):
    """layout='line_overlap' stores the vertex two neighbouring x runs share TWICE; GridConfig.canonicalize_ (the oracle's
    canonical_overlap_fill on the device) makes the copies equal.  The table is then ONE continuous field: points a hair to the left and to
    the right of every cell face along x -- run boundaries, super-block boundaries, dense and hashed levels alike -- encode to the same
    features up to the step across the face, as with tcnn's layout; the uncanonical random table jumps by O(1) at every third face."""
    L, b = 12, 1.3819
    cfg = _grid_cfg(n_levels=L, log2_hashmap_size=20, base_resolution=16, per_level_scale=b, layout='line_overlap', sb_shift=(3, 3, 2))
    lv = _lv_of(cfg)
    assert int(((cfg.local == 1) & (cfg.hashed == 0)).sum()) > 0 and int(((cfg.local == 1) & (cfg.hashed == 1)).sum()) > 0
    g = torch.Generator().manual_seed(7)
    raw = (torch.rand(cfg.total, 2, generator=g) * 2 - 1)
    canon = cfg.canonicalize_(raw.clone().cuda())
    assert np.array_equal(canon.cpu().numpy(), O.canonical_overlap_fill(raw.numpy(), lv))
    assert torch.equal(cfg.canonicalize_(canon.clone()), canon)
    for l in range(L):
        if not cfg.local[l]:
            continue
        sc, r = float(cfg.scale[l]), int(cfg.res[l])
        n = 4000
        v = torch.stack([torch.randint(1, r - 1, (n,), generator=g), torch.randint(0, r - 1, (n,), generator=g), torch.randint(0, r - 1, (n,), generator=g)], -1).float()
        yz = torch.rand(n, 2, generator=g) * 0.8 + 0.1
        eps = 2e-3                                                     # in cells
        def pts(dx):
            p = v.clone(); p[:, 0] += dx; p[:, 1:] += yz
            return ((p - 0.5) / sc).clamp(0.0, 1.0)
        xl, xr = pts(-eps), pts(+eps)
        gl_, gr_ = np.floor(O.grid_pos(xl.numpy(), cfg.scale[l]))[:, 0], np.floor(O.grid_pos(xr.numpy(), cfg.scale[l]))[:, 0]
        ok = torch.from_numpy((gr_ == gl_ + 1) & (gr_ == v[:, 0].numpy()))          # the two points straddle the face at vertex v
        assert int(ok.sum()) > 0.9 * n
        for table, one_field in ((canon, True), (raw.cuda(), False)):
            fl = ops.hashgrid_fwd(cfg, xl.cuda(), table.half().reshape(-1))[l].float().cpu()
            fr = ops.hashgrid_fwd(cfg, xr.cuda(), table.half().reshape(-1))[l].float().cpu()
            jump = (fl - fr).abs().max(-1).values[ok]
            if one_field:
                assert float(jump.max()) < 2 * eps * 2 * 2 + 2e-3, (l, float(jump.max()))       # |slope| <= 2 per cell and feature, fp16 storage
            else:
                third = (v[:, 0][ok] % 3 == 0)
                assert float(jump[third].mean()) > 0.1 and float(jump[~third].max()) < 2 * eps * 2 * 2 + 2e-3


@pytest.mark.parametrize('layout,T', [('tcnn', 29), ('tcnn', 30), ('line_local', 29), ('line_local', 30), ('line_overlap', 29)])
def test_table_beyond_32_bit_offsets(ops, layout, T):
    """5.6e9 entries (21 GiB of 2x16-bit features; T = 30: 1.1e10 entries, 41 GiB -- beyond BASELINE config 5's 31 GiB per encoder):
    level offsets exceed 2^32.  The table is filled on the device with a function of the global entry index; the expected
    features of 65,573 points -- half of them uniform in the cube, half consecutive samples along rays from the centre as a panorama
    batch holds them (neighbouring lanes share table lines; the deep-grid kernel's waves take several steps and end inside one) --
    are evaluated on the host from the oracle's corner indices and weights through the same function, so no host copy of the table
    is needed.  Every table layout."""
    L, b = 20, 1.5
    cfg = _grid_cfg(n_levels=L, log2_hashmap_size=T, base_resolution=16, per_level_scale=b, layout=layout)
    lv = _lv_of(cfg)
    assert cfg.total > 2 ** 32 and np.array_equal(cfg.offset.astype(np.uint64), lv.offset.astype(np.uint64))

    def value(idx):                                      # entry index (int64 tensor) -> feature 0, feature 1 in [-1, 1)
        h = (idx * 40503 + 12345) % 65521
        return h.float() / 32760.5 - 1.0, ((h * 7 + 3) % 65521).float() / 32760.5 - 1.0

    table = torch.empty(cfg.total * 2, dtype=torch.float16, device='cuda')
    step = 1 << 27
    for lo in range(0, cfg.total, step):
        idx = torch.arange(lo, min(lo + step, cfg.total), device='cuda', dtype=torch.int64)
        f0, f1 = value(idx)
        table[2 * lo: 2 * (lo + idx.numel())] = torch.stack([f0, f1], -1).reshape(-1).half()
        del idx, f0, f1
    g = torch.Generator().manual_seed(37)
    x_uniform = torch.rand(32768 + 37, 3, generator=g)
    d = torch.nn.functional.normalize(torch.randn(128, 3, generator=g), dim=-1)
    t = (torch.arange(256, dtype=torch.float32) + 0.5) * (0.99 / 256)
    x_rays = (0.5 + 0.5 * d[:, None, :] * t[None, :, None]).reshape(-1, 3).clamp(0.0, 1.0)          # 128 rays x 256 samples, packed by ray
    x = torch.cat([x_rays, x_uniform])
    n = x.shape[0]
    feat = ops.hashgrid_fwd(cfg, x.cuda(), table).float().cpu()          # [L, n, 2]
    xn = x.numpy()
    for l in range(L):
        idx, f = O.grid_corner_indices(xn, lv, l)
        gi = torch.from_numpy(idx.astype(np.int64)) + int(lv.offset[l])
        v0, v1 = value(gi)
        v0 = v0.half().float(); v1 = v1.half().float()
        w = torch.ones(n, 8)
        for c in range(8):
            for a in range(3):
                fa = torch.from_numpy(f[:, a].astype(np.float32))
                w[:, c] = w[:, c] * (fa if (c >> a) & 1 else (1.0 - fa))
        ref = torch.stack([(w * v0).sum(1), (w * v1).sum(1)], -1)
        assert (feat[l] - ref).abs().max() < 2e-3, (l, float((feat[l] - ref).abs().max()))
    del table
    torch.cuda.empty_cache()


@pytest.mark.parametrize('bitmap', ['1', '0'])
def test_hashgrid_bwd_large_levels(ops, bitmap):
    """log2_hashmap_size = 23: 512 tiles per hashed level, more than code-streaming owners are planned for -> LDS owners fed
    by per-tile bitmaps (default) or the global-atomics scatter (bitmap '0': whenever the workspace cannot hold
    the bitmaps) for those levels, the usual owners for the rest; against the oracle's corner bookkeeping (fp32:
    summation-order tolerance)."""
    no_bitmaps = bitmap == '0'          # (a workspace without room for codes / bitmaps: atomics scatter for the large levels)
    cfg = _grid_cfg(n_levels=6, log2_hashmap_size=23, base_resolution=32, per_level_scale=2.0)
    lv = O.grid_levels(6, 2, 23, 32, 2.0)
    g = torch.Generator().manual_seed(41)
    n = 3000
    x = torch.rand(n, 3, generator=g)
    dfeat = torch.randn(cfg.n_levels, n, 2, generator=g)
    for fixed in (False, True):
        amax = None
        if fixed:
            amax = torch.zeros(24, device='cuda'); amax[:6] = dfeat.abs().amax(dim=(1, 2)).cuda()
        grad = ops.hashgrid_bwd(cfg, x.cuda(), dfeat.cuda(), level_absmax=amax, use_codes=not no_bitmaps).cpu().numpy().reshape(-1, 2)
        ref = np.zeros((cfg.total, 2), np.float64)
        xn = x.numpy()
        for l in range(cfg.n_levels):
            idx, f = O.grid_corner_indices(xn, lv, l)
            for c in range(8):
                w = np.ones(n, np.float32)
                for a in range(3):
                    w = w * (f[:, a] if (c >> a) & 1 else (np.float32(1) - f[:, a]))
                np.add.at(ref, idx[:, c].astype(np.int64) + int(lv.offset[l]), w[:, None].astype(np.float64) * dfeat[l].numpy())
        assert np.abs(grad - ref).max() < 2e-4 * np.abs(ref).max()
        if fixed:       # 64-bit fixed-point global atomics: the sum does not depend on the order they retire in
            again = ops.hashgrid_bwd(cfg, x.cuda(), dfeat.cuda(), level_absmax=amax, use_codes=not no_bitmaps).cpu().numpy().reshape(-1, 2)
            assert np.array_equal(grad, again)
        # accumulate adds on top
        acc = torch.from_numpy(grad.reshape(-1).copy()).cuda()
        ops.hashgrid_bwd(cfg, x.cuda(), dfeat.cuda(), out=acc, accumulate=True, level_absmax=amax, use_codes=not no_bitmaps)
        assert np.abs(acc.cpu().numpy().reshape(-1, 2) - 2 * ref).max() < 4e-4 * np.abs(ref).max()


def test_hashgrid_bwd_bitmap_owners_equal_the_atomics_scatter(ops):
    """Hashed levels of 256-2048 tiles (log2_hashmap_size 22-25) and dense levels of 32-2048 tiles: the owners that read per-tile
    bitmaps and what runs without them (64-bit fixed-point global atomics; code-streaming owners up to 64 dense tiles) add up the same integers -- bit-identical tables, for ragged sizes, ray-ordered samples, a live count
    below the capacity and positions outside the unit cube (which send a level back to the generic owners)."""
    g = torch.Generator().manual_seed(43)
    for log2_t, n_levels, n, kind, live, base in ((22, 5, 5003, 'uniform', None, 64), (23, 6, 40000, 'rays', None, 64), (24, 4, 20001, 'uniform', 12345, 64),
                                                  (22, 5, 9000, 'outside', None, 64), (25, 3, 70000, 'rays', None, 64), (22, 3, 30000, 'rays', None, 96),
                                                  (22, 3, 8000, 'outside', 7000, 96)):
        # (base 64: 16 dense tiles with codes, 128 dense tiles, then hashed levels of 256-2048 tiles; base 96: 64 dense tiles --
        #  bitmaps against the code-streaming dense owners)
        cfg = _grid_cfg(n_levels=n_levels, log2_hashmap_size=log2_t, base_resolution=base, per_level_scale=2.0)
        if kind == 'rays':
            R = n // 128 + 1
            d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
            t = (torch.arange(128) + 0.5) / 128
            x = ((d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5)[:n].contiguous()
        else:
            x = torch.rand(n, 3, generator=g)
        if kind == 'outside':
            x[::7, 0] = -0.25
            x[5::11, 0] = 9.5
            x[3::13, 1] = -3.0
        x = x.cuda()
        dfeat = torch.randn(cfg.n_levels, n, 2, generator=g).cuda()
        amax = torch.zeros(24, device='cuda'); amax[:cfg.n_levels] = dfeat.abs().amax(dim=(1, 2))
        n_dev = None if live is None else torch.tensor([live], dtype=torch.int64, device='cuda')
        a = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax, n_dev=n_dev)
        a32 = ops.hashgrid_bwd(cfg, x, dfeat, n_dev=n_dev)
        # (a workspace without room for the codes and the bitmaps: global atomics for the large levels, position-streaming owners for the rest)
        b = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax, n_dev=n_dev, use_codes=False)
        b32 = ops.hashgrid_bwd(cfg, x, dfeat, n_dev=n_dev, use_codes=False)
        assert float(b.abs().max()) > 0
        assert torch.equal(a, b), (log2_t, kind, float((a - b).abs().max()))
        assert float((a32 - b32).abs().max()) <= 1e-4 * float(b32.abs().max()), (log2_t, kind)
        del a, b, a32, b32, dfeat
    torch.cuda.empty_cache()


@pytest.mark.parametrize('large', [False, True])
def test_job_wide_units_make_two_partial_tables_add_up_to_the_single_call(ops, large):
    """The data-parallel kernels on their own (perf_amd/dp.py's choreography without a process group): a batch cut in two
    uneven parts, perf_dp_stats_pack per part, perf_dp_units over both -> the units the single call derives itself; the
    parts' raw int32 fields added as integers and converted by perf_fixed_unfix equal the single call's table BIT FOR BIT --
    on the benchmark's grid and on a grid whose large levels take the bitmap owners."""
    cfg = _grid_cfg(n_levels=5, log2_hashmap_size=22, base_resolution=64, per_level_scale=2.0) if large else _grid_cfg()
    g = torch.Generator().manual_seed(51)
    n, cut = 30011, 11003
    x = torch.rand(n, 3, generator=g).cuda()
    dfeat = (torch.randn(cfg.n_levels, n, 2, generator=g) * 1e-3).cuda()

    def amax_of(d):
        a = torch.zeros(24, device='cuda'); a[:cfg.n_levels] = d.abs().amax(dim=(1, 2)); return a
    ref = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax_of(dfeat), hr_state=ops.headroom_state('cuda'))
    parts = [(x[:cut].contiguous(), dfeat[:, :cut].contiguous()), (x[cut:].contiguous(), dfeat[:, cut:].contiguous())]
    stats = torch.stack([ops.dp_stats_pack(amax_of(d), None, None, xs.shape[0]) for xs, d in parts])
    hr = ops.headroom_state('cuda')
    shifts, n_total = ops.dp_units(cfg, stats, 2, hr)
    assert int(n_total.item()) == n
    total = torch.zeros(cfg.n_params, dtype=torch.int32, device='cuda')
    for xs, d in parts:
        f = ops.hashgrid_bwd(cfg, xs, d, shifts=shifts, raw_fields=True)
        total += f.view(torch.int32)
    field_max = torch.zeros(24, dtype=torch.int32, device='cuda')
    ops.fixed_unfix(cfg, total, 0, cfg.total, shifts, field_max=field_max)
    got = total.view(torch.float32)
    assert float(ref.abs().max()) > 0
    assert torch.equal(got, ref), float((got - ref).abs().max())
    assert int(field_max[:cfg.n_levels].min()) > 0


def test_composite_distloss_fused_kernels(ops):
    """Compositing + distortion loss in one kernel each way equals the two-kernel chain (forward bit-exact; backward to
    fp32 rounding: prefixes are formed as totals minus suffixes there)."""
    packed, ri, ts, te, sig, rgb = _packed_case(9)
    c = lambda t: t.cuda()
    w, T, al, op, dist, col = ops.composite_fwd(c(sig), c(rgb), c(ts), c(te), c(packed))
    dl = ops.distloss_fwd(w, c(ts), c(te), c(packed))
    w2, T2, op2, dist2, col2, dl2 = ops.composite_distloss_fwd(c(sig), c(rgb), c(ts), c(te), c(packed))
    for a, b in ((w, w2), (T, T2), (op, op2), (dist, dist2), (col, col2), (dl, dl2)):
        assert torch.equal(a, b)
    R = packed.shape[0]
    g = torch.Generator().manual_seed(10)
    g_op = torch.randn(R, 1, generator=g).cuda(); g_d = torch.randn(R, 1, generator=g).cuda()
    scale_dev = torch.tensor([0.37], device='cuda')
    g_w = ops.distloss_bwd(w, c(ts), c(te), c(packed), 2.0, scale_dev=scale_dev)
    ref, _ = ops.composite_bwd(c(sig), c(ts), c(te), c(packed), w, T, g_weights=g_w, g_opacity=g_op, g_distance=g_d)
    got = ops.composite_distloss_bwd(c(sig), c(ts), c(te), c(packed), w, T, op, dist, g_op, g_d, 2.0, scale_dev=scale_dev)
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-7


def test_march_write_with_points_equals_two_kernels(ops):
    g = torch.Generator().manual_seed(12)
    R, res = 500, 32
    occ = (torch.rand(res ** 3, generator=g) < 0.2).to(torch.uint8)
    o = (torch.rand(R, 3, generator=g) - 0.5) * 0.3
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    t0 = torch.rand(R, generator=g) * 0.01
    aabb = [-1., -1, -1, 1, 1, 1]
    bits = ops.occ_pack_bits(occ.cuda())
    ri, ts, te, packed = ops.occ_march(o.cuda(), d.cuda(), t0.cuda(), bits, res, aabb, 1.5, 1e-2, 151)
    x_ref, s_ref = ops.points_from_rays(o.cuda(), d.cuda(), ri, ts, te, aabb)
    ri2, ts2, te2, packed2, x01, sel = ops.occ_march(o.cuda(), d.cuda(), t0.cuda(), bits, res, aabb, 1.5, 1e-2, 151, points_aabb=aabb)
    assert torch.equal(ri, ri2) and torch.equal(ts, ts2) and torch.equal(te, te2) and torch.equal(packed, packed2)
    assert torch.equal(x01, x_ref) and torch.equal(sel, s_ref)


def test_hashgrid_bwd_fixed_point_with_vanishing_gradients(ops):
    """Gradients of 1e-30 .. 1e-42 (and exactly zero) must neither raise the overflow flag nor produce non-finite sums."""
    cfg = _grid_cfg()
    g = torch.Generator().manual_seed(23)
    n = 3000
    x = torch.rand(n, 3, generator=g).cuda()
    for mag in (1e-30, 1e-38, 1e-42, 0.0):
        dfeat = (torch.randn(cfg.n_levels, n, 2, generator=g) * mag).cuda()
        amax = torch.zeros(24, device='cuda'); amax[:cfg.n_levels] = dfeat.abs().amax(dim=(1, 2))
        ops.overflow_flag(x.device).zero_()
        got = ops.hashgrid_bwd(cfg, x, dfeat, level_absmax=amax)
        assert int(ops.overflow_flag(x.device).item()) == 0, mag
        assert bool(torch.isfinite(got).all()) and float(got.abs().max()) <= 8 * n * max(mag, 1e-45) * 10


@pytest.mark.parametrize('max_short', [16, 4])
@pytest.mark.parametrize('n_short', [203, 36001])
def test_quarter_wave_ray_teams_equal_full_wave_rays(ops, n_short, max_short):
    """Per-ray kernels serve four rays per wave: 16 lanes each when all four hold <= 16 samples, all 64 lanes one ray after
    the other otherwise.  The same short rays, alone (quarter-wave path) and interleaved with long rays (full-wave path),
    must give bit-identical visibility counts, exclusive sums, weights and per-ray outputs -- and match the oracle's
    canonical scan."""
    # (203 rays: one wave per ray is launched; 36001 rays: the packed shape, four rays per wave)
    g = torch.Generator().manual_seed(31)
    # (max_short = 4: sixteen rays per wave in 4-lane teams)
    counts_s = torch.randint(0, max_short + 1, (n_short,), generator=g)
    counts_s[:5] = torch.tensor([0, 1, max_short, max_short, 2])

    def build(counts):
        starts = torch.cumsum(counts, 0) - counts
        S = int(counts.sum())
        packed = torch.stack([starts, counts], 1).to(torch.int32)
        return packed, S

    def fill(S):
        ts = torch.rand(S, generator=g) * 0.5
        te = ts + 0.01 + 0.01 * torch.rand(S, generator=g)
        sig = torch.exp(torch.randn(S, generator=g) * 2 + 3)
        rgb = torch.rand(S, 3, generator=g)
        return sig, ts, te, rgb

    packed_a, S_a = build(counts_s)
    sig_a, ts_a, te_a, rgb_a = fill(S_a)
    # the same rays with a 40-sample ray in front of every group of three
    long_cnt = 40
    order = []          # (is_long, index)
    k = 0
    while k < n_short:
        order.append((True, None))
        for _ in range(3):
            if k < n_short:
                order.append((False, k)); k += 1
    counts_b = torch.tensor([long_cnt if is_long else int(counts_s[i]) for is_long, i in order])
    packed_b, S_b = build(counts_b)
    sig_b, ts_b, te_b, rgb_b = fill(S_b)
    pos_of = {}
    for j, (is_long, i) in enumerate(order):
        if not is_long:
            pos_of[i] = j
            s0, c = int(packed_b[j, 0]), int(packed_b[j, 1])
            a0 = int(packed_a[i, 0])
            sig_b[s0:s0 + c] = sig_a[a0:a0 + c]; ts_b[s0:s0 + c] = ts_a[a0:a0 + c]; te_b[s0:s0 + c] = te_a[a0:a0 + c]
            rgb_b[s0:s0 + c] = rgb_a[a0:a0 + c]
    thr = 9.2103
    res = {}
    for name, (packed, sig, ts, te, rgb) in {'a': (packed_a, sig_a, ts_a, te_a, rgb_a), 'b': (packed_b, sig_b, ts_b, te_b, rgb_b)}.items():
        pk = packed.cuda()
        nc, ex = ops.visibility_count(sig.cuda(), ts.cuda(), te.cuda(), pk, early_stop_eps=1e-4, want_exsum=True)
        w, T, al, op, dist, col = ops.composite_fwd(sig.cuda(), rgb.cuda().contiguous(), ts.cuda(), te.cuda(), pk)
        res[name] = dict(nc=nc.cpu(), ex=ex.cpu(), w=w.cpu(), T=T.cpu(), op=op.cpu(), dist=dist.cpu(), col=col.cpu())
    # oracle: canonical exclusive sums and kept counts of the short rays
    sd = (sig_a.numpy() * (te_a.numpy() - ts_a.numpy()).astype(np.float32)).astype(np.float32)
    ex_ref = O.packed_exclusive_sum_canonical(sd, packed_a.numpy())
    assert np.array_equal(res['a']['ex'].numpy(), ex_ref)
    keep, _ = O.visibility_keep_mask(sig_a.numpy(), ts_a.numpy(), te_a.numpy(), packed_a.numpy(), 1e-4)
    kept_ref = np.array([int(keep[s0:s0 + c].sum()) for s0, c in packed_a.numpy()])
    assert np.array_equal(res['a']['nc'].numpy(), kept_ref)
    for i in range(0, n_short, 1 if n_short < 1000 else 37):
        j = pos_of[i]
        a0, c = int(packed_a[i, 0]), int(packed_a[i, 1]); b0 = int(packed_b[j, 0])
        assert int(res['a']['nc'][i]) == int(res['b']['nc'][j])
        for k_ in ('ex', 'w', 'T'):
            assert torch.equal(res['a'][k_][a0:a0 + c], res['b'][k_][b0:b0 + c]), (i, k_)
        for k_ in ('op', 'dist', 'col'):
            assert torch.equal(res['a'][k_][i], res['b'][k_][j]), (i, k_)


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('net', ['geo', 'app'])
def test_fused_field_infer_equals_the_two_kernel_path(dtype, net):
    """perf_field_infer on a small batch = ONE kernel (encode straight into the first layer's MFMA operand registers); same
    outputs and, when asked for, the same level-major features as perf_hashgrid_fwd + perf_mlp_fwd, bit for bit."""
    from perf_amd import ops
    from perf_amd.grid import GridConfig, MlpConfig
    cfg = GridConfig()
    spec = O.geo_spec() if net == 'geo' else O.app_spec()
    params = O.init_field_params(spec); params[spec.n_net:] *= 1e4
    w16 = ops.cast_params(params.cuda(), dtype)
    mlp = MlpConfig(n_levels=16, n_hidden_layers=1, n_output_dims=1, output_activation='Exponential') if net == 'geo' else \
        MlpConfig(n_levels=16, n_hidden_layers=2, n_output_dims=3, output_activation='Sigmoid')
    g = torch.Generator().manual_seed(3)
    for n in (1, 37, 3000, ops.FUSED_MAX_SAMPLES):
        x = torch.rand(n, 3, generator=g).cuda()
        sel = (torch.rand(n, generator=g) > 0.1).to(torch.uint8).cuda()
        out, feat = ops.field_infer(cfg, mlp, x, sel, w16, want_features=True)          # fused (n <= FUSED_MAX_SAMPLES)
        out_only = ops.field_infer(cfg, mlp, x, sel, w16)
        ref_feat = ops.hashgrid_fwd(cfg, x, w16[spec.n_net:])
        ref = ops.mlp_fwd(mlp, w16[:spec.n_net], ref_feat, sel)
        assert torch.equal(feat, ref_feat) and torch.equal(out, ref) and torch.equal(out_only, ref), n
    # a device-side count below the capacity: only the live rows are produced
    n, live = 2048, 777
    x = torch.rand(n, 3, generator=g).cuda()
    nd = torch.tensor([live], dtype=torch.int64, device='cuda')
    out = ops.field_infer(cfg, mlp, x, None, w16, n_dev=nd)
    ref = ops.mlp_fwd(mlp, w16[:spec.n_net], ops.hashgrid_fwd(cfg, x[:live].contiguous(), w16[spec.n_net:]), None)
    assert torch.equal(out[:live], ref)


@pytest.mark.parametrize('lattice', ['repeated', 'single'])
@pytest.mark.parametrize('step', [5e-4, 2e-3])
def test_occ_march_both_lattices_bit_exact(ops, step, lattice):
    """PERF_LATTICE_REPEATED (t_0 = t0, t_{k+1} = fl(t_k + step): the lattice of a marcher that advances by `t += dt`; the
    default since round 4): the kernels evaluate t_k in closed form per binade; the oracle accumulates sequentially
    (np.add.accumulate in fp32).  PERF_LATTICE_SINGLE (t_k = fl(t0 + fl(k step)), rounds 1-3) stays selectable.  Samples,
    interval ends and ray bookkeeping equal bit for bit -- plain and in-kernel origins, with and without the skip grid, the
    head written by the counting pass included -- and the lattice really differs from the single-rounding one."""
    o, d, dist, rgb, occ = _room(24, 48, 64)
    o = o + torch.tensor([0.2, -0.1, 0.05])
    o[:5] = torch.tensor([3.0, 0.0, 0.0])
    R = o.shape[0]
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    far, near = 1.5, 0.0
    max_steps = int(math.ceil((far - near) / step)) + 1
    g = torch.Generator().manual_seed(5)
    u = torch.rand(R, generator=g)
    t0 = (u * step)
    binaries = occ.reshape(64, 64, 64).bool()
    ri, ts, te, packed = O.occ_march(o.numpy(), d.numpy(), binaries.numpy(), aabb, near, far, step, t0.numpy(), max_steps, lattice=lattice)
    other = 'single' if lattice == 'repeated' else 'repeated'
    ri1, ts1, te1, _ = O.occ_march(o.numpy(), d.numpy(), binaries.numpy(), aabb, near, far, step, t0.numpy(), max_steps, lattice=other)
    assert ri.size > 1000 and (ts.size != ts1.size or not np.array_equal(ts, ts1))          # a different lattice indeed
    bits = ops.occ_pack_bits(occ.cuda())
    coarse = ops.occ_build_coarse(bits, 64)
    for origin in (t0.cuda(), (u.cuda(), step, near)):
        for cz in (None, coarse):
            gri, gts, gte, gpacked = ops.occ_march(o.cuda(), d.cuda(), origin, bits, 64, aabb, far, step, max_steps, occ_coarse=cz,
                                                   lattice=lattice)
            assert np.array_equal(gri.cpu().numpy(), ri) and np.array_equal(gpacked.cpu().numpy(), packed)
            assert np.array_equal(gts.cpu().numpy(), ts) and np.array_equal(gte.cpu().numpy(), te)
    if lattice == 'repeated':
        # per-ray tables of runs built one LANE per ray (perf_occ_lattice_runs) and handed to the marching kernels: same bits
        for origin in (t0.cuda(), (u.cuda(), step, near)):
            runs = ops.lattice_runs(origin, step, max_steps)
            assert runs.dtype == torch.int32 and int(runs.view(R, -1)[:, 0].min()) > 0                 # every table fits
            with_runs = origin + (runs,) if isinstance(origin, tuple) else (origin, 0.0, 0.0, runs)
            gri, gts, gte, gpacked = ops.occ_march(o.cuda(), d.cuda(), with_runs, bits, 64, aabb, far, step, max_steps, occ_coarse=coarse, lattice=lattice)
            assert np.array_equal(gri.cpu().numpy(), ri) and np.array_equal(gpacked.cpu().numpy(), packed)
            assert np.array_equal(gts.cpu().numpy(), ts) and np.array_equal(gte.cpu().numpy(), te)
    K = 4
    m2, c2, (ri2, ts2, te2, pk2, x2, s2) = ops.occ_march_count_head(o.cuda(), d.cuda(), t0.cuda(), bits, 64, list(aabb), far, step, max_steps,
                                                                    coarse, K, list(aabb), lattice=lattice)
    if lattice == 'repeated':
        runs = ops.lattice_runs(t0.cuda(), step, max_steps)
        m3, c3, head3 = ops.occ_march_count_head(o.cuda(), d.cuda(), (t0.cuda(), 0.0, 0.0, runs), bits, 64, list(aabb), far, step, max_steps,
                                                 coarse, K, list(aabb), lattice=lattice)
        assert torch.equal(c3, c2) and all(torch.equal(a, b) for a, b in zip(head3, (ri2, ts2, te2, pk2, x2, s2)))
    assert np.array_equal(c2.cpu().numpy(), packed[:, 1])
    for r in np.nonzero(packed[:, 1] > 0)[0][:50]:
        n = min(int(packed[r, 1]), K)
        assert np.array_equal(ts2[r * K:r * K + n].cpu().numpy(), ts[packed[r, 0]:packed[r, 0] + n])
        assert np.array_equal(te2[r * K:r * K + n].cpu().numpy(), te[packed[r, 0]:packed[r, 0] + n])


def test_renderer_with_either_lattice_sync_free_equals_synced():
    """NeRFOCCRenderer.lattice through the whole sampler: on either lattice the sync-free two-phase path (device-side counts) and
    the host-synced path produce the same samples and pixels; the default is 'repeated'."""
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    torch.manual_seed(0)
    scene = NeRFScene(dtype='fp16')
    rays = gen_pano_rays(torch.eye(4), 32, 64)
    dist, rgb = synthetic.room(rays.d)
    pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
    scene.set_train(); scene.prepare_occupancy(pool); scene.set_eval()
    assert scene.renderer.lattice == 'repeated'
    a = scene.render(rays, ['rgb', 'distance'], sync_free=True)
    b = scene.render(rays, ['rgb', 'distance'], sync_free=False)
    assert torch.equal(a['rgb'], b['rgb']) and torch.equal(a['distance'], b['distance'])
    scene.renderer.lattice = 'single'
    c = scene.render(rays, ['distance'], sync_free=True)
    c2 = scene.render(rays, ['distance'], sync_free=False)
    assert torch.equal(c['distance'], c2['distance'])
    assert not torch.equal(a['distance'], c['distance'])                         # (the lattices differ in the last bits of t)


def test_lagged_units_do_not_follow_a_vanishing_gradient(ops):
    """perf_dp_units with a margin (the lagged units of perf_amd/dp.py): the units of the next step may get coarser at once but
    only two bits finer per step -- max |dfeat| of a batch the field already fits vanishes (1e-21 observed), and units derived
    from THAT would overflow on the ordinary batch that follows (the job-wide gate then drops the step)."""
    cfg = _grid_cfg()
    hr = ops.headroom_state('cuda')
    n = 30000

    def stats(scale):
        am = torch.zeros(24, device='cuda'); am[:cfg.n_levels] = scale * torch.linspace(1.0, 2.0, cfg.n_levels, device='cuda')
        return ops.dp_stats_pack(am, None, None, n).view(1, -1)
    shifts, _ = ops.dp_units(cfg, stats(1e-5), 1, hr)                              # an exact call sets the units
    s0 = shifts.clone()
    ops.dp_units(cfg, stats(1e-5), 1, hr, shifts=shifts, want_total=False, margin_bits=1)
    s1 = shifts.clone()
    assert bool((s1[:cfg.n_levels] <= s0[:cfg.n_levels] + 2).all())
    ops.dp_units(cfg, stats(1e-21), 1, hr, shifts=shifts, want_total=False, margin_bits=1)     # the gradient vanishes ...
    s2 = shifts.clone()
    assert bool((s2[:cfg.n_levels] <= s1[:cfg.n_levels] + 2).all()) and bool((s2[:cfg.n_levels] >= s1[:cfg.n_levels]).all())
    ops.dp_units(cfg, stats(0.0), 1, hr, shifts=shifts, want_total=False, margin_bits=1)       # ... entirely
    s3 = shifts.clone()
    assert bool((s3[:cfg.n_levels] <= s2[:cfg.n_levels] + 2).all())
    ops.dp_units(cfg, stats(1e-2), 1, hr, shifts=shifts, want_total=False, margin_bits=1)      # a large gradient: coarser at once
    s4 = shifts.clone()
    assert bool((s4[:cfg.n_levels] < s1[:cfg.n_levels] - 6).all())


def test_lane_per_ray_marching_of_large_shared_lattice_launches_equals_wave_per_ray(ops):
    """Launches of >= 262,144 rays that share one lattice (eval frames) take march_count_shared_kernel (a ray per LANE for the set-up
    and the chunk decisions, then the wave walks its 64 rays); smaller ones the wave-per-ray kernel.  The same rays in one large launch
    and in two halves: keep masks, counts and the head rows the counting pass writes are identical, on both lattices."""
    from perf_amd import synthetic
    from perf_amd.scene import SupInfoPool, gen_pano_rays
    rays = gen_pano_rays(torch.eye(4), 384, 768)                     # 294,912 rays
    o = rays.o.reshape(-1, 3).contiguous() + torch.tensor([0.05, -0.1, 0.02], device='cuda'); d = rays.d.reshape(-1, 3).contiguous()
    dist, rgb = synthetic.pillars(d)
    pool = SupInfoPool(); pool.register_rays(rays.o.reshape(-1, 3), d, rgb.reshape(-1, 3), dist.reshape(-1, 1))
    occ, _ = pool.gen_occ_grid(256)
    bits = ops.occ_pack_bits(occ)
    coarse = ops.occ_build_coarse(bits, 256)
    aabb = [-1., -1, -1, 1, 1, 1]
    step, far, max_steps, K = 5e-4, 1.5, 3001, 2
    R = o.shape[0]
    assert R >= 262144 and R // 2 < 262144
    for lattice in ('repeated', 'single'):
        t0 = (None, 0.0, 0.0, ops.lattice_table(0.0, step, max_steps, lattice))
        for cz in (coarse, None):
            m_all, c_all, head_all = ops.occ_march_count_head(o, d, t0, bits, 256, aabb, far, step, max_steps, cz, K, aabb, lattice=lattice)
            m2, c2 = ops.occ_march_count(o, d, t0, bits, 256, aabb, far, step, max_steps, cz, lattice=lattice)
            assert torch.equal(c_all, c2)
            h = R // 2
            mw = m_all.numel() // R
            for lo, hi in ((0, h), (h, R)):
                m, c, head = ops.occ_march_count_head(o[lo:hi].contiguous(), d[lo:hi].contiguous(), t0, bits, 256, aabb, far, step, max_steps, cz, K, aabb,
                                                      lattice=lattice)
                assert torch.equal(c, c_all[lo:hi])
                # (mask words of chunks without samples are never written: compare the live words and the masks they announce)
                ma, mb = m_all.view(R, mw)[lo:hi], m.view(hi - lo, mw)
                assert torch.equal(ma[:, 0], mb[:, 0])
                for q in range(mw - 1):
                    on = ((ma[:, 0] >> q) & 1).bool()
                    assert torch.equal(ma[on, 1 + q], mb[on, 1 + q]), (lattice, q)
                ri, ts, te, pk, x01, sel = head
                ri_a, ts_a, te_a, pk_a, x_a, s_a = head_all
                assert torch.equal(ts, ts_a[lo * K:hi * K]) and torch.equal(te, te_a[lo * K:hi * K]) and torch.equal(sel, s_a[lo * K:hi * K])
                assert torch.equal(x01, x_a[lo * K:hi * K]) and torch.equal(pk[:, 1], pk_a[lo:hi, 1]) and torch.equal(ri + lo, ri_a[lo * K:hi * K])
            assert int(c_all.sum()) > 100000
            # the WRITE pass of such launches (march_write_shared_kernel: a ray per lane for counts / offsets / packed_info, the wave walks the
            # rays that have samples) against the halves (march_write_kernel): the plain expansion with positions, the two-phase sampler's
            # tail form (rank_lo = K, most counts zero) and a truncated capacity
            if cz is not None:
                for counts_w, rank_lo, cap_cut in ((c_all, 0, 0), (torch.where((c_all > K) & (torch.arange(R, device='cuda') % 3 == 0), c_all - K, torch.zeros_like(c_all)), K, 0),
                                                   (c_all, 0, 1000)):
                    offs, total = ops.exclusive_scan_i32(counts_w)
                    S = int(total.item()) - cap_cut
                    ri_a, ts_a, te_a, pk_a, x_a, s_a = ops.occ_march_write(t0, m_all, counts_w, offs, S, step, max_steps, o, d, aabb, rank_lo=rank_lo, lattice=lattice)
                    for lo, hi in ((0, h), (h, R)):
                        cw = counts_w[lo:hi].contiguous()
                        base = int(offs[lo].item())
                        end = min(int(offs[hi - 1].item()) + int(cw[-1].item()), S)
                        ow = (offs[lo:hi] - base).contiguous()
                        mh = m_all.view(R, mw)[lo:hi].contiguous().view(-1)
                        ri, ts, te, pk, x01, sel = ops.occ_march_write(t0, mh, cw, ow, max(end - base, 0), step, max_steps, o[lo:hi].contiguous(), d[lo:hi].contiguous(),
                                                                       aabb, rank_lo=rank_lo, lattice=lattice)
                        n_ = max(end - base, 0)
                        assert torch.equal(ts[:n_], ts_a[base:end]) and torch.equal(te[:n_], te_a[base:end]) and torch.equal(ri[:n_] + lo, ri_a[base:end]), (lattice, rank_lo, cap_cut)
                        assert torch.equal(x01[:n_], x_a[base:end]) and torch.equal(sel[:n_], s_a[base:end])
                        assert torch.equal(pk[:, 1], pk_a[lo:hi, 1]) and torch.equal(pk[:, 0] + base, pk_a[lo:hi, 0])
                    assert int(pk_a[:, 1].sum()) == S if cap_cut == 0 else int(pk_a[:, 1].sum()) == S


@pytest.mark.parametrize('maxc', [200, 17, 5])
@pytest.mark.parametrize('tail_empty,data_parallel', [(False, False), (True, False), (False, True)])
def test_train_head_in_one_launch_equals_the_three_launch_chains(ops, tail_empty, data_parallel, maxc):
    """perf_train_head_geo / _app (compositing forward -> loss head -> compositing backward, ONE launch; a wavefront per ray, or -- for the
    few samples per ray a training batch keeps late in an episode -- 16- and 4-lane ray teams) against the chains they replace, on rays of
    0 / 1 / 64 / 65 / up to 200 samples (maxc = 200: wavefronts) and on rays of up to 16 / 4 samples around those (quarter waves, 4-lane
    teams, and the groups that hold a longer ray): every per-sample and per-ray output and the gradient the field backward starts from bit
    for bit; the loss values (now summed by the reader from per-ray terms) to summation order.  tail_empty: the last rays hold no sample
    (flatten_eff_distloss's normaliser is the last ray that does)."""
    packed, ri, ts, te, sig, rgb = _packed_case(21, R=300 if maxc == 200 else 1200, maxc=maxc)
    R = packed.shape[0]
    if tail_empty:          # rays R-70.. lose their samples: the normaliser sits more than one ballot back from the end
        keep = R - 70
        S = int(packed[keep, 0])
        packed = packed.clone(); packed[keep:, 1] = 0; packed[keep:, 0] = S
        ts, te, sig, rgb = ts[:S], te[:S], sig[:S], rgb[:S]
    c = lambda t: t.cuda().contiguous()
    g = torch.Generator().manual_seed(22)
    gt_d = torch.rand(R, generator=g) * 1.5; noise = torch.rand(R, generator=g)
    gt_c = torch.rand(R, 3, generator=g); bg = torch.rand(R, 3, generator=g)
    ratio = torch.tensor([0.6], device='cuda')
    bs = 4 * R if data_parallel else R
    # ---- geometry
    w, T, op, dist, col, dl = ops.composite_distloss_fwd(c(sig), c(rgb), c(ts), c(te), c(packed))
    g_op, g_d, sc = ops.geo_loss(op, dist, c(gt_d), c(noise), dl, c(packed), bs, 1.0, 0.5, ratio, 128.0)
    ds = ops.composite_distloss_bwd(c(sig), c(ts), c(te), c(packed), w, T, op, dist, g_op, g_d, 1.0, scale_dev=sc[2:3])
    hd = ops.train_head_geo(c(sig), c(rgb), c(ts), c(te), c(packed), c(gt_d), c(noise), bs, 1.0, 0.5, ratio, 128.0)
    for a, b, name in ((w, hd['weights'], 'w'), (T, hd['trans'], 'T'), (op, hd['opacity'], 'op'), (dist, hd['distance'], 'dist'),
                       (col, hd['color'], 'col'), (dl, hd['distloss_per_ray'], 'dl'), (ds, hd['d_sigma'], 'd_sigma')):
        assert torch.equal(a, b), name
    depth = hd['depth_terms'].sum() / bs
    dloss = hd['distloss_per_ray'].sum() * hd['inv_n'][0]
    assert abs(float(depth) - float(sc[0])) <= 1e-5 * abs(float(sc[0]))
    assert abs(float(dloss) - float(sc[1])) <= 1e-5 * abs(float(sc[1]))
    last = int(torch.nonzero(packed[:, 1] > 0).max())
    assert float(hd['inv_n'][0]) == (np.float32(1.0) / np.float32(bs if data_parallel else last + 1))
    # without noise / ramp / colour
    hd0 = ops.train_head_geo(c(sig), None, c(ts), c(te), c(packed), c(gt_d), None, bs, 1.0, 0.5, None, 128.0)
    g_op0, g_d0, sc0 = ops.geo_loss(op, dist, c(gt_d), None, dl, c(packed), bs, 1.0, 0.5, None, 128.0)
    ds0 = ops.composite_distloss_bwd(c(sig), c(ts), c(te), c(packed), w, T, op, dist, g_op0, g_d0, 1.0, scale_dev=sc0[2:3])
    assert torch.equal(ds0, hd0['d_sigma']) and hd0['color'] is None
    # ---- colour
    w, T, _, op, dist, col = ops.composite_fwd(c(sig), c(rgb), c(ts), c(te), c(packed))
    for bgc in (c(bg), None):
        g_col, sca = ops.app_loss(op, col, bgc, c(gt_c), bs, 1.0, 128.0)
        _, drgb = ops.composite_bwd(c(sig), c(ts), c(te), c(packed), w, T, g_color=g_col, want_dsigma=False, want_drgb=True)
        ha = ops.train_head_app(c(sig), c(rgb), c(ts), c(te), c(packed), bgc, c(gt_c), bs, 1.0, 128.0)
        for a, b, name in ((w, ha['weights'], 'w'), (T, ha['trans'], 'T'), (op, ha['opacity'], 'op'), (dist, ha['distance'], 'dist'),
                           (col, ha['color'], 'col')):
            assert torch.equal(a, b), name
        live = int(packed[:, 1].sum())
        assert torch.equal(drgb[:live], ha['d_rgb'][:live])
        closs = ha['color_terms'].sum() / (3 * bs)
        assert abs(float(closs) - float(sca[0])) <= 1e-5 * abs(float(sca[0]))


def test_train_head_on_a_batch_without_samples(ops):
    """A capacity-sized batch whose rays all came back empty (a transparent field, or everything pruned): the one-launch ray head
    equals the chains on the per-ray outputs, the distortion normaliser falls back to 1 like perf_geo_loss's, no per-sample row is
    written, and a ray count that is not a multiple of four is served."""
    R, cap = 37, 256
    c = lambda t: t.cuda().contiguous()
    packed = torch.zeros(R, 2, dtype=torch.int32)
    g = torch.Generator().manual_seed(3)
    sig, ts, rgb = torch.rand(cap, generator=g), torch.rand(cap, generator=g), torch.rand(cap, 3, generator=g)
    te = ts + 1e-3
    gt_d, noise, gt_c, bg = torch.rand(R, generator=g), torch.rand(R, generator=g), torch.rand(R, 3, generator=g), torch.rand(R, 3, generator=g)
    w, T, op, dist, col, dl = ops.composite_distloss_fwd(c(sig), c(rgb), c(ts), c(te), c(packed))
    g_op, g_d, sc = ops.geo_loss(op, dist, c(gt_d), c(noise), dl, c(packed), R, 1.0, 0.5, None, 128.0)
    hd = ops.train_head_geo(c(sig), c(rgb), c(ts), c(te), c(packed), c(gt_d), c(noise), R, 1.0, 0.5, None, 128.0)
    hd['d_sigma'].fill_(7.0); hd2 = ops.train_head_geo(c(sig), c(rgb), c(ts), c(te), c(packed), c(gt_d), c(noise), R, 1.0, 0.5, None, 128.0)
    for a, b in ((op, hd['opacity']), (dist, hd['distance']), (col, hd['color']), (dl, hd['distloss_per_ray'])):
        assert torch.equal(a, b) and float(a.abs().max()) == 0.0
    assert float(hd['inv_n'][0]) == 1.0
    assert abs(float(hd['depth_terms'].sum() / R) - float(sc[0])) <= 1e-6 * abs(float(sc[0])) and float(sc[1]) == 0.0
    assert torch.equal(hd['depth_terms'], hd2['depth_terms'])
    ha = ops.train_head_app(c(sig), c(rgb), c(ts), c(te), c(packed), c(bg), c(gt_c), R, 1.0, 128.0)
    _, sca = ops.app_loss(op, col, c(bg), c(gt_c), R, 1.0, 128.0)
    assert abs(float(ha['color_terms'].sum() / (3 * R)) - float(sca[0])) <= 1e-6 * abs(float(sca[0]))


@pytest.mark.parametrize('fixed', [True, False])
@pytest.mark.parametrize('nh,n_out,act', [(1, 1, 'Exponential'), (2, 3, 'Sigmoid')])
def test_field_backward_in_one_boundary_call_equals_the_three_calls(ops, fixed, nh, n_out, act):
    """perf_field_bwd (MLP backward -> grid backward -> predicated repair launch, one workspace) against the three entry points
    called one by one: the same flat gradient (bit for bit with the order-independent fixed-point fields), with plain features, with an IndexedFeat, with a device-side live count,
    and on a second call that reuses the cached workspace; the headroom feedback state ends in the same place."""
    cfg = _grid_cfg()
    cfgm, w, feat, sel, tdt, _ = _mlp_case(ops, 'bf16', nh, n_out, act, n=20011, seed=5)
    g = torch.Generator().manual_seed(11)
    n = feat.shape[1]
    x = torch.rand(n, 3, generator=g).cuda()
    dout = (torch.randn(n, n_out, generator=g) * 1e-2).cuda()
    w16, f16, selc = w.to(tdt).cuda(), feat.to(tdt).cuda(), sel.cuda()
    n_net = cfgm.n_params

    def three(feat_arg, n_dev, hr):
        grad = torch.empty(n_net + cfg.n_params, dtype=torch.float32, device='cuda')
        res = ops.mlp_bwd(cfgm, w16, feat_arg, dout, selc, want_absmax=fixed, n_dev=n_dev, dw_out=grad[:n_net])
        ops.hashgrid_bwd_into(cfg, x, res[0], grad[n_net:], level_absmax=res[2] if fixed else None, n_dev=n_dev, hr_state=hr if fixed else None)
        if fixed:
            ops.hashgrid_bwd_redo(cfg, x, res[0], grad[n_net:], n_dev=n_dev, hr_state=hr)
        return grad

    idx = torch.randperm(n, generator=g).to(torch.int32).cuda()
    src = torch.empty_like(f16); src[:, idx.long()] = f16                       # features stored in another order + the row of every sample
    for feat_arg, n_dev in ((f16, None), (ops.IndexedFeat(src, idx), None), (f16, torch.tensor([12345], dtype=torch.int64, device='cuda'))):
        hr_a, hr_b = ops.headroom_state('cuda'), ops.headroom_state('cuda')
        for rep in range(2):                                                     # (second pass: cached workspace, evolved headroom state)
            ops.overflow_flag('cuda').zero_()
            a = three(feat_arg, n_dev, hr_a)
            ops.overflow_flag('cuda').zero_()
            b = ops.field_bwd(cfg, cfgm, x, w16, feat_arg, dout, selc, fixed=fixed, redo=True, hr_state=hr_b if fixed else None, n_dev=n_dev)
            if fixed:
                assert torch.equal(a, b), (nh, rep, float((a - b).abs().max()))
                assert torch.equal(hr_a, hr_b)
            else:       # fp32 LDS atomics add in arrival order: two runs of the SAME call differ in the last bits
                assert torch.equal(a[:n_net], b[:n_net]) and float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()), (nh, rep)
    assert float(b.abs().max()) > 0
