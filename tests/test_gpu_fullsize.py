"""Full-size (BASELINE.json config 2: 2048x1024 panorama, 32,768-ray eval batches, step 5e-4 / far 1.5) checks
through size-independent properties: sortedness and consistency of the packed bookkeeping, determinism, row-shard
equality (the multi-GPU eval partitioning), linearity of the encoding in its table, compositing bounds."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def room_scene():
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    torch.manual_seed(0)
    scene = NeRFScene(dtype='bf16')
    rays = gen_pano_rays(torch.eye(4), 1024, 2048)
    dist, rgb = synthetic.room(rays.d)
    pool = SupInfoPool()
    pool.register_rays(rays.o, rays.d, rgb, dist)
    assert len(pool) == 2097152
    scene.set_train()
    scene.prepare_occupancy(pool)
    with torch.no_grad():                                # a non-trivial density so that early termination is exercised
        n_net = scene.nerf.geo_mlp.mlp.n_params
        scene.nerf.geo_mlp.params[n_net:] *= 2e4
    return scene, pool, rays, dist, rgb


def test_full_batch_march_bookkeeping(room_scene):
    from perf_amd import ops
    scene, pool, rays, dist, rgb = room_scene
    est = scene.estimator
    o = rays.o.reshape(-1, 3)[:32768].contiguous(); d = rays.d.reshape(-1, 3)[1000000:1000000 + 32768].contiguous()
    R = 32768
    t0 = torch.rand(R, device='cuda') * 5e-4
    max_steps = int(math.ceil(1.5 / 5e-4)) + 1
    ri, ts, te, packed = ops.occ_march(o, d, t0, est.occ_bits(), 256, est._aabb_host, 1.5, 5e-4, max_steps)
    ri2, ts2, te2, packed2 = ops.occ_march(o, d, t0, est.occ_bits(), 256, est._aabb_host, 1.5, 5e-4, max_steps,
                                           occ_coarse=est.occ_coarse())
    assert torch.equal(ri, ri2) and torch.equal(ts, ts2) and torch.equal(te, te2) and torch.equal(packed, packed2)
    S = ri.numel()
    assert S > 100000
    assert bool((ri[1:] >= ri[:-1]).all())                                   # sorted by ray
    same = ri[1:] == ri[:-1]
    assert bool((ts[1:][same] > ts[:-1][same]).all())                        # then by t
    assert bool(((te - ts) > 0).all()) and float((te - ts).max()) < 5.01e-4
    assert int(packed[:, 1].sum()) == S
    assert torch.equal(packed[:, 0].long(), torch.cumsum(packed[:, 1].long(), 0) - packed[:, 1].long())
    assert torch.equal(ops.pack_info(ri, R), packed)
    # samples lie in occupied cells (checked on the device against the boolean grid)
    mid = (ts + te) * 0.5
    p = o[ri] + d[ri] * mid[:, None]
    cell = ((p + 1.0) * 0.5 * 256).floor().clamp(0, 255).long()
    assert bool(est.binaries[0][cell[:, 0], cell[:, 1], cell[:, 2]].all())


def test_full_panorama_render_properties(room_scene):
    from perf_amd.scene import Rays
    scene, pool, rays, dist, rgb = room_scene
    out = scene.render(rays, ['rgb', 'distance', 'opacities'])
    assert out['rgb'].shape == (1024, 2048, 3) and out['distance'].shape == (1024, 2048, 1)
    for k in ('rgb', 'distance', 'opacities'):
        assert bool(torch.isfinite(out[k]).all())
    assert float(out['opacities'].min()) >= 0.0 and float(out['opacities'].max()) <= 1.0 + 1e-5
    # determinism of the eval path (no stratification, no atomics in compositing): a second render is identical
    out2 = scene.render(rays, ['rgb', 'distance'])
    assert torch.equal(out['rgb'], out2['rgb']) and torch.equal(out['distance'], out2['distance'])
    # row sharding (multi-GPU eval): a shard rendered on its own equals the same rows of the full render
    shard = Rays(rays.o[256:384], rays.d[256:384])
    outs = scene.render(shard, ['rgb', 'distance'])
    assert torch.equal(outs['rgb'], out['rgb'][256:384]) and torch.equal(outs['distance'], out['distance'][256:384])


def test_encoding_is_linear_in_the_table():
    from perf_amd import ops
    from perf_amd.grid import GridConfig
    cfg = GridConfig()
    g = torch.Generator(device='cuda').manual_seed(3)
    n = 1 << 22
    x = torch.rand(n, 3, device='cuda', generator=g)
    t1 = torch.randn(cfg.n_params, device='cuda', generator=g)
    t2 = torch.randn(cfg.n_params, device='cuda', generator=g)
    f1 = ops.hashgrid_fwd_f32(cfg, x, t1); f2 = ops.hashgrid_fwd_f32(cfg, x, t2)
    f12 = ops.hashgrid_fwd_f32(cfg, x, 2.0 * t1 - 0.5 * t2)
    assert float((f12 - (2.0 * f1 - 0.5 * f2)).abs().max()) < 2e-5
    # and the backward is its adjoint: <enc(x; T), G> == <T, bwd(x; G)>
    G = torch.randn(cfg.n_levels, n, 2, device='cuda', generator=g)
    lhs = float((f1.double() * G.double()).sum())
    rhs = float((t1.double() * ops.hashgrid_bwd(cfg, x, G).double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_training_batch_grid_gradient_properties():
    """The bench's batch (8,192 rays x 128 samples, ray-coherent) through the default fixed-point owners: every level
    conserves mass (trilinear weights sum to 1: sum of the level's gradient entries == sum of its dfeat), the backward is
    the adjoint of the forward, two calls are bit-identical, and the overflow flag stays clear."""
    from perf_amd import ops
    from perf_amd.grid import GridConfig
    cfg = GridConfig()
    g = torch.Generator(device='cuda').manual_seed(4)
    R, S = 8192, 128
    d = torch.nn.functional.normalize(torch.randn(R, 3, device='cuda', generator=g), dim=-1)
    t = (torch.arange(S, device='cuda') + torch.rand(R, 1, device='cuda', generator=g)) * (0.99 / S)
    x = ((d[:, None, :] * t[:, :, None]).reshape(-1, 3) * 0.5 + 0.5).contiguous()
    n = x.shape[0]
    G = torch.randn(cfg.n_levels, n, 2, device='cuda', generator=g) * torch.logspace(-4, 0, cfg.n_levels, device='cuda')[:, None, None]
    amax = torch.zeros(24, device='cuda'); amax[:cfg.n_levels] = G.abs().amax(dim=(1, 2))
    ops.overflow_flag(x.device).zero_()
    g1 = ops.hashgrid_bwd(cfg, x, G, level_absmax=amax)
    g2 = ops.hashgrid_bwd(cfg, x, G, level_absmax=amax)
    assert int(ops.overflow_flag(x.device).item()) == 0
    assert torch.equal(g1, g2)
    for l in range(cfg.n_levels):
        lo, hi = 2 * int(cfg.offset[l]), 2 * (int(cfg.offset[l]) + int(cfg.size[l]))
        got = g1[lo:hi].view(-1, 2).double().sum(0)
        ref = G[l].double().sum(0)
        scale = float(G[l].abs().double().sum())
        assert float((got - ref).abs().max()) < 2e-4 * scale / math.sqrt(n) + 1e-3 * float(amax[l]), l
    table = torch.randn(cfg.n_params, device='cuda', generator=g)
    f = ops.hashgrid_fwd_f32(cfg, x, table)
    lhs = float((f.double() * G.double()).sum())
    rhs = float((table.double() * g1.double()).sum())
    assert abs(lhs - rhs) < 2e-4 * max(1.0, float((f.double() * G.double()).abs().sum()) / math.sqrt(n))
