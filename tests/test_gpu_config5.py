"""BASELINE config 5 at its stated size on ONE GPU (4096x2048 panorama, 256 samples per ray, L = 20 hash grids whose 16-bit
tables no cache holds: T = 2^28 -> 9.2 GiB per encoder, parameter offsets beyond 32 bit; T = 2^30 -> 31 GiB per encoder, ENTRY
offsets beyond 32 bit) -- the same workload bench.py's `config5` block times (perf_amd/panorama.py), with tcnn's table layout and
with the opt-in line-local one (perf_amd.grid.GridConfig; values against the oracle: tests/test_gpu_ops.py), checked through
size-independent properties: packed bookkeeping (sortedness, counts), compositing bounds, determinism, and equality of a
row shard rendered on its own with the same rows of the full render (the multi-GPU eval partitioning of SURVEY.md 8(e):
rank r of G renders rows [r H/G, (r+1) H/G), no communication).  The fields are NOT at their fresh initialisation here
(tables U(-4, 4), densities exp(y + 4)): densities vary, rays terminate early, the compaction is exercised."""
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, SPP = 2048, 4096, 256
TABLE_SCALE, DENSITY_BIAS = 4.0, 4.0      # tables U(-4, 4), sigma = exp(y + 4): y ~ N(0, 1), optical depth ~ 80 per ray -- rays end early


def _field(log2_t, layout='tcnn'):
    from perf_amd.fields import InferenceNeRF
    from perf_amd import panorama as C
    nerf = InferenceNeRF([-1., -1, -1, 1, 1, 1], n_levels=20, log2_hashmap_size=log2_t, per_level_scale=C.per_level_scale(20),
                         dtype='fp16', table_scale=TABLE_SCALE, density_bias=DENSITY_BIAS, layout=layout)
    est, rend = C.make_renderer(SPP)
    return nerf, est, rend


def _check_batch(res, R):
    """Packed bookkeeping of one batch (capacity-sized arrays; the first n rows are live)."""
    n = int(res['n_samples_dev'].item()); m = int(res['n_marched_dev'].item())
    assert m == R * SPP and 0 < n <= m
    ri, ts, te, packed = res['ray_indices'][:n], res['t_starts'][:n], res['t_ends'][:n], res['packed_info']
    assert bool((ri[1:] >= ri[:-1]).all())                                            # sorted by ray ...
    same = ri[1:] == ri[:-1]
    assert bool((ts[1:][same] > ts[:-1][same]).all())                                 # ... then by t
    assert bool((te > ts).all()) and float((te - ts).max()) < 1.01 * 0.99 / SPP
    assert int(packed[:, 1].sum()) == n and int(packed[:, 1].max()) <= SPP
    assert torch.equal(packed[:, 0].long(), torch.cumsum(packed[:, 1].long(), 0) - packed[:, 1].long())
    assert int(ri.min()) >= 0 and int(ri.max()) < R
    w = res['weights'][:n]
    assert bool(torch.isfinite(w).all()) and float(w.min()) >= 0.0
    return n


@pytest.mark.parametrize('layout', ['tcnn', 'line_local', 'line_overlap'])
def test_config5_full_panorama_properties(layout):
    """T = 2^28, the whole 4096x2048x256 panorama (2.1e9 marched ray-samples)."""
    from perf_amd import ops
    from perf_amd import panorama as C
    nerf, est, rend = _field(28, layout)
    assert nerf.grid.layout == layout and (layout == 'tcnn') == (int(nerf.grid.local.sum()) == 0)
    assert nerf.grid.n_levels == 20 and nerf.grid.n_params >= 2 ** 32            # parameter offsets do not fit 32 bits
    assert int(nerf.grid.res[-1]) in (8192, 8193)
    counters = ops.step_counters('cuda')
    seen = []

    def look(res, lo, R):
        if lo // (4 * W) % 64 == 3:                                              # every 64th batch: the packed arrays themselves
            seen.append(_check_batch(res, R))
    outs = C.render_rows(nerf, est, rend, 0, H, 4, SPP, H, W, counters=counters, bookkeeping=look)
    c = counters.tolist()
    assert c[0] == H * W * SPP and 0 < c[1] < c[0]                               # everything marched, early termination pruned some
    assert len(seen) == 8
    for k, v in outs.items():
        assert bool(torch.isfinite(v).all()), k
    op = outs['opacities']
    assert float(op.min()) >= 0.0 and float(op.max()) <= 1.0 + 1e-5
    assert float(outs['rgb'].min()) >= 0.0 and float(outs['rgb'].max()) <= 1.0 + 1e-5
    assert float(outs['distance'].min()) >= 0.0 and float(outs['distance'].max()) <= 0.99 + 5.0 + 1e-4
    assert float(outs['rgb'].std()) > 1e-3 and float(outs['distance'].std()) > 1e-4     # the fields are not constant (every ray ends opaque)
    # row sharding: rank 5 of 8 renders its 256 rows on its own, in batches of another size
    r0, nr = 5 * H // 8, H // 8
    shard = C.render_rows(nerf, est, rend, r0, nr, 2, SPP, H, W)
    for k in outs:
        assert torch.equal(shard[k], outs[k][r0 * W:(r0 + nr) * W]), k
    # batches as 2-D tiles of pixels (what bench.py's config5 block renders: neighbouring rays in both image directions share a
    # launch) give the same pixels
    tiled = C.render_rows(nerf, est, rend, r0, 128, 4, SPP, H, W, tile=(128, 128))
    for k in outs:
        assert torch.equal(tiled[k], outs[k][r0 * W:(r0 + 128) * W]), k
    # determinism: the same rows again, bit for bit
    again = C.render_rows(nerf, est, rend, r0, 16, 4, SPP, H, W)
    for k in outs:
        assert torch.equal(again[k], outs[k][r0 * W:(r0 + 16) * W]), k


@pytest.mark.parametrize('layout', ['tcnn', 'line_local', 'line_overlap'])
def test_config5_tables_beyond_32_bit_entry_offsets(layout):
    """T = 2^30 (31 GiB per encoder, 8.4e9 entries: 64-bit level offsets in entries, not only in parameters): a band of rows
    around the equator and one at the pole, batch-size independence, bounds."""
    from perf_amd import panorama as C
    nerf, est, rend = _field(30, layout)
    assert nerf.grid.total >= 2 ** 32
    for r0 in (0, H // 2 - 8):
        a = C.render_rows(nerf, est, rend, r0, 16, 4, SPP, H, W)
        b = C.render_rows(nerf, est, rend, r0, 16, 1, SPP, H, W)
        for k in a:
            assert torch.equal(a[k], b[k]), (r0, k)
            assert bool(torch.isfinite(a[k]).all())
        assert float(a['opacities'].min()) >= 0.0 and float(a['opacities'].max()) <= 1.0 + 1e-5
        assert float(a['rgb'].std()) > 1e-3
