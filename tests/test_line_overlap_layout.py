"""CPU: the opt-in overlapping-run table layout (perf_amd.grid.GridConfig(layout='line_overlap'), oracle/perf_oracle.py:grid_levels /
grid_corner_indices / canonical_overlap_fill) -- line_local whose 16-byte x runs overlap by one vertex.  The host-side
level table equals the oracle's; the x corner pair of every cell lies in ONE run, except for the last cell of a super-block row, whose
second corner is the next super-block's first vertex; a vertex has at most two entries, both in the same super-block row, and after the
canonical fill both hold one value (so the table is ONE field: every vertex reads the same value from every cell that touches it);
the eight corners of a cell lie in (1 + 1/4)(1 + 1/2) = 1.9 lines on average.  (Values against the oracle on the GPU:
tests/test_gpu_ops.py::test_deep_grid_forward_both_layouts, ::test_overlapping_runs_hold_one_field.)"""
import numpy as np
import pytest
import torch

from oracle import perf_oracle as O
from perf_amd.grid import GridConfig

L, B = 20, 1.3819


def _levels(log2_t, sb_shift, min_res, n_levels=L):
    cfg = GridConfig(n_levels=n_levels, log2_hashmap_size=log2_t, base_resolution=16, per_level_scale=B, layout='line_overlap', sb_shift=sb_shift,
                     local_min_res=min_res)
    lv = O.grid_levels(n_levels, 2, log2_t, 16, B, layout='line_overlap', sb_shift=sb_shift, local_min_res=min_res)
    return cfg, lv


@pytest.mark.parametrize('log2_t,sb_shift,min_res', [(15, (2, 2, 1), 16), (20, (3, 3, 2), 64), (24, (5, 6, 8), 64), (28, (7, 5, 7), 64)])
def test_host_level_table_equals_the_oracle(log2_t, sb_shift, min_res):
    cfg, lv = _levels(log2_t, sb_shift, min_res)
    assert cfg.total == lv.total
    for name in ('res', 'size', 'offset', 'hashed', 'local', 'nsx', 'nsxy'):
        assert np.array_equal(np.asarray(getattr(cfg, name)).astype(np.int64), np.asarray(getattr(lv, name)).astype(np.int64)), name
    assert np.array_equal(np.asarray(cfg.scale, np.float32), lv.scale)
    per_sb = 1 << sum(sb_shift)
    plain = O.grid_levels(L, 2, log2_t, 16, B, layout='line_local', sb_shift=sb_shift, local_min_res=min_res)
    for l in range(L):
        assert bool(lv.local[l]) == bool(plain.local[l])
        if lv.local[l]:
            assert int(lv.size[l]) % per_sb == 0 and int(lv.offset[l]) % per_sb == 0
            # a super-block row holds 3/4 as many cells: a dense level needs up to 4/3 (+ a super-block) as many entries, a hashed one the same 2^T
            assert int(plain.size[l]) <= int(lv.size[l]) <= max(int(plain.size[l]) * 4 // 3 + int(plain.nsxy[l] // max(int(plain.nsx[l]), 1)) * int(lv.nsxy[l] // max(int(lv.nsx[l]), 1)) * per_sb, 1 << log2_t)
            if lv.hashed[l]:
                assert int(lv.size[l]) == 1 << log2_t


def _cells(lv, l, g):
    """corner indices [N, 8] of the integer cells g [N, 3] of level l (the oracle's own function, fed each cell's centre)."""
    x = ((g.astype(np.float64)) / float(lv.scale[l])).astype(np.float32)
    idx, _ = O.grid_corner_indices(x, lv, l)
    assert np.array_equal(np.floor(O.grid_pos(x, lv.scale[l])).astype(np.int64), g)
    return idx.astype(np.int64)


@pytest.mark.parametrize('log2_t,sb_shift', [(20, (3, 3, 2)), (22, (5, 6, 8))])
def test_a_dense_level_is_one_field_with_every_x_pair_in_one_run(log2_t, sb_shift):
    cfg, lv = _levels(log2_t, sb_shift, 64, n_levels=8)                   # (eight levels up to resolution 160: a table of a few million entries)
    l = int(np.argmax(lv.local & ~lv.hashed))
    r = int(lv.res[l])
    gx = np.arange(0, r - 1)
    gyz = np.array([0, 3, 4, 7, r // 2, r - 2])
    g = np.stack(np.meshgrid(gx, gyz, gyz, indexing='ij'), -1).reshape(-1, 3)
    idx = _cells(lv, l, g)
    assert int(idx.max()) < int(lv.size[l])
    cells_per_row = 3 << (sb_shift[0] - 2)
    edge = (g[:, 0] % cells_per_row) == cells_per_row - 1
    assert edge.sum() > 0
    for yz in range(4):                                                  # each (y, z) corner pair: entries 2 yz (first x corner) and 2 yz + 1
        a, b = idx[:, 2 * yz], idx[:, 2 * yz + 1]
        assert np.array_equal(b[~edge], a[~edge] + 1)                    # neighbours inside one run ...
        assert ((a[~edge] & 3) == (g[~edge, 0] % 3)).all()               # ... at positions gx % 3, gx % 3 + 1 of run gx // 3
        assert ((a[edge] & 3) == 2).all() and ((b[edge] & 3) == 0).all()          # the row's last cell: second corner = first vertex ...
        assert (b[edge] // (1 << sum(sb_shift)) != a[edge] // (1 << sum(sb_shift))).all()      # ... of ANOTHER super-block
    # a vertex has at most two entries; the second one is position 3 of the run in front of its canonical run, same super-block row
    ent = {}
    for c in range(8):
        v = g + np.array([c & 1, (c >> 1) & 1, c >> 2])
        for p, i in zip(map(tuple, v), idx[:, c]):
            ent.setdefault(p, set()).add(int(i))
    assert max(len(s) for s in ent.values()) == 2
    twice = [sorted(s) for s in ent.values() if len(s) == 2]
    assert len(twice) > 100
    for lo, hi in twice:
        assert hi - lo == 32 - 3 and lo % 4 == 3 and hi % 4 == 0         # [run j, position 3] and [run j + 1, position 0]: the next block along x
    # after the canonical fill every vertex reads ONE value whichever cell asks
    rng = np.random.default_rng(3)
    table = rng.standard_normal((lv.total, 2)).astype(np.float32)
    canon = O.canonical_overlap_fill(table, lv)
    off = int(lv.offset[l])
    for s in ent.values():
        vals = {tuple(canon[off + i]) for i in s}
        assert len(vals) == 1
    assert any(tuple(table[off + lo]) != tuple(table[off + hi]) for lo, hi in twice[:10])          # (the raw table was not one)
    # the product's in-place fill is the oracle's
    t = torch.from_numpy(table.copy())
    cfg.canonicalize_(t)
    assert np.array_equal(t.numpy(), canon)
    # entries the fill does not own are untouched; it is idempotent
    assert np.array_equal(O.canonical_overlap_fill(canon, lv), canon)
    changed = np.flatnonzero((canon != table).any(-1))
    assert changed.size > 0 and (changed % 4 == 3).all() and int(changed.min()) >= int(lv.offset[int(np.argmax(lv.local))])      # (line-local levels start on multiples of 32 entries)


@pytest.mark.parametrize('log2_t,sb_shift', [(15, (2, 2, 1)), (20, (3, 3, 2)), (24, (5, 6, 8)), (24, (7, 5, 7))])
def test_a_hashed_level_keeps_super_blocks_together_and_a_cell_in_fewer_lines(log2_t, sb_shift):
    cfg, lv = _levels(log2_t, sb_shift, 16 if log2_t == 15 else 64)
    l = L - 1
    assert lv.local[l] and lv.hashed[l]
    rng = np.random.default_rng(5)
    x = rng.random((20000, 3), dtype=np.float32) * 0.999
    idx, _ = O.grid_corner_indices(x, lv, l)
    idx = idx.astype(np.int64)
    assert int(idx.max()) < int(lv.size[l])
    per_sb = 1 << sum(sb_shift)
    g = np.floor(O.grid_pos(x, lv.scale[l])).astype(np.int64)
    cells_per_row = 3 << (sb_shift[0] - 2)
    sb = (g[:, 0] // cells_per_row, g[:, 1] >> sb_shift[1], g[:, 2] >> sb_shift[2])
    key = sb[0] + (sb[1] << 20) + (sb[2] << 40)
    slot = idx[:, 0] // per_sb
    first = {}
    for k, s in zip(key.tolist(), slot.tolist()):
        assert first.setdefault(k, s) == s                               # cells of one super-block share its slot
    edge = (g[:, 0] % cells_per_row) == cells_per_row - 1
    for yz in range(4):
        assert np.array_equal(idx[~edge, 2 * yz + 1], idx[~edge, 2 * yz] + 1)
    lines = np.array([np.unique(row >> 5).size for row in idx])
    plain = (1 + 1 / 4) * (1 + 1 / 2)
    if sb_shift[0] >= 5:
        assert plain < lines.mean() < plain + 0.15 and lines.max() <= 8           # 1.9 + the rows' last cells (1 in 24)
    inside = ((g[:, 1] & 3) < 3) & ((g[:, 2] & 1) < 1) & ~edge
    assert inside.sum() > 1000 and (lines[inside] == 1).all()


def test_the_oracles_encode_of_a_canonical_table_is_continuous_across_every_cell_face():
    """oracle/perf_oracle.py:hashgrid_encode over a canonically filled line_overlap table: points a hair left and right of cell faces along
    x -- run boundaries (every third face), super-block boundaries, dense and hashed levels -- encode to the same features up to the step
    across the face; the raw random table jumps at the run boundaries.  (The GPU twin: tests/test_gpu_ops.py::test_overlapping_runs_hold_one_field.)"""
    n_levels = 10
    cfg, lv = _levels(20, (3, 3, 2), 64, n_levels=n_levels)
    assert (lv.local & ~lv.hashed).any() and (lv.local & lv.hashed).any()
    rng = np.random.default_rng(11)
    raw = (rng.random((lv.total, 2), dtype=np.float32) * 2 - 1)
    canon = O.canonical_overlap_fill(raw, lv)
    eps = 2e-3
    for l in np.flatnonzero(lv.local):
        sc, r = float(lv.scale[l]), int(lv.res[l])
        n = 1500
        v = np.stack([rng.integers(1, r - 1, n), rng.integers(0, r - 1, n), rng.integers(0, r - 1, n)], -1).astype(np.float64)
        yz = rng.random((n, 2)) * 0.8 + 0.1
        def pts(dx):
            p = v.copy(); p[:, 0] += dx; p[:, 1:] += yz
            return np.clip((p - 0.5) / sc, 0.0, 1.0).astype(np.float32)
        xl, xr = pts(-eps), pts(+eps)
        gl, gr = np.floor(O.grid_pos(xl, lv.scale[l]))[:, 0], np.floor(O.grid_pos(xr, lv.scale[l]))[:, 0]
        ok = (gr == gl + 1) & (gr == v[:, 0])
        assert ok.sum() > 0.9 * n
        for table, one_field in ((canon, True), (raw, False)):
            fl = O.hashgrid_encode(torch.from_numpy(xl), torch.from_numpy(table), lv).numpy()[:, 2 * l:2 * l + 2]
            fr = O.hashgrid_encode(torch.from_numpy(xr), torch.from_numpy(table), lv).numpy()[:, 2 * l:2 * l + 2]
            jump = np.abs(fl - fr).max(-1)[ok]
            if one_field:
                assert jump.max() < 2 * eps * 2 * 2 + 1e-5, (l, jump.max())
            else:
                third = (v[:, 0][ok] % 3 == 0)
                assert jump[third].mean() > 0.1 and jump[~third].max() < 2 * eps * 2 * 2 + 1e-5


def test_the_default_super_block_shape_is_the_layouts_own():
    from perf_amd.grid import SB_SHIFT
    for layout in ('line_local', 'line_overlap'):
        cfg = GridConfig(n_levels=L, log2_hashmap_size=24, base_resolution=16, per_level_scale=B, layout=layout)
        lv = O.grid_levels(L, 2, 24, 16, B, layout=layout)
        assert cfg.sb_shift == SB_SHIFT[layout] == O.SB_SHIFT[layout] == tuple(lv.sb_shift)
        assert cfg.total == lv.total and np.array_equal(np.asarray(cfg.offset).astype(np.int64), np.asarray(lv.offset).astype(np.int64))
    assert SB_SHIFT['line_overlap'] == (7, 5, 7) and sum(SB_SHIFT['line_overlap']) == sum(SB_SHIFT['line_local']) == 19      # 2 MiB either way


def test_a_level_slice_carries_the_layout():
    from perf_amd import _lib
    from perf_amd.sharded import GridSlice
    cfg, lv = _levels(20, (3, 3, 2), 64)
    first_local = int(np.argmax(lv.local))
    sl = GridSlice(cfg, [first_local - 1, first_local, first_local + 3])
    d = sl.desc()
    assert d.layout == _lib.LAYOUT_LINE_OVERLAP and cfg.desc().layout == _lib.LAYOUT_LINE_OVERLAP
    assert int(d.nsx[1]) == int(cfg.nsx[first_local]) and int(d.size[2]) == int(cfg.size[first_local + 3])
