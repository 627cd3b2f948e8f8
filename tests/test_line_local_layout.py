"""CPU: the opt-in line-local table layout (perf_amd.grid.GridConfig(layout='line_local'), oracle/perf_oracle.py:grid_levels /
grid_corner_indices) -- the host-side level table equals the oracle's, and the index function has the properties the
layout is defined by: a dense level is addressed injectively, a 4 x 4 x 2-vertex block is one 128-byte line, a super-block is one
contiguous range, a hashed level keeps whole super-blocks together, and the eight corners of a cell lie in (1 + 1/4)(1 + 1/4)(1 + 1/2)
= 2.3 lines on average.  (Values against the oracle on the GPU: tests/test_gpu_ops.py::test_deep_grid_forward_both_layouts.)"""
import numpy as np
import pytest

from oracle import perf_oracle as O
from perf_amd.grid import GridConfig

L, B = 20, 1.3819


def _levels(log2_t, sb_shift, min_res):
    cfg = GridConfig(n_levels=L, log2_hashmap_size=log2_t, base_resolution=16, per_level_scale=B, layout='line_local', sb_shift=sb_shift,
                     local_min_res=min_res)
    lv = O.grid_levels(L, 2, log2_t, 16, B, layout='line_local', sb_shift=sb_shift, local_min_res=min_res)
    return cfg, lv


@pytest.mark.parametrize('log2_t,sb_shift,min_res', [(15, (2, 2, 1), 16), (20, (3, 3, 2), 64), (24, (5, 6, 8), 64), (28, (5, 6, 8), 64)])
def test_host_level_table_equals_the_oracle(log2_t, sb_shift, min_res):
    cfg, lv = _levels(log2_t, sb_shift, min_res)
    assert cfg.total == lv.total
    for name in ('res', 'size', 'offset', 'hashed', 'local', 'nsx', 'nsxy'):
        assert np.array_equal(np.asarray(getattr(cfg, name)).astype(np.int64), np.asarray(getattr(lv, name)).astype(np.int64)), name
    assert np.array_equal(np.asarray(cfg.scale, np.float32), lv.scale)
    per_sb = 1 << sum(sb_shift)
    for l in range(L):
        assert bool(lv.local[l]) == (int(lv.res[l]) >= min_res)
        if lv.local[l]:
            assert int(lv.size[l]) % per_sb == 0 and int(lv.offset[l]) % per_sb == 0     # whole super-blocks, starting on a super-block boundary: a block IS a cache line
            if lv.hashed[l]:
                ns = int(lv.size[l]) // per_sb
                assert ns & (ns - 1) == 0                                                  # the slot hash is a mask
    # the coarse levels keep tcnn's rule
    ref = O.grid_levels(L, 2, log2_t, 16, B)
    for l in range(L):
        if not lv.local[l]:
            assert int(lv.size[l]) == int(ref.size[l]) and bool(lv.hashed[l]) == bool(ref.hashed[l])


def test_a_level_slice_keeps_the_alignment_rule():
    """perf_amd.sharded.GridSlice (a rank's levels of a level-sharded table): line-local levels start on super-block boundaries of the
    slice's own table, and pack() puts each level's entries at its offset."""
    import torch
    from perf_amd.sharded import GridSlice
    cfg, lv = _levels(20, (3, 3, 2), 64)
    per_sb = 1 << 8
    first_local = int(np.argmax(lv.local))
    sl = GridSlice(cfg, [first_local - 1, first_local, first_local + 3])
    assert int(sl.offset[0]) == 0 and int(sl.offset[1]) % per_sb == 0 and int(sl.offset[2]) % per_sb == 0
    assert int(sl.offset[1]) >= int(cfg.size[first_local - 1]) and sl.total == int(sl.offset[2]) + int(cfg.size[first_local + 3])
    parts = [torch.full((2 * int(cfg.size[l]),), float(k + 1)) for k, l in enumerate(sl.levels)]
    t = sl.pack(parts, torch.float32)
    assert t.numel() == 2 * sl.total
    for k, l in enumerate(sl.levels):
        lo = 2 * int(sl.offset[k])
        assert bool((t[lo: lo + 2 * int(cfg.size[l])] == k + 1).all())
    assert float(t.sum()) == sum((k + 1) * 2 * int(cfg.size[l]) for k, l in enumerate(sl.levels))      # zeros between


def _vertex_index(lv, l, v):
    """Table entry of integer vertex v [N, 3] of level l: corner 0 of the cell that starts there (the oracle's own function, fed the
    point at that cell's centre)."""
    x = ((v.astype(np.float64)) / float(lv.scale[l])).astype(np.float32)                  # pos = x * scale + 0.5 -> floor = v
    idx, _ = O.grid_corner_indices(x, lv, l)
    return idx


def test_a_dense_line_local_level_is_addressed_injectively_block_by_block():
    cfg, lv = _levels(24, (3, 3, 2), 64)
    l = int(np.argmax(lv.local & ~lv.hashed))
    r = int(lv.res[l])
    g = np.arange(0, r - 1)
    v = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(-1, 3)
    idx = _vertex_index(lv, l, v)
    assert int(idx.max()) < int(lv.size[l])
    assert np.unique(idx[:, 0]).size == v.shape[0]                                        # one entry per vertex
    # the eight corners of a cell are the entries of the eight vertices
    look = {tuple(p): int(i) for p, i in zip(v[:4096], idx[:4096, 0])}
    for p, row in zip(v[:4096], idx[:4096]):
        for c in range(8):
            q = (p[0] + (c & 1), p[1] + ((c >> 1) & 1), p[2] + (c >> 2))
            if q in look:
                assert look[q] == int(row[c])
    # a 4 x 4 x 2-vertex block is one 128-byte line (32 consecutive entries, x fastest, then y, then z)
    blocks = {}
    for p, i in zip(v, idx[:, 0]):
        blocks.setdefault((p[0] >> 2, p[1] >> 2, p[2] >> 1), []).append((int(i), tuple(p)))
    full = [b for b in blocks.values() if len(b) == 32]
    assert len(full) > 100
    for b in full[:200]:
        ents = sorted(b)
        assert ents[0][0] % 32 == 0 and [e[0] for e in ents] == list(range(ents[0][0], ents[0][0] + 32))
        x0, y0, z0 = ents[0][1]
        assert [e[1] for e in ents] == [(x0 + (k & 3), y0 + ((k >> 2) & 3), z0 + (k >> 4)) for k in range(32)]
    # a super-block (8 x 8 x 4 vertices here) is one contiguous range of 256 entries
    sbs = {}
    for p, i in zip(v, idx[:, 0]):
        sbs.setdefault((p[0] >> 3, p[1] >> 3, p[2] >> 2), []).append(int(i))
    whole = [e for e in sbs.values() if len(e) == 256]
    assert len(whole) > 20
    for ents in whole:
        assert min(ents) % 256 == 0 and sorted(ents) == list(range(min(ents), min(ents) + 256))


@pytest.mark.parametrize('log2_t,sb_shift', [(15, (2, 2, 1)), (20, (3, 3, 2)), (24, (5, 6, 8))])
def test_a_hashed_line_local_level_keeps_super_blocks_together_and_a_cell_in_few_lines(log2_t, sb_shift):
    cfg, lv = _levels(log2_t, sb_shift, 16 if log2_t == 15 else 64)
    l = L - 1
    assert lv.local[l] and lv.hashed[l]
    rng = np.random.default_rng(5)
    x = rng.random((20000, 3), dtype=np.float32) * 0.999
    idx, _ = O.grid_corner_indices(x, lv, l)
    assert int(idx.max()) < int(lv.size[l])
    per_sb = 1 << sum(sb_shift)
    g = np.floor(O.grid_pos(x, lv.scale[l])).astype(np.int64)
    sb = (g[:, 0] >> sb_shift[0], g[:, 1] >> sb_shift[1], g[:, 2] >> sb_shift[2])
    # two vertices of the same super-block share its slot; inside the slot the offset is the dense block rule
    key = sb[0] + (sb[1] << 20) + (sb[2] << 40)
    slot = idx[:, 0] // per_sb
    first = {}
    for k, s in zip(key.tolist(), slot.tolist()):
        assert first.setdefault(k, s) == s
    # lines touched by the eight corners of a cell: 2.34 on average, never more than 8, one when the cell sits inside a block
    lines = np.array([np.unique(row >> 5).size for row in idx.astype(np.int64)])
    assert 2.2 < lines.mean() < 2.5 and lines.max() <= 8
    inside = ((g[:, 0] & 3) < 3) & ((g[:, 1] & 3) < 3) & ((g[:, 2] & 1) < 1)
    assert inside.sum() > 1000 and (lines[inside] == 1).all()
