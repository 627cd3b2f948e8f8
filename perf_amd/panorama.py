"""BASELINE config 5 on one GPU (4096x2048 panorama x 256 samples per ray, L = 20 hash grids whose 16-bit tables are sized to
HBM): the fixed-count eval renderer and the row-block panorama loop over `NeRFOCCRenderer.render` that bench.py's `config5` block
times and tests/test_gpu_config5.py checks.  Rays shard by panorama rows (eval needs no communication): a rank
renders `render_rows(row0, nrows)` of its own.  tools/config5.py is the command line around this module."""
import time

import torch

from . import ops

H5, W5, SPP5, LEVELS5 = 2048, 4096, 256, 20
FINEST5 = 8192.0
ALGO_BYTES_PER_ENCODE5 = LEVELS5 * 8 * 2 * 2           # L x 2^3 corners x F x sizeof(16-bit) = 640 B at L = 20


def per_level_scale(levels=LEVELS5, finest=FINEST5, base=16):
    import math
    return math.exp(math.log(finest / base) / (levels - 1))


def make_renderer(spp=SPP5):
    """(estimator, renderer) of the fixed-count eval render: all-occupied grid, `spp` lattice intervals of 0.99 / spp, the
    reference's early stop (T < 1e-4), one-phase density pass -- what bench.py's `render` block uses for config 2."""
    from perf_amd.nerfacc_impl import OccGridEstimator
    from perf_amd.renderer import NeRFOCCRenderer
    aabb = [-1., -1, -1, 1, 1, 1]
    est = OccGridEstimator(aabb, resolution=256).cuda(); est.eval()
    est.set_binaries(torch.ones(256 ** 3, dtype=torch.uint8, device='cuda'))
    rend = NeRFOCCRenderer(max_radius=2, bg_color='rand_noise'); rend.eval()
    rend.render_step_size = 0.99 / spp
    rend.max_steps = spp
    rend.head_samples = None
    return est, rend


@torch.no_grad()
def render_rows(nerf, est, rend, row0, nrows, rows_per_batch=4, spp=SPP5, height=H5, width=W5, outs=None, counters=None,
                keep=('rgb', 'distance', 'opacities'), bookkeeping=None, tile=None, max_batches=None, ray_order='row'):
    """Rows [row0, row0 + nrows) of the height x width panorama through NeRFOCCRenderer.render (marching, no-grad density pass,
    visibility compaction, colour field, compositing), `rows_per_batch` rows (x width rays x spp samples) per batch, rays
    generated in-kernel, device-side counts.  outs: {key: [nrows * width, C]} preallocated (or None: allocated);
    counters: int64 [8] device block accumulating {marched, kept} (perf_step_bookkeeping).  bookkeeping(res, lo, R) is called per
    batch with the renderer's result dict (tests; lo = batch number x R).
    tile = (rows, columns): the batches are 2-D TILES of pixels instead of full-width strips (same pixels, same values: rays are
    independent; what changes is which rays share a launch -- the neighbours of a ray in BOTH image directions, whose samples
    meet in the same table lines).  ray_order ('row' | 'morton', tiles only): the order of a tile's rays inside its batch -- row by row,
    or along the Z-order curve (consecutive rays = a compact 2-D patch of pixels: the 2 x 2 rays of a workgroup and the 2 x 4 of an
    XCD's turn in the deep-grid encode kernel are neighbours in both directions).  max_batches: stop after that many batches (timing passes)."""
    from perf_amd import ops
    pose = torch.eye(4, device='cpu')
    n = nrows * width
    if outs is None:
        cw = {'rgb': 3, 'distance': 1, 'opacities': 1}
        outs = {k: torch.empty(n, cw[k], dtype=torch.float32, device='cuda') for k in keep}
    if tile is not None:
        th, tw = tile
        b = 0
        for r in range(row0, row0 + nrows, th):
            nr = min(th, row0 + nrows - r)
            o_band, d_band = ops.pano_raygen(pose, height, width, row0=r, nrows=nr)              # [nr, width, 3]
            for c in range(0, width, tw):
                nc = min(tw, width - c)
                o = o_band[:, c:c + nc].reshape(-1, 3).contiguous(); d = d_band[:, c:c + nc].reshape(-1, 3).contiguous()
                R = o.shape[0]
                inv = None
                if ray_order == 'morton':
                    perm, inv = _morton_order(nr, nc)
                    o = o[perm]; d = d[perm]
                rend.sample_capacity = R * spp
                near = torch.zeros(R, 1, device='cuda'); far = torch.ones(R, 1, device='cuda')
                res = rend.render(nerf, est, o, d, near, far)
                for k in outs:
                    outs[k].view(nrows, width, -1)[r - row0:r - row0 + nr, c:c + nc].copy_((res[k] if inv is None else res[k][inv]).view(nr, nc, -1))
                if counters is not None:
                    ops.step_bookkeeping(None, None, counters, res['n_marched_dev'], res['n_samples_dev'])
                if bookkeeping is not None:
                    bookkeeping(res, b * R, R)
                b += 1
                if max_batches is not None and b >= max_batches:
                    return outs
        return outs
    for r in range(row0, row0 + nrows, rows_per_batch):
        nr = min(rows_per_batch, row0 + nrows - r)
        o, d = ops.pano_raygen(pose, height, width, row0=r, nrows=nr)
        o = o.reshape(-1, 3); d = d.reshape(-1, 3)
        R = o.shape[0]
        rend.sample_capacity = R * spp
        near = torch.zeros(R, 1, device='cuda'); far = torch.ones(R, 1, device='cuda')
        res = rend.render(nerf, est, o, d, near, far)
        lo = (r - row0) * width
        for k in outs:
            outs[k][lo:lo + R].copy_(res[k])
        if counters is not None:
            ops.step_bookkeeping(None, None, counters, res['n_marched_dev'], res['n_samples_dev'])
        if bookkeeping is not None:
            bookkeeping(res, lo, R)
    return outs


_MORTON = {}


def _morton_order(nr, nc):
    """(perm, inv) int64 device tensors: perm[m] = row-major index of the m-th pixel of an nr x nc tile along the Z-order curve
    (bits of row and column interleaved; any shape: pixels sorted by their Z-order key), inv its inverse."""
    key = (nr, nc)
    if key not in _MORTON:
        rr, cc = torch.meshgrid(torch.arange(nr), torch.arange(nc), indexing='ij')
        z = torch.zeros(nr, nc, dtype=torch.int64)
        for b in range(16):
            z |= ((cc >> b) & 1) << (2 * b)
            z |= ((rr >> b) & 1) << (2 * b + 1)
        perm = torch.argsort(z.reshape(-1), stable=True)
        inv = torch.empty_like(perm); inv[perm] = torch.arange(perm.numel())
        _MORTON[key] = (perm.cuda(), inv.cuda())
    return _MORTON[key]


def render_panorama_block(log2_t, rows_per_batch=4, spp=SPP5, height=H5, width=W5, levels=LEVELS5, dtype='fp16', timing_batches=8,
                          pmc=None, layout='tcnn', tile=(128, 128), ray_order='row', **layout_kw):
    """BASELINE config 5 on ONE GPU, whole panorama: height x width rays x spp samples through both L-level fields (16-bit tables
    of 2^log2_t entries per hashed level, inference only: perf_amd.fields.InferenceNeRF) + compositing.  -> dict for bench.py's
    `config5` block: ray-samples/s, the encode kernel's algorithmic fraction of the HBM peak, and -- from the committed PMC pass
    `pmc` (profiles/r05_config5_pmc.json) when it holds this table size -- the MOVED fraction."""
    from perf_amd import ops
    from perf_amd.fields import InferenceNeRF
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nerf = InferenceNeRF([-1., -1, -1, 1, 1, 1], n_levels=levels, log2_hashmap_size=log2_t, per_level_scale=per_level_scale(levels), dtype=dtype,
                         layout=layout, **layout_kw)
    est, rend = make_renderer(spp)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    counters = ops.step_counters('cuda')
    if tile is not None and tile[0] * tile[1] != rows_per_batch * width:
        raise ValueError('render_panorama_block: a tile holds as many rays as a strip batch (the launches are compared at equal size)')
    outs = render_rows(nerf, est, rend, height // 2, rows_per_batch, rows_per_batch, spp, height, width)          # warm-up: one batch
    outs = {k: torch.empty(height * width, v.shape[1], dtype=torch.float32, device='cuda') for k, v in outs.items()}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    render_rows(nerf, est, rend, 0, height, rows_per_batch, spp, height, width, outs=outs, counters=counters, tile=tile, ray_order=ray_order)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    c = counters.tolist()
    marched, kept = int(c[0]), int(c[1])
    # the encode kernel alone: HIP events around the launches of `timing_batches` batches spread from pole to pole
    ops.start_kernel_timing()
    step_rows = max(height // timing_batches, rows_per_batch)
    nb = 0
    for r in range(step_rows // 2, height - rows_per_batch + 1, step_rows):
        if tile is None:
            render_rows(nerf, est, rend, r, rows_per_batch, rows_per_batch, spp, height, width); nb += 1
        else:       # one tile of the band at that latitude (a fresh one: another column range each time)
            r0 = min(r, height - tile[0])
            keep1 = render_rows(nerf, est, rend, r0, tile[0], rows_per_batch, spp, height, width, tile=tile, max_batches=2, ray_order=ray_order); nb += 2
            del keep1
    kern = ops.stop_kernel_timing()
    enc_n, enc_ms = kern['perf_hashgrid_fwd']
    per_launch = rows_per_batch * width * spp                    # nothing is pruned at a fresh initialisation: kept = marched
    algo = levels * 8 * 2 * 2
    enc_gbs = algo * per_launch / (enc_ms * 1e-3) / 1e9
    blk = {'what': f'BASELINE config 5 on one GPU: {width}x{height} panorama x {spp} samples/ray, L = {levels} hash grids up to resolution '
                   f'{int(FINEST5)}, T = 2^{log2_t} ({dtype} tables only: inference), both fields + compositing through NeRFOCCRenderer.render, '
                   f'{height // rows_per_batch} batches of {rows_per_batch * width} rays ' + ('(full-width strips of %d rows)' % rows_per_batch if tile is None else
                                                                                            '(%d x %d-pixel tiles)' % tile) + ', fresh initialisation (nothing pruned), device-side counts',
           'log2_hashmap_size': log2_t, 'table_layout': layout, 'batch_shape': 'strip' if tile is None else list(tile), 'ray_order': ray_order if tile is not None else 'row', 'table_GiB_per_encoder': round(nerf.table_bytes() / 2 ** 30, 2),
           'table_entries': int(nerf.grid.total), 'offsets_exceed_32_bit': bool(nerf.grid.n_params >= 2 ** 32),
           'distinct_vertices_per_entry_of_hashed_levels': 0.75 if layout == 'line_overlap' else 1.0,      # (overlapping x runs: 24 vertices in 32 entries)
           'build_seconds': round(t_build, 3), 'seconds_per_panorama': round(el, 4), 'rays_per_s': height * width / el,
           'ray_samples_per_s': kept / el, 'marched_samples': marched, 'kept_samples': kept,
           'output_checksum': {k: float(v.double().sum()) for k, v in outs.items()},
           'roofline': {'bound': 'hbm', 'unit': 'GB/s', 'peak': 8000.0, 'kernel': 'perf_hashgrid_fwd (generic L-level encode)',
                        'algorithmic_bytes_per_encode_sample': algo, 'ms_per_launch': round(enc_ms, 4), 'launches_timed': enc_n,
                        'samples_per_launch': per_launch, 'achieved': round(enc_gbs, 1), 'frac': round(enc_gbs / 8000.0, 4),
                        'whole_render_algorithmic_GBps': round(2 * algo * kept / el / 1e9, 1),
                        'whole_render_frac': round(2 * algo * kept / el / 1e9 / 8000.0, 4), 'traffic': None, 'moved_frac': None,
                        'definition': 'achieved = 640 B (20 levels x 8 corners x 2 features x 2 B) x samples of a launch / mean launch duration '
                                      '(HIP events); whole_render = 2 encodes x 640 B x kept ray-samples / wall time of the panorama'},
           'kernel_ms_per_batch': {k: round(n_ * ms / nb, 3) for k, (n_, ms) in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:6]}}
    if pmc:
        row = (pmc.get('tables') or {}).get(f'T{log2_t}' + ('' if layout == 'tcnn' else '_' + layout))
        if row and row.get('samples_per_launch') == per_launch:
            moved = row['hbm_bytes_per_launch']
            blk['roofline']['traffic'] = moved
            blk['roofline']['moved_GBps'] = round(moved / (enc_ms * 1e-3) / 1e9, 1)
            blk['roofline']['moved_frac'] = round(moved / (enc_ms * 1e-3) / 1e9 / 8000.0, 4)
            blk['roofline']['traffic_source'] = pmc.get('source')
    del nerf, outs
    torch.cuda.empty_cache()
    return blk
