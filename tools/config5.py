"""BASELINE config 5: 2048x4096 panorama, 256 samples per ray, L = 20 levels, hash tables that no cache holds
(F = 2, fp16, log2_hashmap_size up to 30: 37 GiB per encoder, 64-bit level offsets), inference.

  python tools/config5.py [--log2 26 28 30] [--rays 16384] [--batches 8]

One batch = `rays` panorama rays x 256 fixed lattice samples: positions -> density field (encode + 40->64->1 MLP)
-> colour field (encode + 40->64->64->3 MLP) -> compositing.  Only 16-bit tables are allocated (no fp32 master, no
optimiser state: such a field cannot be trained on one GPU).  Reports ray-samples/s and the encode
kernel's share; run under `rocprofv3 --pmc FETCH_SIZE` for the moved-bytes roofline (profiles/README.md)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig, MlpConfig


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
                keep=('rgb', 'distance', 'opacities'), bookkeeping=None):
    """Rows [row0, row0 + nrows) of the height x width panorama through NeRFOCCRenderer.render (marching, no-grad density pass,
    visibility compaction, colour field, compositing), `rows_per_batch` rows (x width rays x spp samples) per batch, rays
    generated in-kernel, device-side counts.  outs: {key: [nrows * width, C]} preallocated (or None: allocated);
    counters: int64 [8] device block accumulating {marched, kept} (perf_step_bookkeeping).  bookkeeping(st) is called per batch
    with the renderer's result dict (tests)."""
    from perf_amd import ops
    pose = torch.eye(4)
    n = nrows * width
    if outs is None:
        cw = {'rgb': 3, 'distance': 1, 'opacities': 1}
        outs = {k: torch.empty(n, cw[k], dtype=torch.float32, device='cuda') for k in keep}
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


def render_panorama_block(log2_t, rows_per_batch=4, spp=SPP5, height=H5, width=W5, levels=LEVELS5, dtype='fp16', timing_batches=8,
                          pmc=None):
    """BASELINE config 5 on ONE GPU, whole panorama: height x width rays x spp samples through both L-level fields (16-bit tables
    of 2^log2_t entries per hashed level, inference only: perf_amd.fields.InferenceNeRF) + compositing.  -> dict for bench.py's
    `config5` block: ray-samples/s, the encode kernel's algorithmic fraction of the HBM peak, and -- from the committed PMC pass
    `pmc` (profiles/r05_config5_pmc.json) when it holds this table size -- the MOVED fraction."""
    from perf_amd import ops
    from perf_amd.fields import InferenceNeRF
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nerf = InferenceNeRF([-1., -1, -1, 1, 1, 1], n_levels=levels, log2_hashmap_size=log2_t, per_level_scale=per_level_scale(levels), dtype=dtype)
    est, rend = make_renderer(spp)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    counters = ops.step_counters('cuda')
    outs = render_rows(nerf, est, rend, height // 2, rows_per_batch, rows_per_batch, spp, height, width)          # warm-up: one batch
    outs = {k: torch.empty(height * width, v.shape[1], dtype=torch.float32, device='cuda') for k, v in outs.items()}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    render_rows(nerf, est, rend, 0, height, rows_per_batch, spp, height, width, outs=outs, counters=counters)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    c = counters.tolist()
    marched, kept = int(c[0]), int(c[1])
    # the encode kernel alone: HIP events around the launches of `timing_batches` batches spread from pole to pole
    ops.start_kernel_timing()
    step_rows = max(height // timing_batches, rows_per_batch)
    nb = 0
    for r in range(step_rows // 2, height - rows_per_batch + 1, step_rows):
        render_rows(nerf, est, rend, r, rows_per_batch, rows_per_batch, spp, height, width); nb += 1
    kern = ops.stop_kernel_timing()
    enc_n, enc_ms = kern['perf_hashgrid_fwd']
    per_launch = rows_per_batch * width * spp                    # nothing is pruned at a fresh initialisation: kept = marched
    algo = levels * 8 * 2 * 2
    enc_gbs = algo * per_launch / (enc_ms * 1e-3) / 1e9
    blk = {'what': f'BASELINE config 5 on one GPU: {width}x{height} panorama x {spp} samples/ray, L = {levels} hash grids up to resolution '
                   f'{int(FINEST5)}, T = 2^{log2_t} ({dtype} tables only: inference), both fields + compositing through NeRFOCCRenderer.render, '
                   f'{height // rows_per_batch} batches of {rows_per_batch * width} rays, fresh initialisation (nothing pruned), device-side counts',
           'log2_hashmap_size': log2_t, 'table_GiB_per_encoder': round(nerf.table_bytes() / 2 ** 30, 2),
           'table_entries': int(nerf.grid.total), 'offsets_exceed_32_bit': bool(nerf.grid.n_params >= 2 ** 32),
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
        row = (pmc.get('tables') or {}).get(f'T{log2_t}')
        if row and row.get('samples_per_launch') == per_launch:
            moved = row['hbm_bytes_per_launch']
            blk['roofline']['traffic'] = moved
            blk['roofline']['moved_GBps'] = round(moved / (enc_ms * 1e-3) / 1e9, 1)
            blk['roofline']['moved_frac'] = round(moved / (enc_ms * 1e-3) / 1e9 / 8000.0, 4)
            blk['roofline']['traffic_source'] = pmc.get('source')
    del nerf, outs
    torch.cuda.empty_cache()
    return blk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--log2', type=int, nargs='*', default=[26, 28, 30])
    ap.add_argument('--rays', type=int, default=16384)
    ap.add_argument('--spp', type=int, default=256)
    ap.add_argument('--batches', type=int, default=8)
    ap.add_argument('--levels', type=int, default=20)
    ap.add_argument('--train-log2', type=int, nargs='*', default=[],
                    help='also time a TRAINING step of the density field (fp32 master + Adam state + 16-bit copy: 18 B per '
                         'parameter) at these table sizes: encode, 40->64->1 MLP forward/backward, grid backward, fused Adam')
    ap.add_argument('--pano-log2', type=int, nargs='*', default=[],
                    help="bench.py's `config5` block by itself: the whole panorama through NeRFOCCRenderer.render at these table sizes")
    ap.add_argument('--pano-batches', type=int, default=0,
                    help='with --pano-log2: only this many 4-row batches spread from pole to pole instead of the whole panorama (what the rocprofv3 '
                         '--pmc passes of profiles/r05_config5_pmc.json run)')
    args = ap.parse_args()
    dev = 'cuda'
    if args.pano_log2:
        res = {}
        for T in args.pano_log2:
            if args.pano_batches > 0:
                from perf_amd.fields import InferenceNeRF
                nerf = InferenceNeRF([-1., -1, -1, 1, 1, 1], n_levels=LEVELS5, log2_hashmap_size=T, per_level_scale=per_level_scale(), dtype='fp16')
                est, rend = make_renderer(SPP5)
                step_rows = H5 // args.pano_batches
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for r in range(step_rows // 2, H5 - 3, step_rows):
                    render_rows(nerf, est, rend, r, 4, 4)
                torch.cuda.synchronize()
                res[f'T{T}'] = {'batches': args.pano_batches, 'seconds': time.perf_counter() - t0, 'samples_per_launch': 4 * W5 * SPP5}
                del nerf
                torch.cuda.empty_cache()
            else:
                res[f'T{T}'] = render_panorama_block(T)
            print(json.dumps({f'T{T}': res[f'T{T}']}, indent=1), flush=True)
        os.makedirs('gpurun_out', exist_ok=True)
        json.dump(res, open('gpurun_out/config5_pano.json', 'w'), indent=1)
        return
    H, W = 2048, 4096
    L = args.levels
    b = float(torch.exp(torch.log(torch.tensor(8192.0 / 16)) / (L - 1)))          # finest resolution 8192


    def fill_random(n_elems, dtype):
        t = torch.empty(n_elems, dtype=dtype, device=dev)
        step = 1 << 28
        for lo in range(0, n_elems, step):
            hi = min(lo + step, n_elems)
            t[lo:hi] = (torch.rand(hi - lo, device=dev) * 2 - 1).to(dtype) * 0.1
        return t


    out = {}
    pose = torch.eye(4)
    for T in args.log2:
        cfg = GridConfig(n_levels=L, log2_hashmap_size=T, base_resolution=16, per_level_scale=b)
        geo_mlp = MlpConfig(L, 1, 1, 'Exponential')
        app_mlp = MlpConfig(L, 2, 3, 'Sigmoid')
        gib = cfg.n_params * 2 / 2 ** 30
        tg = fill_random(cfg.n_params, torch.float16)
        ta = fill_random(cfg.n_params, torch.float16)
        wg = (torch.randn(geo_mlp.n_params, device=dev) * 0.2).half()
        wa = (torch.randn(app_mlp.n_params, device=dev) * 0.2).half()
        R, S = args.rays, args.spp
        rows = R // W if R >= W else 1
        aabb = torch.tensor([-1., -1, -1, 1, 1, 1])
        step = 1.4 / S

        def one_batch(k):
            # a block of panorama rows (ray generation in-kernel), fixed lattice of S samples per ray
            row0 = min(H - rows, ((2 * (k % args.batches) + 1) * H) // (2 * args.batches))       # rows spread from pole to pole
            o, d = ops.pano_raygen(pose, H, W, row0=row0, nrows=rows)
            o = o.reshape(-1, 3)[:R]; d = d.reshape(-1, 3)[:R]
            n = R * S
            ri = torch.arange(R, device=dev).repeat_interleave(S)
            ts = (torch.arange(S, device=dev, dtype=torch.float32) * step).repeat(R)
            te = ts + step
            packed = torch.stack([torch.arange(R, device=dev, dtype=torch.int32) * S, torch.full((R,), S, device=dev, dtype=torch.int32)], 1).contiguous()
            x01, sel = ops.points_from_rays(o.contiguous(), d.contiguous(), ri, ts, te, aabb)
            fg = ops.hashgrid_fwd(cfg, x01, tg)
            sig = ops.mlp_fwd(geo_mlp, wg, fg, sel)
            del fg
            fa = ops.hashgrid_fwd(cfg, x01, ta)
            rgb = ops.mlp_fwd(app_mlp, wa, fa, sel)
            del fa
            return ops.composite_fwd(sig.view(-1), rgb, ts, te, packed)

        one_batch(0); torch.cuda.synchronize()
        ops.start_kernel_timing()
        t0 = time.perf_counter()
        for k in range(args.batches):
            res = one_batch(k)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        kern = ops.stop_kernel_timing()
        n_samples = args.batches * R * S
        enc_ms = kern['perf_hashgrid_fwd'][1]
        enc_sps = R * S / (enc_ms * 1e-3)
        out[f'T{T}'] = {'levels': L, 'table_GiB_per_encoder': round(gib, 2), 'total_entries': int(cfg.total), 'offsets_exceed_32_bit': bool(cfg.total >= 2 ** 32),
                        'ray_samples_per_s': n_samples / t, 'ms_per_batch': t / args.batches * 1e3, 'batch': f'{R} rays x {S} spp',
                        'encode_ms_per_launch': round(enc_ms, 3), 'encode_Gsamples_per_s': round(enc_sps / 1e9, 3),
                        'encode_algorithmic_GBs': round(enc_sps * L * 8 * 2 * 2 / 1e9, 1),
                        'kernel_ms_per_batch': {k_: round(c * ms / args.batches, 3) for k_, (c, ms) in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])},
                        'opacity_mean': float(res[3].mean())}
        print(json.dumps({f'T{T}': out[f'T{T}']}, indent=1), flush=True)
        del tg, ta
        torch.cuda.empty_cache()
    for T in args.train_log2:
        # one trainable L-level density field: the tcnn-layout module, its explicit backward and the fused Adam -- the pieces a
        # rank of the level-sharded encoder (perf_amd/sharded.py) runs on its slice of the levels
        from perf_amd import tcnn
        from perf_amd.scene import FusedAdam
        enc = {"otype": "HashGrid", "n_levels": L, "n_features_per_level": 2, "log2_hashmap_size": T, "base_resolution": 16, "per_level_scale": b}
        net = tcnn.NetworkWithInputEncoding(3, 1, enc, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                                       "n_neurons": 64, "n_hidden_layers": 1}, dtype='fp16')
        opt = FusedAdam(net, 1e-3)
        n = 1 << 20
        x = torch.rand(n, 3, device=dev) * 0.98 + 0.01
        dout = torch.randn(n, 1, device=dev) * 1e-3

        def step():
            y = net(x, out_fp32=True)
            y.backward(dout)
            opt.step()

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.start_kernel_timing()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        kern = ops.stop_kernel_timing()
        out[f'train_T{T}'] = {'levels': L, 'params': int(net.params.numel()), 'state_GiB': round(net.params.numel() * 18 / 2 ** 30, 2),
                              'samples_per_step': n, 'ms_per_step': round(t * 1e3, 3), 'samples_per_s': n / t,
                              'grid_gradient_mode': tcnn.GRID_GRAD_ACCUM,
                              'kernel_ms_per_step': {k_: round(c * ms / reps, 3) for k_, (c, ms) in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])}}
        print(json.dumps({f'train_T{T}': out[f'train_T{T}']}, indent=1), flush=True)
        del net, opt
        torch.cuda.empty_cache()
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/config5.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
