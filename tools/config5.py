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


from perf_amd.panorama import (ALGO_BYTES_PER_ENCODE5, FINEST5, H5, LEVELS5, SPP5, W5, make_renderer, per_level_scale,      # noqa: E402,F401
                               render_panorama_block, render_rows)


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
    ap.add_argument('--layout', default='tcnn', choices=['tcnn', 'line_local', 'line_overlap'], help='table layout of the --pano-log2 fields (perf_amd.grid.GridConfig)')
    ap.add_argument('--sb-shift', type=int, nargs=3, default=None, help='line_local: log2 vertices of a super-block along x, y, z')
    ap.add_argument('--local-min-res', type=int, default=None, help='line_local: levels of at least this resolution are stored line-local')
    ap.add_argument('--strips', action='store_true', help='--pano-log2: 4-row strips instead of the default 128 x 128-pixel tiles')
    ap.add_argument('--ray-order', default='row', choices=['row', 'morton'], help='order of a tile\'s rays inside its batch (perf_amd.panorama.render_rows)')
    ap.add_argument('--tile', type=int, nargs=2, default=None, help='batches are tiles of ROWS x COLUMNS pixels instead of 4-row strips')
    ap.add_argument('--pano-batches', type=int, default=0,
                    help='with --pano-log2: only this many 4-row batches spread from pole to pole instead of the whole panorama (what the rocprofv3 '
                         '--pmc passes of profiles/r05_config5_pmc.json run)')
    args = ap.parse_args()
    dev = 'cuda'
    if args.pano_log2:
        res = {}
        lkw = {}
        if args.sb_shift is not None:
            lkw['sb_shift'] = tuple(args.sb_shift)
        if args.local_min_res is not None:
            lkw['local_min_res'] = args.local_min_res
        for T in args.pano_log2:
            if args.pano_batches > 0:
                from perf_amd.fields import InferenceNeRF
                nerf = InferenceNeRF([-1., -1, -1, 1, 1, 1], n_levels=LEVELS5, log2_hashmap_size=T, per_level_scale=per_level_scale(), dtype='fp16',
                                     layout=args.layout, **lkw)
                est, rend = make_renderer(SPP5)
                step_rows = H5 // args.pano_batches
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for r in range(step_rows // 2, H5 - 3, step_rows):
                    if args.tile:
                        render_rows(nerf, est, rend, min(r, H5 - args.tile[0]), args.tile[0], 4, tile=tuple(args.tile), max_batches=1, ray_order=args.ray_order)
                    else:
                        render_rows(nerf, est, rend, r, 4, 4)
                torch.cuda.synchronize()
                res[f'T{T}'] = {'batches': args.pano_batches, 'seconds': time.perf_counter() - t0, 'samples_per_launch': 4 * W5 * SPP5}
                del nerf
                torch.cuda.empty_cache()
            else:
                res[f'T{T}'] = render_panorama_block(T, layout=args.layout, tile=tuple(args.tile) if args.tile else (None if args.strips else (128, 128)), ray_order=args.ray_order, **lkw)
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
