"""Worker of tests/test_gpu_dist.py::test_config5_row_shard_through_the_level_sharded_path: WORLD_SIZE ranks share one GPU
(gloo).  BASELINE config 5's inference path at its stated shape -- a row shard of a 4096x2048 panorama, 256 samples per ray,
L = 20 hash grids -- rendered by every rank through the LEVEL-SHARDED fields (tables cut by level over the ranks, positions
all-gathered, one all-to-all of features per field) and, for comparison, through the unsharded fields on the same rays.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/config5_worker.py <out.pt> <rows_per_rank> <log2_T> [tcnn|line_local|line_overlap]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path, rows, log2_t = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    layout = sys.argv[4] if len(sys.argv) > 4 else 'tcnn'
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0'))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    from perf_amd import ops
    from perf_amd.fields import NGPNeRF
    from perf_amd.nerfacc_impl import OccGridEstimator
    from perf_amd.renderer import NeRFOCCRenderer
    from perf_amd.sharded import LevelShardedNeRF
    AABB = [-1., -1, -1, 1, 1, 1]
    H, W, SPP = 2048, 4096, 256
    nerf = NGPNeRF(aabb=AABB, n_levels=20, dtype='fp16', log2_hashmap_size=log2_t)        # same seed on every rank: same parameters
    with torch.no_grad():                                                                # non-trivial fields: tables U(-1, 1)
        for net in (nerf.geo_mlp, nerf.app_mlp):
            net.params[net.mlp.n_params:] *= 1e4
    nerf.eval()
    if layout != 'tcnn':
        # the opt-in line-local table layout exists for the 16-bit-only inference fields (same seed on every rank: same tables)
        from perf_amd.fields import InferenceNeRF
        from perf_amd.panorama import per_level_scale
        nerf = InferenceNeRF(AABB, n_levels=20, log2_hashmap_size=log2_t, per_level_scale=per_level_scale(20), dtype='fp16', table_scale=1.0,
                             density_bias=2.0, layout=layout)
    sharded = LevelShardedNeRF(nerf).eval()
    est = OccGridEstimator(AABB, resolution=256).cuda(); est.eval()
    est.set_binaries(torch.ones(256 ** 3, dtype=torch.uint8, device='cuda'))
    rend = NeRFOCCRenderer(max_radius=2, bg_color='rand_noise'); rend.eval()
    rend.render_step_size = 0.99 / SPP
    rend.max_steps = SPP
    rend.head_samples = None
    row0 = H // 2 - rows * world // 2 + rank * rows                                      # this rank's rows, around the equator
    o, d = ops.pano_raygen(torch.eye(4), H, W, row0=row0, nrows=rows)
    o = o.reshape(-1, 3); d = d.reshape(-1, 3)
    R = o.shape[0]
    rend.sample_capacity = R * SPP
    near, far = torch.zeros(R, 1, device='cuda'), torch.ones(R, 1, device='cuda')
    with torch.no_grad():
        a = rend.render(sharded, est, o, d, near, far)
        b = rend.render(nerf, est, o, d, near, far)
    n_kept = int(b['n_samples_dev'].item()); n_marched = int(b['n_marched_dev'].item())
    same = {k: bool(torch.equal(a[k], b[k])) for k in ('rgb', 'distance', 'opacities')}
    same['kept'] = int(a['n_samples_dev'].item()) == n_kept
    res = {'rank': rank, 'rays': R, 'marched': n_marched, 'kept': n_kept, 'same': same, 'levels': sharded.nets['geo_mlp'][0].local.levels,
           'rgb_range': (float(b['rgb'].min()), float(b['rgb'].max())), 'opacity_mean': float(b['opacities'].mean())}
    allr = [None] * world
    dist.all_gather_object(allr, res)
    if rank == 0:
        torch.save(allr, out_path)
    dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    main()
