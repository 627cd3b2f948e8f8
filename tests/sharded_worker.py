"""Worker of tests/test_gpu_dist.py::test_level_sharded_encode_*: WORLD_SIZE ranks share one GPU (gloo); every rank
encodes ITS samples through the level-sharded encoder and rank 0 checks them against the unsharded kernel on the same
table.   python -m torch.distributed.run --nproc-per-node 2 ... tests/sharded_worker.py <out.pt> <n_levels> <log2_T>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path, n_levels, log2_t = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0'))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    from perf_amd import ops
    from perf_amd.grid import GridConfig
    from perf_amd.sharded import LevelShardedEncoder, assign_levels
    b = float(torch.exp(torch.log(torch.tensor(2048.0 / 16)) / (n_levels - 1)))
    cfg = GridConfig(n_levels=n_levels, log2_hashmap_size=log2_t, base_resolution=16, per_level_scale=b)
    enc = LevelShardedEncoder(cfg, dtype='fp16', seed=99)
    n = 6000
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.rand(n, 3, generator=g).cuda()
    feat = enc.encode(x)
    # reference: the unsharded table = the same random stream, level by level
    gt = torch.Generator().manual_seed(99)
    full = torch.cat([(torch.rand(int(cfg.size[l]) * 2, generator=gt) * 2 - 1) * 1e-4 for l in range(n_levels)]).half().cuda()
    ref = ops.hashgrid_fwd(cfg, x, full)
    ok_fwd = bool(torch.equal(feat, ref))
    # backward: every rank brings the gradient of its own samples; the sharded table gradient must equal the slice of the
    # unsharded gradient over ALL ranks' samples
    dfeat = torch.randn(n_levels, n, 2, generator=g).cuda()
    grad_local = enc.table_gradient(dfeat)
    xs = [torch.empty_like(x) for _ in range(world)]; ds = [torch.empty_like(dfeat) for _ in range(world)]
    dist.all_gather(xs, x); dist.all_gather(ds, dfeat)
    ref_grad = ops.hashgrid_bwd(cfg, torch.cat(xs), torch.cat(ds, 1).contiguous())
    sl = torch.cat([ref_grad[2 * int(cfg.offset[l]): 2 * int(cfg.offset[l] + cfg.size[l])] for l in enc.local.levels])
    err = float((grad_local - sl).abs().max() / (sl.abs().max() + 1e-20))
    res = torch.tensor([float(ok_fwd), err], device='cuda')
    allr = [torch.empty_like(res) for _ in range(world)]
    dist.all_gather(allr, res)
    if rank == 0:
        torch.save({'fwd_equal': [bool(r[0].item()) for r in allr], 'bwd_rel_err': [float(r[1].item()) for r in allr],
                    'assignment': assign_levels(cfg, world), 'world': world}, out_path)
    dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    main()
