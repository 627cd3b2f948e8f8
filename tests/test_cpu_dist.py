"""world_size-2 gloo test (CPU) of the data-parallel logic: index-stream slicing and global loss normalisation make
the all-reduced gradient equal to the single-process gradient.  The field here is a tiny torch stand-in: what is
tested is the sharding arithmetic of perf_amd/scene.py, which is device independent."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _global_batch(n_pool, bs, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n_pool, (bs,), generator=g)


def _loss_pieces(w, feats, target, idx, bs):
    pred = feats[idx] @ w
    return F.smooth_l1_loss(pred, target[idx], beta=1e-2, reduction='sum') / bs        # local sum / GLOBAL count


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    n_pool, bs = 1000, 64
    feats = torch.randn(n_pool, 8); target = torch.randn(n_pool, 1)
    w = torch.zeros(8, 1, requires_grad=True)
    idx = _global_batch(n_pool, bs, seed=1234)                    # same stream on every rank ...
    per = bs // world
    local = idx[rank * per:(rank + 1) * per]                       # ... contiguous slice per rank (scene.py)
    loss = _loss_pieces(w, feats, target, local, bs) * 128.0
    loss.backward()
    dist.all_reduce(w.grad, op=dist.ReduceOp.SUM)                  # the one collective of the step
    if rank == 0:
        torch.save(w.grad.clone(), out)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradient_equals_single_process(tmp_path):
    out = str(tmp_path / 'g.pt')
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    g2 = torch.load(out)
    torch.manual_seed(0)
    feats = torch.randn(1000, 8); target = torch.randn(1000, 1)
    w = torch.zeros(8, 1, requires_grad=True)
    idx = _global_batch(1000, 64, seed=1234)
    (_loss_pieces(w, feats, target, idx, 64) * 128.0).backward()
    assert torch.allclose(g2, w.grad, rtol=1e-5, atol=1e-6)


def test_pool_slicing_matches_reference_stream():
    """SupInfoPool.rand_ray_color_data(rank, world_size): the union of the rank slices is the 1-GPU batch
    (the pool's gather logic is plain torch, so it runs on CPU tensors)."""
    from perf_amd.scene import SupInfoPool
    g = torch.Generator().manual_seed(3)
    n = 777
    pool = SupInfoPool()
    pool.register_rays(torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g), torch.rand(n, 3, generator=g),
                       torch.rand(n, 1, generator=g))
    full = pool.rand_ray_color_data(64, generator=torch.Generator().manual_seed(5))
    parts = [pool.rand_ray_color_data(64, generator=torch.Generator().manual_seed(5), rank=r, world_size=2) for r in range(2)]
    assert torch.equal(torch.cat([p[0].d for p in parts]), full[0].d)
    assert torch.equal(torch.cat([p[1] for p in parts]), full[1])
    assert torch.equal(torch.cat([p[2] for p in parts]), full[2])
