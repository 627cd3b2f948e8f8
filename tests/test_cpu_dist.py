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


# ---- NeRFScene._apply_grad itself with world_size 2 (gloo, CPU tensors): the real product function ----------------------
def _apply_worker(rank, world, port, out, comm_dtype):
    """Both ranks hold the same parameters; rank r contributes gradient g_r and sample count c_r.  Three steps: both ranks
    have samples; only rank 1 has; none has (the step must be skipped on BOTH ranks)."""
    import types
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from perf_amd.scene import NeRFScene
    n = 1000
    net = types.SimpleNamespace(params=torch.nn.Parameter(torch.linspace(-1, 1, n)))
    opt = torch.optim.Adam([net.params], lr=1e-2)
    me = types.SimpleNamespace(comm_dtype=comm_dtype, sample_counters=None, _capturing=False, _steps_since_check=-10 ** 9)
    hist = []
    g = torch.Generator().manual_seed(100 + rank)
    for counts in ((5, 7), (0, 3), (0, 0)):
        grad = torch.zeros(n + 1)
        if counts[rank] > 0:
            grad[:n] = torch.randn(n, generator=g)
        ran = {'overlap': False}
        NeRFScene._apply_grad(me, net, grad, opt, (dist, rank, world), lambda: ran.__setitem__('overlap', True), n_kept=counts[rank])
        assert ran['overlap']
        hist.append(net.params.detach().clone())
    if rank == 0:
        torch.save(hist, out)
    dist.barrier()
    dist.destroy_process_group()


def test_apply_grad_world2_sums_gradients_and_skips_empty_steps(tmp_path):
    import pytest
    for comm_dtype, tol in (('fp32', 1e-6), ('bf16', 2e-2)):
        out = str(tmp_path / f'h_{comm_dtype}.pt')
        mp.spawn(_apply_worker, args=(2, _free_port(), out, comm_dtype), nprocs=2, join=True)
        hist = torch.load(out)
        # single-process reference: Adam on the summed gradients of the steps that had samples somewhere
        p = torch.nn.Parameter(torch.linspace(-1, 1, 1000)); opt = torch.optim.Adam([p], lr=1e-2)
        gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
        ref = []
        for counts in ((5, 7), (0, 3), (0, 0)):
            total = torch.zeros(1000)
            for r in range(2):
                if counts[r] > 0:
                    total += torch.randn(1000, generator=gens[r])
            if sum(counts) > 0:
                p.grad = total
                opt.step()
            ref.append(p.detach().clone())
        for a, b in zip(hist, ref):
            # Adam's first steps are sign-like: compare the travelled distance, tightly for the exact fp32 payload
            assert float((a - b).abs().max()) <= tol, (comm_dtype, float((a - b).abs().max()))
        assert torch.equal(hist[2], hist[1])                        # the all-empty step changed nothing
