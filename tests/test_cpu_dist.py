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
    me = types.SimpleNamespace(comm_dtype=comm_dtype, sample_counters=None, _capturing=False, _dp_timing=None, _poll_health=lambda *a, **k: None,
                               renderer=types.SimpleNamespace(sample_capacity=None))
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


# ---- the sharded exchange (perf_amd/dp.py) with world_size 2 over gloo: int32 reduce-scatter -> Adam on the rank's slice ->
#      all-gather of the 16-bit copy.  The compute steps are torch stand-ins of the HIP kernels (same contracts); what is
#      tested is the choreography: slices and padding, gates, flags, buffer reuse, master gathering. ---------------------------
class _CpuKernels:
    """torch stand-ins of the HIP kernels perf_amd.dp.ShardedExchange drives (same contracts, include/perf_hip.h)."""
    SHIFT = 12
    SLOT = 80

    def __init__(self):
        self.flag = torch.zeros(1, dtype=torch.int32)
        self.units_calls = []

    def stats_pack(self, level_absmax, field_max_prev, n_dev, n, out):
        out.zero_()
        out[:24] = level_absmax.view(torch.int32)
        out[24:48] = -1 if field_max_prev is None else field_max_prev
        live = n if n_dev is None else min(n, int(n_dev))
        out[48] = live

    def units(self, stats_all, world, shifts, n_total, margin_bits=0):
        st = stats_all.view(world, -1)
        shifts.fill_(self.SHIFT)                # (a fixed unit: what is tested is the choreography, not the headroom rule)
        if n_total is not None:
            n_total.fill_(int(st[:, 48].sum()))
        self.seen_field_max = st[:, 24:48].max(0).values.clone()
        self.seen_absmax = st[:, :24].clone().view(torch.float32).max(0).values
        self.units_calls.append(int(margin_bits))

    def unfix(self, shard, lo, hi, shifts, field_max, flag):
        n = 2 * (hi - lo)
        ints = shard[:n].clone()
        field_max.zero_()
        if n:
            field_max[0] = int(ints.abs().max())
            if int(field_max[0]) >= (1 << 29):
                flag.fill_(1)
        shard.view(torch.float32)[:n] = ints.float() * 2.0 ** -self.SHIFT

    def slot_pack(self, level_absmax, field_max, n_dev, n, flag, n_marched, capacity, rank, world, out):
        out.zero_()
        s = out.view(world, self.SLOT)[rank]
        s[:24] = level_absmax
        s[24:48] = (field_max & 0xffff).float(); s[48:72] = (field_max >> 16).float()
        live = n if n_dev is None else min(n, int(n_dev))
        for k in range(4):
            s[72 + k] = float((live >> (16 * k)) & 0xffff)
        s[76] = 1.0 if int(flag) != 0 else 0.0
        s[77] = 1.0 if (n_marched is not None and capacity > 0 and int(n_marched) > capacity) else 0.0

    def slot_unpack(self, slots, world, stats_all, job_flags, n_total):
        s = slots.view(world, self.SLOT)
        job_flags[0] = s[:, 76].sum(); job_flags[1] = s[:, 77].sum()
        n_total.fill_(int(sum(int(s[r, 72 + k]) << (16 * k) for r in range(world) for k in range(4))))
        if stats_all is not None:
            st = stats_all.view(world, -1)
            st.zero_()
            st[:, :24] = s[:, :24].contiguous().view(torch.int32)
            st[:, 24:48] = s[:, 24:48].int() | (s[:, 48:72].int() << 16)
            st[:, 48] = (s[:, 72].int() | (s[:, 73].int() << 16))

    def bookkeeping(self, step_dev, gate, counters, n_marched, n_kept, capacity, overflow, remote_flags, eff_gate):
        truncated = (n_marched is not None and capacity > 0 and int(n_marched) > capacity) or float(remote_flags[1]) > 0.0
        take = int(gate) > 0 and int(overflow) == 0 and float(remote_flags[0]) == 0.0 and not truncated
        if take:
            step_dev += 1
        overflow.zero_()
        eff_gate.fill_(1 if take else 0)

    def adam(self, p, m, v, g, w16, step_dev, lr_dev, gate):
        if int(gate) <= 0:
            return
        t = int(step_dev)
        m.mul_(0.9).add_(g, alpha=0.1)
        v.mul_(0.999).addcmul_(g, g, value=0.001)
        p.sub_(float(lr_dev) / (1 - 0.9 ** t) * m / ((v / (1 - 0.999 ** t)).sqrt() + 1e-8))
        w16.copy_(p)

    def overflow_flag(self):
        return self.flag


def _fields_of(step, rank, n_grid):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return torch.randint(-2000, 2000, (n_grid,), generator=g, dtype=torch.int32)


# (samples rank 0, samples rank 1, what goes wrong on rank 1 ONLY): a local overflow flag of its grid backward; a batch truncated
# at the capacity; fields whose SUM reaches 2^29 in rank 1's slice (the flag is raised by unfix, after the reduce-scatter)
_DP_STEPS = ((5, 7, None), (0, 3, None), (0, 0, None), (4, 4, 'flag'), (2, 2, None), (3, 3, 'truncated'), (6, 1, None),
             (2, 5, 'sum_overflow'), (1, 1, None))
_CAPACITY = 100


def _step_taken(c0, c1, wrong):
    return (c0 + c1) > 0 and wrong is None


def _run_exchange(ex, kern, step_i, rank, rank_counts, wrong, opt, n_net):
    """One step on one rank."""
    amax = torch.full((24,), 0.5 + step_i)
    ex.exchange_units(amax, torch.tensor([rank_counts]), 10 ** 6)
    ex.payload.zero_()
    if rank_counts > 0:
        ex.payload[:ex.n_grid].copy_(_fields_of(step_i, rank, ex.n_grid))
    if wrong == 'sum_overflow':
        ex.payload[2 * 30] = (1 << 28) + 5          # entry 30 lives in rank 1's slice; each rank stays below 2^29, the sum does not
    if wrong == 'flag' and rank == 1:
        kern.flag.fill_(1)
    n_marched = torch.tensor([_CAPACITY + 50 if (wrong == 'truncated' and rank == 1) else 10])
    dw = torch.full((n_net,), float(rank_counts))
    opt.lr_dev.fill_(1e-2)
    return ex.reduce_and_step(dw, opt, n_marched=n_marched, capacity=_CAPACITY).clone()


def _exchange_worker(rank, world, port, out, units):
    import types
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from perf_amd.dp import Collectives, ShardedExchange
    n_net, n_grid = 8, 2 * 37                      # 37 entries: the slices are 20 and 17 entries, with padding behind
    kern = _CpuKernels()
    ex = ShardedExchange(n_net, n_grid, world, rank, Collectives(dist), 'cpu', torch.bfloat16, kern, units=units)
    p0 = torch.linspace(-1, 1, n_net + n_grid)
    opt = types.SimpleNamespace(p=p0.clone(), exp_avg=torch.zeros(n_net + n_grid), exp_avg_sq=torch.zeros(n_net + n_grid),
                                step_dev=torch.zeros(1, dtype=torch.int32), lr_dev=torch.zeros(1))
    ex.seed_working_copy(p0.to(torch.bfloat16))
    hist, steps, gates, mlp = [], [], [], []
    for s_i, (c0, c1, wrong) in enumerate(_DP_STEPS):
        w16 = _run_exchange(ex, kern, s_i, rank, (c0, c1)[rank], wrong, opt, n_net)
        hist.append(w16); steps.append(int(opt.step_dev)); gates.append(int(ex.eff_gate)); mlp.append(opt.p[:n_net].clone())
    stale = opt.p.clone()
    ex.gather_master(opt.p)
    torch.save({'hist': hist, 'p': opt.p, 'stale': stale, 'steps': steps, 'gates': gates, 'mlp': mlp, 'lo': ex.lo, 'hi': ex.hi,
                'prev_field_max_seen': kern.seen_field_max, 'absmax_seen': kern.seen_absmax, 'units_calls': kern.units_calls,
                'exp_avg': opt.exp_avg[:n_net].clone()}, out + f'.{rank}')
    dist.barrier()
    dist.destroy_process_group()


def _check_sharded_exchange(tmp_path, units):
    out = str(tmp_path / f'ex_{units}.pt')
    mp.spawn(_exchange_worker, args=(2, _free_port(), out, units), nprocs=2, join=True)
    r0, r1 = torch.load(out + '.0'), torch.load(out + '.1')
    # both ranks hold the same working copy after every step, and the same master after gather_master
    for a, b in zip(r0['hist'], r1['hist']):
        assert torch.equal(a, b)
    n_taken = sum(_step_taken(*st) for st in _DP_STEPS)
    # THE JOB-WIDE GATE: whatever went wrong on rank 1 only (local overflow flag, truncated batch, an overflow that only shows
    # in its slice of the SUMMED table), both ranks skip the step: equal gates, step counts, MLP weights and moments
    assert r0['gates'] == r1['gates'] == [1 if _step_taken(*st) else 0 for st in _DP_STEPS]
    assert r0['steps'] == r1['steps'] and r0['steps'][-1] == n_taken
    for a, b in zip(r0['mlp'], r1['mlp']):
        assert torch.equal(a, b)
    assert torch.equal(r0['exp_avg'], r1['exp_avg'])
    assert torch.equal(r0['p'], r1['p'])
    assert (r0['lo'], r0['hi'], r1['lo'], r1['hi']) == (0, 20, 20, 37)
    # before the gather a rank's master of the OTHER slice is stale (never touched), its own slice is current
    n_net = 8
    assert torch.equal(r0['stale'][n_net:n_net + 40], r0['p'][n_net:n_net + 40]) and not torch.equal(r0['stale'][n_net + 40:], r0['p'][n_net + 40:])
    # single-process reference: the same kernels on the summed fields
    kern = _CpuKernels()
    n_grid = 74
    p = torch.linspace(-1, 1, n_net + n_grid)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int32); lr = torch.full((1,), 1e-2)
    w16 = p.to(torch.bfloat16)
    for s_i, (c0, c1, wrong) in enumerate(_DP_STEPS):
        total = sum(_fields_of(s_i, r, n_grid) for r, c in enumerate((c0, c1)) if c > 0) if (c0 + c1) > 0 else torch.zeros(n_grid, dtype=torch.int32)
        g = torch.cat([torch.full((n_net,), float(c0 + c1)), total.float() * 2.0 ** -kern.SHIFT])
        gate = torch.tensor([1 if _step_taken(c0, c1, wrong) else 0])
        if int(gate):
            step += 1
        kern.adam(p, m, v, g, w16, step, lr, gate)
        assert torch.equal(r0['hist'][s_i], w16), s_i             # incl. the skipped steps
    assert torch.equal(r0['p'], p)
    # the statistics of a step reach the NEXT step's units: the largest field of the slices, the largest |dfeat|
    assert int(r0['prev_field_max_seen'][0]) > 0
    n_steps = len(_DP_STEPS)
    if units == 'lagged':
        # one exact exchange (the first step: nothing to lag behind), then one derivation per step from its own statistics,
        # one bit coarser; the last one saw the last step's max |dfeat|
        assert r0['units_calls'] == [0] + [1] * n_steps
        assert float(r0['absmax_seen'][0]) == 0.5 + (n_steps - 1)
    else:
        assert r0['units_calls'] == [0] * n_steps
    return r0


def test_sharded_exchange_world2_equals_the_single_process_step(tmp_path):
    _check_sharded_exchange(tmp_path, 'exact')


def test_sharded_exchange_world2_with_lagged_units(tmp_path):
    _check_sharded_exchange(tmp_path, 'lagged')


def test_level_shard_exchange_volume_for_config5():
    """perf_amd.sharded.exchange_volume: the bytes per link of BASELINE config 5's level-sharded encode (DESIGN.md 5.3 quotes
    them for 8 ranks): levels are split evenly, every rank sends every peer the same order of magnitude, and the totals follow
    (W-1)/W x (12 + 4 L) bytes per sample for 16-bit features."""
    from perf_amd.grid import GridConfig
    from perf_amd.sharded import exchange_volume
    L = 20
    b = float(torch.exp(torch.log(torch.tensor(8192.0 / 16)) / (L - 1)))
    grid = GridConfig(n_levels=L, log2_hashmap_size=30, base_resolution=16, per_level_scale=b)
    n = 1 << 20
    v = exchange_volume(grid, 8, n)
    rows = [len(a) for a in v['levels_per_rank']]
    assert sorted(l for a in v['levels_per_rank'] for l in a) == list(range(L)) and max(rows) - min(rows) <= 1
    total = sum(v['per_rank_bytes_out'])
    assert total == 8 * 7 * 12 * n + 7 * L * n * 4             # positions to 7 peers; every level's features of the other 7 ranks' samples leave their owner
    assert abs(total / 8 / n - (7 * 12 + 7 * 4 * L / 8)) < 1e-9    # bytes a rank sends per sample of its own: 84 (positions) + 70 (features)
    assert v['worst_link_bytes'] <= 12 * n + 3 * n * 4 and max(v['table_GiB_per_rank']) < 6.0
    vt = exchange_volume(grid, 8, n, training=True)
    assert sum(vt['per_rank_bytes_out']) == total + 7 * L * n * 8
