"""Data-parallel gradient exchange of one network (rays shard over the GPUs of a node, the weights are
replicated, ONE gradient exchange per step) -- not in the reference, which is single-GPU (core_exp_runner.py:266).

Design for xGMI (point-to-point links, ring collectives are per-link bound), per training step and per rank:

    MLP backward (local samples)
      [units = 'exact' only: all-gather of PERF_DP_STATS words per rank -> perf_dp_units: the units the single process
       would use; 'lagged' (default): the units were derived at the end of the previous step from ITS statistics]
      -> grid backward into int32 fields (perf_hashgrid_bwd, raw_fields)
      -> reduce-scatter(SUM) of the int32 fields: 26.6 MB in, 26.6/W MB out per rank; integer sums are exact and order
         independent (exact units: the summed table equals the single-process table BIT FOR BIT); the step's
         parameter-independent work (the geometry step's colour render, the next batch draw) runs beside it
      -> perf_fixed_unfix on THIS rank's slice
      -> ONE small all-reduce: [MLP weight gradient | per-rank slots: max |dfeat|, largest summed fields, live samples,
         overflow flag, truncation bit]  -> the job-wide step gate (identical on every rank) and the next step's units
      -> Adam on the MLP part and on this rank's 1/W slice of the table (fp32 master, both moments: sharded state)
      -> all-gather of the 16-bit working copy of the slices (13.3 MB in total) -- what the next forward reads.

Against the plain all-reduce of the fp32 gradient (2 x 26.6 MB over the ring) this moves 26.6 + 13.3 MB and runs Adam on
1/W of the parameters.  The fp32 master of the other ranks' slices goes stale on a rank; `gather_master()` refreshes it
(checkpoints, end of an episode).  All buffers are allocated once: the step is a fixed launch sequence, capturable in a
hipGraph together with its collectives.

The compute steps are injected (`kernels`), so the choreography -- slices, padding, gates, buffer reuse -- is tested on CPU
with world_size-2 gloo (tests/test_cpu_dist.py) and on one GPU shared by two ranks (tests/test_gpu_dist.py).
"""
import torch

from . import _lib

MAX_LEVELS = _lib.MAX_LEVELS
DP_STATS = _lib.DP_STATS
DP_SLOT = _lib.DP_SLOT


def slice_bounds(n_entries, world, rank):
    """Table entries [lo, hi) owned by `rank`: equal slices of an even number of entries (16-byte aligned int32 pairs);
    the last slices may be shorter or empty."""
    per = -(-n_entries // world)
    per += per & 1
    lo = min(rank * per, n_entries)
    return lo, min(lo + per, n_entries), per


class Collectives:
    """The four collectives of a step over torch.distributed (backend 'nccl' = RCCL on ROCm; 'gloo' in tests)."""

    def __init__(self, dist, group=None):
        self.dist = dist
        self.group = group
        self._rs_ok = True

    def all_gather(self, out, inp, async_op=False):
        return self.dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)

    def all_reduce(self, t, async_op=False):
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def reduce_scatter(self, out, inp, rank, async_op=False):
        if self._rs_ok:
            try:
                return self.dist.reduce_scatter_tensor(out, inp, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            except (RuntimeError, NotImplementedError):
                if self.dist.get_backend(self.group) == 'nccl':
                    raise
                self._rs_ok = False             # (a test backend without reduce-scatter on this device: all-reduce + slice)
        self.dist.all_reduce(inp, op=self.dist.ReduceOp.SUM, group=self.group)
        out.copy_(inp[rank * out.numel():(rank + 1) * out.numel()])
        return None


def _wait(work):
    if work is not None:
        work.wait()


class ShardedExchange:
    """Buffers and choreography of the sharded step of ONE network with a flat parameter vector [MLP | table].

    units = 'lagged' (default): the fixed-point units of step t come from the statistics of step t-1, which travelled in
    the slots of THAT step's small all-reduce -- three collectives per step (reduce-scatter, small all-reduce, all-gather),
    none of them between the MLP backward and the grid backward; the units are one bit coarser than the exact ones (room for
    the step-to-step growth of max |dfeat| the lag cannot see).  The first step of an exchange has no previous statistics
    and takes the exact path.  units = 'exact': a statistics all-gather before the grid backward -- the units the single
    process would use, summed table equal to the single process's BIT FOR BIT (tests), four collectives.

    The step gate is JOB-WIDE in both modes: every rank's overflow flag (its grid backward OR its slice of the summed
    table) and truncation bit ride in its slot of the small all-reduce, which runs AFTER the reduce-scatter and the
    conversion of the slice; perf_step_bookkeeping then sees the same {samples, overflow, truncated} on every rank.

    kernels: an object with
        stats_pack(level_absmax, field_max_prev_or_None, n_dev, n, out)
        units(stats_all, world, shifts_out, n_total_out, margin_bits)  (applies the headroom feedback)
        grid_bwd_raw(x01, dfeat, payload_view, n_dev, shifts)         (int32 fields into payload_view)
        unfix(shard, lo, hi, shifts, field_max_out, flag)             (in place int32 -> fp32)
        slot_pack(level_absmax, field_max, n_dev, n, flag, n_marched, capacity, rank, world, slots_out)
        slot_unpack(slots, world, stats_all_or_None, job_flags_out, n_total_out)
        bookkeeping(step_dev, gate, counters, n_marched, n_kept, capacity, overflow, remote_flags, eff_gate)
        adam(p, m, v, g, w16, step_dev, lr_dev, gate)                 (on slices)
        overflow_flag()                                               -> device int32 [1] the grid backward ORs into
    """
    LAG_MARGIN_BITS = 1

    def __init__(self, n_net, n_grid, world, rank, coll, device, w16_dtype, kernels, units='lagged'):
        assert n_grid % 2 == 0 and units in ('lagged', 'exact')
        self.n_net, self.n_grid, self.world, self.rank = n_net, n_grid, world, rank
        self.coll, self.k, self.units = coll, kernels, units
        self.lo, self.hi, self.per = slice_bounds(n_grid // 2, world, rank)
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=device)
        self.payload = z(2 * self.per * world, torch.int32)        # the whole table's fields (+ zero padding behind n_grid)
        self.shard = z(2 * self.per, torch.int32)                  # this rank's slice after the reduce-scatter
        self.stats_local = z(DP_STATS, torch.int32)
        self.stats_all = z(DP_STATS * world, torch.int32)
        self.shifts = z(MAX_LEVELS, torch.int32)
        self.n_total = z(1, torch.int64)
        self.eff_gate = z(1, torch.int64)
        self.field_max = z(MAX_LEVELS, torch.int32)
        self.level_absmax = z(MAX_LEVELS, torch.float32)           # of the step in flight (exchange_units keeps it for the slot)
        self.job_flags = z(2, torch.float32)                       # {overflow, truncated} summed over the ranks
        self.have_prev = False                                     # a previous step left field maxima (exact mode) ...
        self.have_units = False                                    # ... and units for the next one (lagged mode)
        self._live = (None, 0)
        self.ar = z(n_net + DP_SLOT * world, torch.float32)        # [MLP weight gradient | one slot per rank]
        self.w16_slice = z(2 * self.per, w16_dtype)
        self.w16_full = z(n_net + 2 * self.per * world, w16_dtype)  # [MLP | table (+ padding)]: the network's working copy
        self.timing = None                                         # {name: [(event, event)]} while bench.py times the collectives

    # ---- views ---------------------------------------------------------------------------------------------
    @property
    def dw(self):
        """The summed MLP weight gradient of the last step (first n_net words of the small all-reduce)."""
        return self.ar[:self.n_net]

    def grid_payload_f32(self):
        """The first n_grid words of the payload, as the fp32 view perf_hashgrid_bwd's signature asks for."""
        return self.payload[:self.n_grid].view(torch.float32)

    def own(self, flat):
        """This rank's slice of a flat [MLP | table] vector."""
        return flat[self.n_net + 2 * self.lo:self.n_net + 2 * self.hi]

    def _timed(self, name, fn):
        """Run a collective; under `timing` synchronously between two events on the current stream (eager passes only)."""
        if self.timing is None:
            return fn(False)
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        fn(True)
        b.record()
        self.timing.setdefault(name, []).append((a, b))
        return None

    # ---- the step --------------------------------------------------------------------------------------------
    def exchange_units(self, level_absmax, n_dev, n, overlap=None):
        """The units of this step's grid backward (self.shifts).  Lagged mode with a previous step: they are there already
        (derived at the end of that step), nothing is exchanged.  Otherwise: statistics all-gather -> job-wide units."""
        self.level_absmax.copy_(level_absmax)
        self._live = (n_dev, n)                 # this step's live-sample count: travels in the slot (reduce_and_step)
        if self.units == 'lagged' and self.have_units:
            if overlap is not None:
                overlap()
            return self.shifts, self.n_total
        self.k.stats_pack(level_absmax, self.field_max if self.have_prev else None, n_dev, n, self.stats_local)
        work = self._timed('all_gather_stats', lambda sync: self.coll.all_gather(self.stats_all, self.stats_local,
                                                                                async_op=overlap is not None and not sync))
        if overlap is not None:
            overlap()
        _wait(work)
        self.k.units(self.stats_all, self.world, self.shifts, self.n_total, 0)
        return self.shifts, self.n_total

    def reduce_and_step(self, dw, opt, counters=None, n_marched=None, n_kept=None, capacity=0, overlap=None):
        """After the grid backward filled the payload: reduce-scatter of the fields (with `overlap` running beside it) ->
        conversion of this rank's slice -> small all-reduce [MLP weight gradient | the ranks' slots] -> job-wide gate ->
        Adam on the MLP part (replicated) and on this rank's table slice -> all-gather of the refreshed 16-bit slices.
        opt: an object with p, exp_avg, exp_avg_sq (flat fp32 vectors), step_dev, lr_dev.  Returns the full 16-bit working
        copy [MLP | table]."""
        n_net = self.n_net
        flag = self.k.overflow_flag()
        work = self._timed('reduce_scatter', lambda sync: self.coll.reduce_scatter(self.shard, self.payload, self.rank,
                                                                                 async_op=overlap is not None and not sync))
        if overlap is not None:
            overlap()
        _wait(work)
        self.k.unfix(self.shard, self.lo, self.hi, self.shifts, self.field_max, flag)      # (ORs the slice's flag into `flag`)
        self.have_prev = True
        self.ar[:n_net].copy_(dw)
        n_dev, n = self._live
        self.k.slot_pack(self.level_absmax, self.field_max, n_dev, n, flag, n_marched, capacity, self.rank, self.world, self.ar[n_net:])
        self._timed('all_reduce_small', lambda sync: self.coll.all_reduce(self.ar))
        lagged = self.units == 'lagged'
        self.k.slot_unpack(self.ar[n_net:], self.world, self.stats_all if lagged else None, self.job_flags, self.n_total)
        self.k.bookkeeping(opt.step_dev, self.n_total, counters, n_marched, n_kept, capacity, flag, self.job_flags, self.eff_gate)
        # MLP weights: every rank holds the same summed gradient and takes the same step
        self.k.adam(opt.p[:n_net], opt.exp_avg[:n_net], opt.exp_avg_sq[:n_net], self.ar[:n_net], self.w16_full[:n_net],
                    opt.step_dev, opt.lr_dev, self.eff_gate)
        n_own = 2 * (self.hi - self.lo)
        if n_own > 0:
            self.k.adam(self.own(opt.p), self.own(opt.exp_avg), self.own(opt.exp_avg_sq), self.shard.view(torch.float32)[:n_own],
                        self.w16_slice[:n_own], opt.step_dev, opt.lr_dev, self.eff_gate)
        self._timed('all_gather_w16', lambda sync: self.coll.all_gather(self.w16_full[n_net:], self.w16_slice))
        if lagged:
            # the units of the NEXT step from THIS step's statistics (max |dfeat|, sample count, largest summed fields)
            self.k.units(self.stats_all, self.world, self.shifts, None, self.LAG_MARGIN_BITS)
            self.have_units = True
        return self.w16_full[:n_net + self.n_grid]

    def seed_working_copy(self, w16):
        """Before the first step: the gate may skip a step, and a skipped step must leave a valid working copy behind."""
        self.w16_full[:self.n_net + self.n_grid].copy_(w16)
        n_own = 2 * (self.hi - self.lo)
        if n_own > 0:
            self.w16_slice[:n_own].copy_(self.own(w16))

    def gather_master(self, p):
        """Refresh the fp32 master `p` ([MLP | table]) of the slices owned by other ranks (one fp32 all-gather)."""
        full = torch.empty(2 * self.per * self.world, dtype=p.dtype, device=p.device)
        mine = torch.zeros(2 * self.per, dtype=p.dtype, device=p.device)
        n_own = 2 * (self.hi - self.lo)
        if n_own > 0:
            mine[:n_own].copy_(self.own(p))
        self.coll.all_gather(full, mine)
        p[self.n_net:].copy_(full[:self.n_grid])
        return p
