"""Tensor-level wrappers over the C ABI (include/perf_hip.h).

Every function takes contiguous CUDA tensors, allocates outputs/workspaces with torch's caching
allocator, and enqueues on torch's current stream.  Nothing here has a CPU fallback: a CPU tensor
or a missing libperf_hip.so raises.
"""
import ctypes
import math
import os
from typing import NamedTuple

import torch

from . import _lib
from ._lib import DTYPE_BF16, DTYPE_FP16
from .grid import GridConfig, MlpConfig

_T16 = {DTYPE_BF16: torch.bfloat16, DTYPE_FP16: torch.float16}
_CODE = {torch.bfloat16: DTYPE_BF16, torch.float16: DTYPE_FP16, 'bf16': DTYPE_BF16, 'fp16': DTYPE_FP16,
         DTYPE_BF16: DTYPE_BF16, DTYPE_FP16: DTYPE_FP16}


# ---- optional per-kernel timing with HIP events on the launch stream (used by bench.py) ------------------
_PROF = None


def start_kernel_timing():
    """Record a HIP event pair around every C-ABI launch until stop_kernel_timing()."""
    global _PROF
    _PROF = {}


def stop_kernel_timing():
    """-> {entry point: (n_launches, mean milliseconds)}; synchronises the device."""
    global _PROF
    prof, _PROF = _PROF, None
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in (prof or {}).items() if v}


def _call(name, *args, label=None):
    """label: the name the launch is timed under when it differs from the entry point (one entry point, two kinds of launch)."""
    if _PROF is None:
        return _lib.call(name, *args)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.call(name, *args)
    b.record()
    _PROF.setdefault(label or name, []).append((a, b))


def dtype_code(d) -> int:
    return _CODE[d]


def torch_dtype(d):
    return _T16[dtype_code(d)]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.PerfError('perf_amd ops need CUDA (HIP) tensors; there is no CPU path')
    if not t.is_contiguous():
        raise _lib.PerfError('perf_amd ops need contiguous tensors')
    return ctypes.c_void_p(t.data_ptr())


def _nd(n_dev):
    """Device-side sample count (int64 tensor with one element, e.g. the `total` of exclusive_scan_i32) or None."""
    if n_dev is None:
        return None
    if n_dev.dtype != torch.int64 or not n_dev.is_cuda:
        raise _lib.PerfError('n_dev must be an int64 CUDA tensor')
    return ctypes.c_void_p(n_dev.data_ptr())


def _f32(t, name):
    if t.dtype != torch.float32:
        raise _lib.PerfError(f'{name} must be float32, got {t.dtype}')
    return t


def _aabb6(aabb):
    vals = [float(v) for v in (aabb.detach().cpu().tolist() if torch.is_tensor(aabb) else aabb)]
    if len(vals) != 6:
        raise _lib.PerfError('aabb must hold 6 values')
    return (ctypes.c_float * 6)(*vals)


# ---- parameters ----------------------------------------------------------------------------------
def cast_params(src: torch.Tensor, dtype, out: torch.Tensor = None) -> torch.Tensor:
    _f32(src, 'params')
    code = dtype_code(dtype)
    if out is None:
        out = torch.empty(src.numel(), dtype=_T16[code], device=src.device)
    _call('perf_cast_params', _p(src), _p(out), src.numel(), code, _stream())
    return out


def adam_step(p, m, v, g, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, w16=None, zero_grad=True):
    code = dtype_code(w16.dtype) if w16 is not None else 0
    _call('perf_adam_step', _p(_f32(p, 'p')), _p(_f32(m, 'm')), _p(_f32(v, 'v')), _p(_f32(g, 'g')), _p(w16),
              p.numel(), code, int(step), float(lr), float(beta1), float(beta2), float(eps), int(bool(zero_grad)),
              _stream())


def adam_step_dev(p, m, v, g, step_dev, lr_dev, beta1=0.9, beta2=0.999, eps=1e-8, w16=None, zero_grad=False, gate=None, clear_flag=None):
    """gate: device int64 [1]; the update is skipped when it holds 0 (a batch without samples).  clear_flag (device int32 [1]): zeroed by
    the launch, taken or not -- how the overflow flag is consumed when the bookkeeping rode in the repair launch (field_bwd(book=...))."""
    code = dtype_code(w16.dtype) if w16 is not None else 0
    _call('perf_adam_step_dev', _p(_f32(p, 'p')), _p(_f32(m, 'm')), _p(_f32(v, 'v')), _p(_f32(g, 'g')), _p(w16),
          p.numel(), code, _p(step_dev), _p(lr_dev), _nd(gate), float(beta1), float(beta2), float(eps), int(bool(zero_grad)),
          _p(clear_flag), _stream())


class StepBook:
    """The arguments of step_bookkeeping, kept for a call that carries the bookkeeping in one of its own launches (field_bwd(book=...):
    perf_field_bwd_book).  done: set by the call that did it -- the optimizer then skips its own bookkeeping launch and lets Adam clear the
    overflow flag."""

    def __init__(self, step_dev=None, gate=None, counters=None, n_marched=None, n_kept=None, capacity=0, overflow=None, remote_flags=None,
                 eff_gate=None, schedule=None, overflow_redone=False):
        self.args = dict(step_dev=step_dev, gate=gate, counters=counters, n_marched=n_marched, n_kept=n_kept, capacity=capacity, overflow=overflow,
                         remote_flags=remote_flags, eff_gate=eff_gate, schedule=schedule, overflow_redone=overflow_redone)
        self.done = False

    def struct(self):
        a = self.args
        table, it, lr_out, ratio_out = a['schedule'] if a['schedule'] is not None else (None, None, None, None)
        for name in ('gate', 'counters', 'n_marched', 'n_kept', 'eff_gate'):
            _nd(a[name])                                   # (dtype / device checks of the int64 scalars)
        q = lambda t: None if t is None else _p(t).value
        return _lib.StepBook(q(a['step_dev']), q(a['gate']), q(a['counters']), q(a['n_marched']), q(a['n_kept']), int(a['capacity'] or 0),
                             q(a['overflow']), q(a['remote_flags']), q(a['eff_gate']), q(table), q(it), q(lr_out), q(ratio_out),
                             int(table.shape[0]) if table is not None else 0, int(bool(a['overflow_redone'])))

    def launch(self):
        """The stand-alone launch (perf_step_bookkeeping) with the same arguments."""
        step_bookkeeping(**self.args)


def step_bookkeeping(step_dev=None, gate=None, counters=None, n_marched=None, n_kept=None, capacity=0, overflow=None,
                     remote_flags=None, eff_gate=None, schedule=None, overflow_redone=False):
    """One launch (perf_step_bookkeeping): decides whether the optimizer step is TAKEN (samples present, no fixed-point
    overflow flag -- local int32 `overflow` or `remote_flags` (float32 [2] = {overflow, truncated} summed over the ranks of a
    data-parallel job, dp_slot_unpack) --, batch not truncated at `capacity`), advances step_dev and writes eff_gate
    (int64 [1]) accordingly, accumulates counters (int64 [8]).
    schedule = (table f32 [n, 2] = rows of (lr, distortion ramp), iter_dev int32 [1], lr_out f32 [1], ratio_out f32 [1] or None):
    the device-side schedule of a graph-replayed phase (see perf_step_bookkeeping).  overflow_redone: the flagged gradient was
    repaired in place (hashgrid_bwd_redo): count the event, take the step."""
    ref = next(t for t in (step_dev, counters, eff_gate) if t is not None)
    if not ref.is_cuda:
        raise _lib.PerfError('perf_amd ops need CUDA (HIP) tensors; there is no CPU path')
    table, it, lr_out, ratio_out = schedule if schedule is not None else (None, None, None, None)
    _call('perf_step_bookkeeping', _p(step_dev), _nd(gate), _nd(counters), _nd(n_marched), _nd(n_kept), int(capacity or 0),
          _p(overflow), _p(remote_flags), int(bool(overflow_redone)), _nd(eff_gate), _p(table), int(table.shape[0]) if table is not None else 0, _p(it),
          _p(lr_out), _p(ratio_out), _stream())


def step_counters(device):
    """Zeroed int64 [8] block of perf_step_bookkeeping: {marched, kept, steps, max marched of one batch, steps skipped for
    fixed-point overflow, steps skipped for truncation, 0, 0}."""
    return torch.zeros(_lib.STEP_COUNTERS, dtype=torch.int64, device=device)


# ---- positions -----------------------------------------------------------------------------------
def points_from_rays(rays_o, rays_d, ray_indices, t_starts, t_ends, aabb, n_dev=None):
    n = ray_indices.numel()
    if ray_indices.dtype != torch.int64:
        raise _lib.PerfError('ray_indices must be int64')
    x01 = torch.empty(n, 3, dtype=torch.float32, device=rays_o.device)
    sel = torch.empty(n, dtype=torch.uint8, device=rays_o.device)
    _call('perf_points_from_rays', _p(_f32(rays_o, 'rays_o')), _p(_f32(rays_d, 'rays_d')), _p(ray_indices),
              _p(_f32(t_starts, 't_starts')), _p(_f32(t_ends, 't_ends')), _aabb6(aabb), _p(x01), _p(sel), n, _nd(n_dev), _stream())
    return x01, sel


def points_normalize(x, aabb):
    x = _f32(x, 'x').reshape(-1, 3)
    n = x.shape[0]
    x01 = torch.empty(n, 3, dtype=torch.float32, device=x.device)
    sel = torch.empty(n, dtype=torch.uint8, device=x.device)
    _call('perf_points_normalize', _p(x), _aabb6(aabb), _p(x01), _p(sel), n, _stream())
    return x01, sel


# ---- hash grid -----------------------------------------------------------------------------------
def hashgrid_fwd(grid: GridConfig, x01, table16, n_dev=None):
    """x01 [n,3] f32, table16 [total*2] 16-bit -> feat [L, n, 2] 16-bit (level major).  n_dev (device int64 [1]): only the
    first min(n, n_dev) samples are encoded (n = capacity = level stride)."""
    n = x01.shape[0]
    feat = torch.empty(grid.n_levels, n, 2, dtype=table16.dtype, device=x01.device)
    d = grid.desc()
    _call('perf_hashgrid_fwd', ctypes.byref(d), _p(_f32(x01, 'x01')), _p(table16), _p(feat), n, _nd(n_dev),
              dtype_code(table16.dtype), _stream())
    return feat


def hashgrid_fwd2(grid: GridConfig, x01, table16_a, table16_b):
    """Both tables (same geometry) at the same points -> (feat_a, feat_b), each [L, n, 2] 16-bit."""
    n = x01.shape[0]
    fa = torch.empty(grid.n_levels, n, 2, dtype=table16_a.dtype, device=x01.device)
    fb = torch.empty(grid.n_levels, n, 2, dtype=table16_a.dtype, device=x01.device)
    d = grid.desc()
    _call('perf_hashgrid_fwd2', ctypes.byref(d), _p(_f32(x01, 'x01')), _p(table16_a), _p(table16_b), _p(fa), _p(fb), n,
          dtype_code(table16_a.dtype), _stream())
    return fa, fb


def hashgrid_fwd_f32(grid: GridConfig, x01, table):
    n = x01.shape[0]
    feat = torch.empty(grid.n_levels, n, 2, dtype=torch.float32, device=x01.device)
    d = grid.desc()
    _call('perf_hashgrid_fwd_f32', ctypes.byref(d), _p(_f32(x01, 'x01')), _p(_f32(table, 'table')), _p(feat), n, _stream())
    return feat


_OVERFLOW_FLAG = {}


def overflow_flag(device):
    """Device int32 that the fixed-point grid backward ORs with 1 when a field nears the int32 range."""
    key = str(device)
    if key not in _OVERFLOW_FLAG:
        _OVERFLOW_FLAG[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _OVERFLOW_FLAG[key]


def headroom_state(device):
    """Zeroed state block of the fixed-point headroom feedback (one per table that is trained, see perf_hashgrid_bwd)."""
    return torch.zeros(_lib.HEADROOM_STATE_WORDS, dtype=torch.int32, device=device)


def hashgrid_bwd(grid: GridConfig, x01, dfeat, out=None, accumulate=False, level_absmax=None, n_dev=None, use_codes=True,
                 hr_state=None, shifts=None, raw_fields=False):
    """dfeat [L, n, 2] f32 -> gradient table [total*2] f32.  `out` (a contiguous fp32 view, e.g. the grid
    part of a flat gradient) is overwritten, or added to when accumulate=True.  level_absmax (device, 16 floats
    from mlp_bwd) selects the packed fixed-point accumulation.  shifts (device int32 [24]): job-wide units of a
    data-parallel step (dp_units) instead of level_absmax / hr_state; raw_fields: `out` receives the int32 field pairs
    (same storage; view it as int32) for an integer reduce-scatter followed by fixed_unfix."""
    n = x01.shape[0]
    if out is None:
        out = torch.empty(grid.n_params, dtype=torch.float32, device=x01.device)
        accumulate = False
    d = grid.desc()
    fixed = level_absmax is not None or shifts is not None
    # (a workspace without room for the tile codes selects the position-streaming owners: use_codes=False, tests)
    ws_bytes = _lib.load().perf_hashgrid_bwd_workspace_bytes(ctypes.byref(d), n if use_codes else 0)
    ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=x01.device)
    flag = overflow_flag(x01.device) if fixed else None
    _call('perf_hashgrid_bwd', ctypes.byref(d), _p(_f32(x01, 'x01')), _p(_f32(dfeat, 'dfeat')), _p(_f32(out, 'grad')),
              n, _nd(n_dev), int(bool(accumulate)), _p(level_absmax), _p(flag),
              _p(hr_state if (level_absmax is not None and shifts is None) else None), _p(shifts), int(bool(raw_fields)),
              None, _p(ws), ws_bytes, _stream())
    return out


def hashgrid_bwd_redo_supported(grid: GridConfig) -> bool:
    """Grids whose levels all fit LDS owners (<= 255 hashed / 64 dense tiles of 16,384 entries): PeRF's L16/T18 does."""
    for l in range(grid.n_levels):
        tiles = -(-int(grid.size[l]) // 16384)
        if tiles > (255 if grid.hashed[l] else 64):
            return False
    return True


def hashgrid_bwd_redo(grid: GridConfig, x01, dfeat, out, n_dev=None, hr_state=None):
    """The repair launch of a fixed-point hashgrid_bwd into the same `out`: a no-op dispatch unless that call raised the
    overflow flag, else the table gradient again with fp32 LDS accumulation (perf_hashgrid_bwd, redo_flag)."""
    d = grid.desc()
    _call('perf_hashgrid_bwd', ctypes.byref(d), _p(_f32(x01, 'x01')), _p(_f32(dfeat, 'dfeat')), _p(_f32(out, 'grad')),
          x01.shape[0], _nd(n_dev), 0, None, None, _p(hr_state), None, 0, _p(overflow_flag(x01.device)), None, 0, _stream(),
          label='perf_hashgrid_bwd(redo: predicated no-op)')
    return out


_FIELD_BWD_WS = {}
FIELD_BWD_ONE_CALL = True       # False: the three entry points one by one (tools/shim_step_profile.py --three-calls: the A/B of the host cost)


def field_bwd(grid: GridConfig, mlp: MlpConfig, x01, w16_net, feat16, dout, sel=None, fixed=True, redo=True, hr_state=None, n_dev=None,
              grad=None, extra=0, book=None):
    """The whole backward of one field in ONE boundary call (perf_field_bwd: MLP backward -> grid backward -> predicated fp32 repair)
    -> flat fp32 gradient [network | grid (+ `extra` trailing slots)].  feat16: [L, n, 2] or an IndexedFeat.  The workspace (MLP
    partials, tile codes, dfeat) is cached per (device, grid, network) and grown on demand: consecutive backwards on one stream reuse
    it -- do not overlap two backwards of the same field on different streams."""
    index, stride = None, 0
    if isinstance(feat16, IndexedFeat):
        feat16, index = feat16.feat, feat16.index
        stride = feat16.shape[1]
    n = index.shape[0] if index is not None else feat16.shape[1]
    dev = x01.device
    n_all = mlp.n_params + grid.n_params
    if grad is None:
        grad = torch.empty(n_all + extra, dtype=torch.float32, device=dev)
    if _PROF is not None or not FIELD_BWD_ONE_CALL:
        # per-launch timing (start_kernel_timing: bench.py's `kernels` / `roofline` blocks): the same three entry points one by one,
        # so that each is bracketed by its own pair of events
        res = mlp_bwd(mlp, w16_net, IndexedFeat(feat16, index) if index is not None else feat16, dout, sel, want_absmax=fixed, n_dev=n_dev,
                      dw_out=grad[:mlp.n_params])
        hashgrid_bwd_into(grid, x01, res[0], grad[mlp.n_params:n_all], level_absmax=res[2] if fixed else None, n_dev=n_dev,
                          hr_state=hr_state if fixed else None)
        if fixed and redo:
            hashgrid_bwd_redo(grid, x01, res[0], grad[mlp.n_params:n_all], n_dev=n_dev, hr_state=hr_state)
        return grad
    gd, md = grid.desc(), mlp.desc()
    nbytes = _lib.load().perf_field_bwd_workspace_bytes(ctypes.byref(gd), ctypes.byref(md), n, None, None, None)
    if nbytes < 0:
        raise _lib.PerfError('perf_field_bwd_workspace_bytes: bad arguments')
    # ONE workspace per (device, field), grown geometrically: the operator-shim path calls with a different exact n every step -- a
    # buffer per size would pin a pool of differently sized blocks and push the caching allocator into hipMalloc (measured: +0.3 ms
    # per step); any buffer that is large enough serves (the library lays it out from n)
    key = (str(dev), id(gd), mlp.n_levels, mlp.n_hidden_layers, mlp.n_output_dims)
    ws = _FIELD_BWD_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        grown = max(nbytes, int(1.5 * ws.numel() * 4) if ws is not None else 0)
        ws = torch.empty(grown // 4 + 4, dtype=torch.float32, device=dev)
        if not torch.cuda.is_current_stream_capturing():       # (a block first made INSIDE a capture belongs to that graph's pool: not kept)
            if len(_FIELD_BWD_WS) >= 16 and key not in _FIELD_BWD_WS:
                _FIELD_BWD_WS.pop(next(iter(_FIELD_BWD_WS)))
            _FIELD_BWD_WS[key] = ws
    if book is not None and fixed and redo and hr_state is not None and n > 0 and book.args['overflow'] is not None \
            and book.args['overflow'].data_ptr() == overflow_flag(dev).data_ptr():
        # book (a StepBook): the step's bookkeeping rides in the repair launch (perf_field_bwd_book) -- one launch per step fewer; the flag
        # stays for adam_step_dev(clear_flag=...)
        sb = book.struct()
        _call('perf_field_bwd_book', ctypes.byref(gd), ctypes.byref(md), _p(_f32(x01, 'x01')), _p(w16_net), _p(feat16), _p(index), stride, _p(sel),
              _p(_f32(dout, 'dout')), _p(grad), _p(overflow_flag(dev)), _p(hr_state), _p(ws), ws.numel() * 4, n, _nd(n_dev),
              dtype_code(w16_net.dtype), ctypes.byref(sb), _stream())
        book.done = True
        return grad
    _call('perf_field_bwd', ctypes.byref(gd), ctypes.byref(md), _p(_f32(x01, 'x01')), _p(w16_net), _p(feat16), _p(index), stride, _p(sel),
          _p(_f32(dout, 'dout')), _p(grad), int(bool(fixed)), int(bool(fixed and redo)), _p(overflow_flag(dev)) if fixed else None,
          _p(hr_state) if fixed else None, _p(ws), ws.numel() * 4, n, _nd(n_dev), dtype_code(w16_net.dtype), _stream())
    return grad


def hashgrid_bwd_into(grid, x01, dfeat, out, level_absmax=None, n_dev=None, hr_state=None, shifts=None, raw_fields=False):
    return hashgrid_bwd(grid, x01, dfeat, out=out, accumulate=False, level_absmax=level_absmax, n_dev=n_dev, hr_state=hr_state,
                        shifts=shifts, raw_fields=raw_fields)


# ---- job-wide fixed-point units of a data-parallel step -------------------------------------------------------------------
def dp_stats_pack(level_absmax, field_max_prev, n_dev, n, out=None):
    """-> int32 [PERF_DP_STATS]: this rank's block of the statistics every rank all-gathers before its grid backward."""
    if out is None:
        out = torch.empty(_lib.DP_STATS, dtype=torch.int32, device=level_absmax.device)
    _call('perf_dp_stats_pack', _p(level_absmax), _p(field_max_prev), _nd(n_dev), int(n), _p(out), _stream())
    return out


def dp_units(grid: GridConfig, stats_all, world, hr_state, shifts=None, n_total=None, margin_bits=0, want_total=True):
    """stats_all int32 [world, PERF_DP_STATS] -> (shifts int32 [24], n_total int64 [1]); applies the headroom feedback to hr_state.
    margin_bits: make the units that many bits coarser (lagged units of perf_amd/dp.py)."""
    dev = stats_all.device
    if shifts is None:
        shifts = torch.empty(_lib.MAX_LEVELS, dtype=torch.int32, device=dev)
    if n_total is None and want_total:
        n_total = torch.empty(1, dtype=torch.int64, device=dev)
    d = grid.desc()
    _call('perf_dp_units', ctypes.byref(d), _p(stats_all), int(world), _p(hr_state), _p(shifts), _nd(n_total), int(margin_bits), _stream())
    return shifts, n_total


def dp_slot_pack(level_absmax, field_max, n_dev, n, overflow, n_marched, capacity, rank, world, out):
    """This rank's slot of the small all-reduce (perf_dp_slot_pack); `out`: float32 [world * PERF_DP_SLOT], the other slots are zeroed."""
    _call('perf_dp_slot_pack', _p(level_absmax), _p(field_max), _nd(n_dev), int(n), _p(overflow), _nd(n_marched), int(capacity or 0),
          int(rank), int(world), _p(_f32(out, 'slots')), _stream())
    return out


def dp_slot_unpack(slots, world, stats_all=None, job_flags=None, n_total=None):
    """All-reduced slots -> stats_all (int32 [world, PERF_DP_STATS], optional), job_flags (float32 [2] = {overflow, truncated}
    summed over the ranks), n_total (int64 [1])."""
    _call('perf_dp_slot_unpack', _p(_f32(slots, 'slots')), int(world), _p(stats_all), _p(job_flags), _nd(n_total), _stream())
    return job_flags, n_total


def fixed_unfix(grid: GridConfig, fields, entry_lo, entry_hi, shifts, field_max=None, flag=None):
    """In place: int32 field pairs of table entries [entry_lo, entry_hi) (`fields`: a contiguous 4-byte tensor holding them)
    -> fp32 gradients; field_max int32 [24] receives the slice's largest |field| per level."""
    d = grid.desc()
    _call('perf_fixed_unfix', ctypes.byref(d), _p(fields), int(entry_lo), int(entry_hi), _p(shifts), _p(field_max), _p(flag), _stream())
    return fields


def hashgrid_corners(grid: GridConfig, x01):
    """-> int32 [L, n, 8] absolute table entries of the 8 corners (bit0 = x, bit1 = y, bit2 = z)."""
    n = x01.shape[0]
    idx = torch.empty(grid.n_levels, n, 8, dtype=torch.int32, device=x01.device)
    d = grid.desc()
    _call('perf_hashgrid_corners', ctypes.byref(d), _p(_f32(x01, 'x01')), _p(idx), n, _stream())
    return idx


def hashgrid_bwd_input(grid: GridConfig, x01, dfeat, table):
    n = x01.shape[0]
    dx = torch.empty(n, 3, dtype=torch.float32, device=x01.device)
    d = grid.desc()
    _call('perf_hashgrid_bwd_input', ctypes.byref(d), _p(_f32(x01, 'x01')), _p(_f32(dfeat, 'dfeat')),
              _p(_f32(table, 'table')), _p(dx), n, _stream())
    return dx


def hashgrid_bwd_bwd_input(grid: GridConfig, x01, dfeat, table, ggx, want_ddfeat=True, want_dx=True):
    """Backward of hashgrid_bwd_input w.r.t. (dfeat, x01) given ggx = dL/d(dx) [n,3] -> (d_dfeat [L,n,2] | None, d_x [n,3] | None)."""
    n = x01.shape[0]
    dd = torch.empty(grid.n_levels, n, 2, dtype=torch.float32, device=x01.device) if want_ddfeat else None
    dx = torch.empty(n, 3, dtype=torch.float32, device=x01.device) if want_dx else None
    d = grid.desc()
    _call('perf_hashgrid_bwd_bwd_input', ctypes.byref(d), _p(_f32(x01, 'x01')), _p(_f32(dfeat, 'dfeat')), _p(_f32(table, 'table')),
          _p(_f32(ggx, 'ggx')), _p(dd), _p(dx), n, _stream())
    return dd, dx


def hashgrid_bwd_bwd_param(grid: GridConfig, x01, dfeat, ggx):
    """Backward of hashgrid_bwd_input w.r.t. the table given ggx -> grad_table [total*2] f32."""
    n = x01.shape[0]
    out = torch.empty(grid.n_params, dtype=torch.float32, device=x01.device)
    d = grid.desc()
    _call('perf_hashgrid_bwd_bwd_param', ctypes.byref(d), _p(_f32(x01, 'x01')), _p(_f32(dfeat, 'dfeat')), _p(_f32(ggx, 'ggx')), _p(out), n,
          _stream())
    return out


# ---- MLP -----------------------------------------------------------------------------------------
def mlp_fwd(mlp: MlpConfig, w16, feat16, sel=None, n_dev=None):
    n = feat16.shape[1]
    out = torch.empty(n, mlp.n_output_dims, dtype=torch.float32, device=feat16.device)
    d = mlp.desc()
    _call('perf_mlp_fwd', ctypes.byref(d), _p(w16), _p(feat16), _p(sel), _p(out), n, _nd(n_dev), dtype_code(w16.dtype), _stream())
    return out


import os as _os
# perf_field_infer runs encode + MLP as one kernel up to this many rows (16-level grids); the library reads the same switch
FUSED_MAX_SAMPLES = int(_os.environ.get('PERF_FUSED_MAX_SAMPLES', '4096'))


def field_infer(grid: GridConfig, mlp: MlpConfig, x01, sel, w16, n_dev=None, want_features=False):
    """act(MLP(encode(x01))) * sel without gradient in one boundary call; w16 = the network's 16-bit working copy
    [MLP weights | table].  want_features: also return the level-major 16-bit features [L, n, 2] the result was computed from
    (-> (out, feat))."""
    n = x01.shape[0]
    n_net = mlp.n_params
    out = torch.empty(n, mlp.n_output_dims, dtype=torch.float32, device=x01.device)
    feat = torch.empty(grid.n_levels, n, 2, dtype=w16.dtype, device=x01.device) if want_features else None
    fused = n <= FUSED_MAX_SAMPLES and grid.n_levels <= 16
    if _PROF is not None and not fused:
        # per-kernel timing (bench.py): the same two kernels the boundary call launches, issued one by one so that each gets
        # its own event pair
        feat = hashgrid_fwd(grid, x01, w16[n_net:], n_dev=n_dev)
        out = mlp_fwd(mlp, w16[:n_net], feat, sel, n_dev=n_dev)
        return (out, feat) if want_features else out
    scratch = None if (fused or want_features) else torch.empty(grid.n_levels * n, dtype=torch.int32, device=x01.device)
    gd, md = grid.desc(), mlp.desc()
    _call('perf_field_infer', ctypes.byref(gd), ctypes.byref(md), _p(_f32(x01, 'x01')), _p(sel), _p(w16[n_net:]), _p(w16[:n_net]),
          _p(out), n, _nd(n_dev), _p(scratch), scratch.numel() * 4 if scratch is not None else 0, _p(feat), dtype_code(w16.dtype), _stream())
    return (out, feat) if want_features else out


class IndexedFeat(NamedTuple):
    """Level-major features [L, n_src, 2] of MORE samples than the caller means, and the row of each sample it does mean
    (int32 [n]; rows past the live count are not read): what compact_prefix(index_features=True) hands to the gradient pass
    instead of a compacted copy (perf_mlp_bwd's feat_index)."""
    feat: torch.Tensor
    index: torch.Tensor

    def materialize(self):
        """The compacted copy after all ([L, n, 2]; rows past the live count repeat a valid row)."""
        return self.feat.index_select(1, self.index.long().clamp_(0, self.feat.shape[1] - 1))


def mlp_bwd(mlp: MlpConfig, w16, feat16, dout, sel=None, need_dfeat=True, want_absmax=False, n_dev=None, dw_out=None):
    """Returns (dfeat [L,n,2] f32 or None, dw [n_net_params] f32[, level_absmax [16] f32]).  dw_out: write the weight gradient
    there (e.g. the head of a flat [network | grid] gradient) instead of into a fresh tensor.  feat16: [L, n, 2] or an
    IndexedFeat (n = its index's length)."""
    index, stride = None, 0
    if isinstance(feat16, IndexedFeat):
        feat16, index = feat16.feat, feat16.index
        assert index.dtype == torch.int32 and index.is_contiguous() and feat16.is_contiguous()
        stride = feat16.shape[1]
    n = index.shape[0] if index is not None else feat16.shape[1]
    d = mlp.desc()
    lib = _lib.load()
    ws_bytes = lib.perf_mlp_bwd_workspace_bytes(ctypes.byref(d), n)
    ws = torch.empty(max(ws_bytes, 16) // 4, dtype=torch.float32, device=feat16.device)
    dfeat = torch.empty(mlp.n_levels, n, 2, dtype=torch.float32, device=feat16.device) if need_dfeat else None
    dw = dw_out if dw_out is not None else torch.empty(mlp.n_params, dtype=torch.float32, device=feat16.device)
    amax = torch.empty(_lib.MAX_LEVELS, dtype=torch.float32, device=feat16.device) if want_absmax else None
    _call('perf_mlp_bwd', ctypes.byref(d), _p(w16), _p(feat16), _p(index), stride, _p(sel), _p(_f32(dout, 'dout')), _p(dfeat), _p(dw),
          _p(amax), _p(ws), ws.numel() * 4, n, _nd(n_dev), dtype_code(w16.dtype), _stream())
    return (dfeat, dw, amax) if want_absmax else (dfeat, dw)


# ---- rays ----------------------------------------------------------------------------------------
def pano_raygen(pose, height, width, row0=0, nrows=None, device='cuda'):
    nrows = height - row0 if nrows is None else nrows
    pose_h = (ctypes.c_float * 16)(*[float(v) for v in torch.as_tensor(pose).detach().cpu().reshape(-1).tolist()])
    o = torch.empty(nrows, width, 3, dtype=torch.float32, device=device)
    d = torch.empty(nrows, width, 3, dtype=torch.float32, device=device)
    _call('perf_pano_raygen', pose_h, height, width, row0, nrows, _p(o), _p(d), _stream())
    return o, d


def pano_raygen_dev(pose_dev, height, width, row0=0, nrows=None, out=None):
    """pano_raygen with the pose (float32 [4,4] or [12+]) in device memory; `out` = (o, d) preallocated [nrows*width,3] buffers
    (a captured hipGraph writes the same buffers on every replay)."""
    nrows = height - row0 if nrows is None else nrows
    dev = pose_dev.device
    if out is None:
        o = torch.empty(nrows, width, 3, dtype=torch.float32, device=dev)
        d = torch.empty(nrows, width, 3, dtype=torch.float32, device=dev)
    else:
        o, d = out
    _call('perf_pano_raygen_dev', _p(_f32(pose_dev, 'pose')), height, width, row0, nrows, _p(o), _p(d), _stream())
    return o, d


# ---- occupancy marching ----------------------------------------------------------------------------
def occ_pack_bits(binaries: torch.Tensor) -> torch.Tensor:
    b = binaries.reshape(-1)
    if b.dtype == torch.bool:
        b = b.view(torch.uint8)
    n = b.numel()
    bits = torch.empty((n + 31) // 32, dtype=torch.int32, device=b.device)
    _call('perf_occ_pack_bits', _p(b), _p(bits), n, _stream())
    return bits


def occ_jitter_points(seed, call, cell_lo, n, res, aabb, device, out=None):
    """Jittered evaluation points of cells [cell_lo, cell_lo + n) of a res^3 grid -> x [n, 3] (perf_occ_jitter_points)."""
    x = out if out is not None else torch.empty(n, 3, dtype=torch.float32, device=device)
    _call('perf_occ_jitter_points', int(seed) & 0xFFFFFFFFFFFFFFFF, int(call), int(cell_lo), int(n), int(res), _aabb6(aabb), _p(x), _stream())
    return x


def occ_ema_update(occs, occ, ema_decay, sum_out):
    """In place occs = max(occs * ema_decay, occ); sum_out (device float64 [1]) += the new values."""
    if sum_out.dtype != torch.float64:
        raise _lib.PerfError('sum_out must be float64')
    _call('perf_occ_ema_update', _p(_f32(occs, 'occs')), _p(_f32(occ, 'occ')), occs.numel(), float(ema_decay), _p(sum_out), _stream())


def occ_threshold(occs, sum_dev, occ_thre, out=None):
    """-> bool [n]: occs > min(sum / n, occ_thre)."""
    b = out if out is not None else torch.empty(occs.numel(), dtype=torch.bool, device=occs.device)
    _call('perf_occ_threshold', _p(_f32(occs, 'occs')), occs.numel(), _p(sum_dev), float(occ_thre), _p(b.view(torch.uint8)), _stream())
    return b


def exclusive_scan_i32(counts: torch.Tensor, bias=None):
    """-> (offsets, total int64 [1]); with `bias` (host int) also total + bias as a third result (same launch)."""
    n = counts.numel()
    lib = _lib.load()
    out = torch.empty_like(counts)
    totals = torch.empty(2, dtype=torch.int64, device=counts.device)
    ws = torch.empty(lib.perf_scan_workspace_bytes(n) // 8 + 1, dtype=torch.int64, device=counts.device)
    _call('perf_exclusive_scan_i32', _p(counts), _p(out), _p(totals), n, int(bias or 0), _p(totals[1:]) if bias is not None else None,
          _p(ws), ws.numel() * 8, _stream())
    return (out, totals[:1]) if bias is None else (out, totals[:1], totals[1:])


def occ_build_coarse(occ_bits, res):
    """Dilated 4^3-block occupancy for the marching kernel's empty-space skip (None when res % 8 != 0)."""
    words = _lib.load().perf_occ_coarse_words(int(res))
    if words == 0:
        return None
    coarse = torch.empty(words, dtype=torch.int32, device=occ_bits.device)
    _call('perf_occ_build_coarse', _p(occ_bits), int(res), _p(coarse), _stream())
    return coarse


def _origin(t0):
    """Lattice-origin argument of the marching entry points -> (pointer, t0_scale, t0_base).  t0: a float32 tensor [R] (origins
    as given), or a tuple (u, scale, base[, table]): u = stratified draws [R] or None; origin = u * scale (+ base), resp. base."""
    if isinstance(t0, tuple):
        u, scale, base = t0[:3]
        return (_p(_f32(u, 't0')) if u is not None else None), float(scale if u is not None else 0.0), float(base)
    return _p(_f32(t0, 't0')), 0.0, 0.0


def _lat_table(t0):
    """The `lattice_table` argument of the marching entry points: t0 = (None, scale, base, table): the precomputed lattice of a
    launch whose rays all start at the same origin (lattice_table()); t0 = (u, scale, base, runs): the per-ray tables of runs of
    a stratified batch (lattice_runs(), int32); None otherwise."""
    if isinstance(t0, tuple) and len(t0) > 3 and t0[3] is not None:
        if t0[0] is None:
            return _p(_f32(t0[3], 'lattice_table'))
        if t0[3].dtype != torch.int32:
            raise _lib.PerfError('per-ray lattice runs must be int32 (ops.lattice_runs)')
        return _p(t0[3])
    return None


def lattice_runs(t0, step, max_steps, n_rays=None):
    """Per-ray tables of runs of the repeated-addition lattice for a batch with per-ray origins (perf_occ_lattice_runs: one lane per
    ray).  t0: a float32 tensor of origins or a tuple (u, scale, base) -- see _origin.  -> int32 tensor to append to the tuple as its
    fourth element: (u, scale, base, runs)."""
    u = t0[0] if isinstance(t0, tuple) else t0
    n = int(u.shape[0]) if n_rays is None else int(n_rays)
    out = torch.empty(_lib.load().perf_occ_lattice_runs_len(n), dtype=torch.int32, device=u.device)
    _call('perf_occ_lattice_runs', *_origin(t0), n, float(step), int(max_steps), _p(out), _stream())
    return out


def lattice_table(t0_base, step, max_steps, lattice=None, device='cuda'):
    """t_k of the lattice that starts at t0_base, for every k a marching launch of `max_steps` intervals can ask for
    (perf_occ_lattice_table) -- pass it as the fourth element of a (None, 0.0, t0_base, table) origin."""
    n = _lib.load().perf_occ_lattice_table_len(int(max_steps))
    out = torch.empty(n, dtype=torch.float32, device=device)
    _call('perf_occ_lattice_table', float(t0_base), float(step), int(max_steps), _lib.LATTICE[lattice], _p(out), _stream())
    return out


def occ_march_count(rays_o, rays_d, t0, occ_bits, res, aabb, far_plane, step, max_steps, occ_coarse=None, lattice=None):
    """Pass 1 of the marching: -> (keep masks, per-ray counts int32 [R]).  t0: see _origin."""
    R = rays_o.shape[0]
    dev = rays_o.device
    mw = _lib.load().perf_occ_mask_words(max_steps)
    masks = torch.empty(max(R * mw, 1), dtype=torch.int64, device=dev)
    counts = torch.empty(R, dtype=torch.int32, device=dev)
    _call('perf_occ_march_count', _p(_f32(rays_o, 'rays_o')), _p(_f32(rays_d, 'rays_d')), *_origin(t0), R,
          _p(occ_bits), _p(occ_coarse), int(res), _aabb6(aabb), float(far_plane), float(step), int(max_steps), _lib.LATTICE[lattice],
          _lat_table(t0), _p(masks), _p(counts), _stream())
    return masks, counts


def occ_march_count_head(rays_o, rays_d, t0, occ_bits, res, aabb, far_plane, step, max_steps, occ_coarse, head_k, points_aabb,
                         lattice=None):
    """occ_march_count that also writes the first head_k samples of every ray to rows r*head_k.. of R*head_k-row arrays
    (padding rows have sel = 0) -> (masks, counts, (ray_indices, t_starts, t_ends, packed_info, x01, sel))."""
    R = rays_o.shape[0]
    dev = rays_o.device
    mw = _lib.load().perf_occ_mask_words(max_steps)
    masks = torch.empty(max(R * mw, 1), dtype=torch.int64, device=dev)
    counts = torch.empty(R, dtype=torch.int32, device=dev)
    S = R * int(head_k)
    ri = torch.empty(S, dtype=torch.int64, device=dev)
    ts = torch.empty(S, dtype=torch.float32, device=dev)
    te = torch.empty(S, dtype=torch.float32, device=dev)
    packed = torch.empty(R, 2, dtype=torch.int32, device=dev)
    x01 = torch.empty(S, 3, dtype=torch.float32, device=dev)
    sel = torch.empty(S, dtype=torch.uint8, device=dev)
    _call('perf_occ_march_count_head', _p(_f32(rays_o, 'rays_o')), _p(_f32(rays_d, 'rays_d')), *_origin(t0), R,
          _p(occ_bits), _p(occ_coarse), int(res), _aabb6(aabb), float(far_plane), float(step), int(max_steps), _lib.LATTICE[lattice],
          _lat_table(t0), _p(masks), _p(counts), int(head_k), _p(ri), _p(ts), _p(te), _p(packed), _aabb6(points_aabb), _p(x01), _p(sel), _stream())
    return masks, counts, (ri, ts, te, packed, x01, sel)


def occ_march_write(t0, masks, counts, offsets, S, step, max_steps, rays_o=None, rays_d=None, points_aabb=None, rank_lo=0,
                    lattice=None):
    """Pass 2: expand the masks into S-row sample arrays -> (ray_indices, t_starts, t_ends, packed_info[, x01, sel]).
    counts / offsets: how many samples of every ray to write, starting at rank rank_lo, and where."""
    R = counts.shape[0]
    dev = counts.device
    ri = torch.empty(S, dtype=torch.int64, device=dev)
    ts = torch.empty(S, dtype=torch.float32, device=dev)
    te = torch.empty(S, dtype=torch.float32, device=dev)
    packed = torch.empty(R, 2, dtype=torch.int32, device=dev)
    if points_aabb is not None:
        x01 = torch.empty(S, 3, dtype=torch.float32, device=dev)
        sel = torch.empty(S, dtype=torch.uint8, device=dev)
        _call('perf_occ_march_write_points', *_origin(t0), R, float(step), int(max_steps), _lib.LATTICE[lattice], _lat_table(t0), _p(masks), _p(counts), _p(offsets), S,
              _p(ri), _p(ts), _p(te), _p(packed), _p(rays_o), _p(rays_d), _aabb6(points_aabb), _p(x01), _p(sel), int(rank_lo), _stream())
        return ri, ts, te, packed, x01, sel
    if rank_lo != 0:
        raise _lib.PerfError('rank_lo needs points_aabb (perf_occ_march_write_points)')
    _call('perf_occ_march_write', *_origin(t0), R, float(step), int(max_steps), _lib.LATTICE[lattice], _lat_table(t0), _p(masks), _p(counts), _p(offsets), S,
          _p(ri), _p(ts), _p(te), _p(packed), _stream())
    return ri, ts, te, packed


def occ_march(rays_o, rays_d, t0, occ_bits, res, aabb, far_plane, step, max_steps, capacity=None, occ_coarse=None,
              points_aabb=None, lattice=None):
    """Returns (ray_indices i64 [S], t_starts, t_ends f32 [S], packed_info i32 [R,2]).
    capacity=None reads the total back (one host sync, like the reference's boolean indexing);
    an int capacity keeps the call sync-free and returns arrays of that length plus `total` on device.
    points_aabb (6 floats): also return the sample positions (x01 [S,3], sel [S]) normalised to that box, written by
    the same kernel that writes the samples (appended to the result)."""
    masks, counts = occ_march_count(rays_o, rays_d, t0, occ_bits, res, aabb, far_plane, step, max_steps, occ_coarse, lattice=lattice)
    offsets, total = exclusive_scan_i32(counts)
    S = int(total.item()) if capacity is None else int(capacity)
    out = occ_march_write(t0, masks, counts, offsets, S, step, max_steps, rays_o, rays_d, points_aabb, lattice=lattice)
    if capacity is None:
        return out
    return out[:4] + (total,) + out[4:]


def head_tail_counts(counts, head_samples, kept_head=None):
    """kept_head None: min(counts, K); else the tail counts of the rays whose whole head survived (0 for decided rays)."""
    out = torch.empty_like(counts)
    _call('perf_head_tail_counts', _p(counts), counts.shape[0], int(head_samples), _p(kept_head), _p(out), _stream())
    return out


def visibility_count2(head, tail, early_stop_eps=1e-4):
    """head / tail = (sigmas, t_starts, t_ends, packed_info) of the two sample sets -> kept counts int32 [R]."""
    R = head[3].shape[0]
    thr = float(-math.log(early_stop_eps)) if early_stop_eps > 0 else float('inf')
    new_counts = torch.empty(R, dtype=torch.int32, device=head[3].device)
    _call('perf_visibility_count2', _p(head[0]), _p(head[1]), _p(head[2]), _p(head[3]), _p(tail[0]), _p(tail[1]), _p(tail[2]), _p(tail[3]),
          R, thr, _p(new_counts), _stream())
    return new_counts


def compact_prefix2(head, tail, new_counts, capacity, feat_h=None, feat_t=None):
    """head / tail = (sigmas, t_starts, t_ends, packed_info, x01, sel) -> (ray_indices, t_starts, t_ends, sigmas, packed_info,
    total, x01, sel[, feat]) with `capacity` rows; feat_h / feat_t: level-major features of the two sample sets."""
    R = new_counts.shape[0]
    dev = new_counts.device
    new_offsets, total = exclusive_scan_i32(new_counts)
    S = int(capacity)
    ri = torch.empty(S, dtype=torch.int64, device=dev)
    ts = torch.empty(S, dtype=torch.float32, device=dev)
    te = torch.empty(S, dtype=torch.float32, device=dev)
    sg = torch.empty(S, dtype=torch.float32, device=dev)
    xo = torch.empty(S, 3, dtype=torch.float32, device=dev)
    so = torch.empty(S, dtype=torch.uint8, device=dev)
    packed_out = torch.empty(R, 2, dtype=torch.int32, device=dev)
    fo = torch.empty(feat_h.shape[0], S, 2, dtype=feat_h.dtype, device=dev) if feat_h is not None else None
    _call('perf_compact_prefix2', _p(head[0]), _p(head[1]), _p(head[2]), _p(head[3]), _p(head[4]), _p(head[5]),
          _p(tail[0]), _p(tail[1]), _p(tail[2]), _p(tail[3]), _p(tail[4]), _p(tail[5]), _p(new_counts), _p(new_offsets), R, S,
          _p(ri), _p(ts), _p(te), _p(sg), _p(xo), _p(so), _p(packed_out),
          _p(feat_h), feat_h.shape[1] if feat_h is not None else 0, _p(feat_t), feat_t.shape[1] if feat_t is not None else 0,
          _p(fo), S, feat_h.shape[0] if feat_h is not None else 0, _stream())
    res = (ri, ts, te, sg, packed_out, total, xo, so)
    return res + (fo,) if fo is not None else res


# ---- compositing -----------------------------------------------------------------------------------
def visibility_count(sigmas, t_starts, t_ends, packed, early_stop_eps=1e-4, want_exsum=False, march_counts=None, head_samples=0):
    """-> kept counts int32 [R] (+ exsum); with march_counts / head_samples also the two-phase sampler's tail counts (same
    launch): (kept, tail_counts)."""
    R = packed.shape[0]
    thr = float(-math.log(early_stop_eps)) if early_stop_eps > 0 else float('inf')
    new_counts = torch.empty(R, dtype=torch.int32, device=packed.device)
    ex = torch.empty_like(sigmas) if want_exsum else None
    tail = torch.empty(R, dtype=torch.int32, device=packed.device) if march_counts is not None else None
    _call('perf_visibility_count', _p(_f32(sigmas, 'sigmas')), _p(t_starts), _p(t_ends), _p(packed), R, thr,
              _p(new_counts), _p(ex), _p(march_counts), int(head_samples), _p(tail), _stream())
    if march_counts is not None:
        return new_counts, tail
    return (new_counts, ex) if want_exsum else new_counts


INDEX_FEATURES = True      # compact_prefix: rows instead of a feature copy (index_features=False: the copy, tests)


def compact_prefix(packed, new_counts, t_starts, t_ends, sigmas=None, capacity=None, x01=None, sel=None, feat=None,
                   index_features=None):
    """-> (ray_indices, t_starts, t_ends, sigmas, packed_info) of the kept prefixes; with capacity (sync-free mode: the
    arrays keep that length, the kept count stays on the device) also `total` (int64 [1]); with x01/sel also the compacted
    positions, appended to the result; with feat (level-major features of all input samples) their compacted copy or --
    index_features (the default) -- an IndexedFeat that points into `feat` (which must then outlive it)."""
    R = packed.shape[0]
    dev = packed.device
    new_offsets, total = exclusive_scan_i32(new_counts)
    S = int(total.item()) if capacity is None else int(capacity)
    ri = torch.empty(S, dtype=torch.int64, device=dev)
    ts = torch.empty(S, dtype=torch.float32, device=dev)
    te = torch.empty(S, dtype=torch.float32, device=dev)
    sg = torch.empty(S, dtype=torch.float32, device=dev) if sigmas is not None else None
    xo = torch.empty(S, 3, dtype=torch.float32, device=dev) if x01 is not None else None
    so = torch.empty(S, dtype=torch.uint8, device=dev) if sel is not None else None
    packed_out = torch.empty(R, 2, dtype=torch.int32, device=dev)
    by_index = feat is not None and (INDEX_FEATURES if index_features is None else index_features)
    fo = torch.empty(feat.shape[0], S, 2, dtype=feat.dtype, device=dev) if (feat is not None and not by_index) else None     # level major like feat
    fi = torch.empty(S, dtype=torch.int32, device=dev) if by_index else None
    _call('perf_compact_prefix', _p(packed), _p(new_counts), _p(new_offsets), R, _p(t_starts), _p(t_ends), _p(sigmas),
          _p(ri), _p(ts), _p(te), _p(sg), _p(packed_out), _p(x01), _p(sel), _p(xo), _p(so),
          _p(feat if fo is not None else None), feat.shape[1] if fo is not None else 0, _p(fo), S, feat.shape[0] if fo is not None else 0,
          _p(fi), _stream())
    if by_index:
        fo = IndexedFeat(feat, fi)
    res = (ri, ts, te, sg, packed_out)
    if capacity is not None:
        res = res + (total,)
    if x01 is not None:
        res = res + (xo, so)
    if feat is not None:
        res = res + (fo,)
    return res


def composite_fwd(sigmas, rgbs, t_starts, t_ends, packed, want_samples=True):
    R = packed.shape[0]
    S = sigmas.numel()
    dev = sigmas.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    w = f(S) if want_samples else None
    T = f(S) if want_samples else None
    al = f(S) if want_samples else None
    op, dist = f(R, 1), f(R, 1)
    col = f(R, 3) if rgbs is not None else None
    _call('perf_composite_fwd', _p(_f32(sigmas, 'sigmas')), _p(rgbs), _p(t_starts), _p(t_ends), _p(packed), R, _p(w),
              _p(T), _p(al), _p(op), _p(dist), _p(col), _stream())
    return w, T, al, op, dist, col


def composite_distloss_fwd(sigmas, rgbs, t_starts, t_ends, packed):
    """composite_fwd + the per-ray distortion-loss sums in one launch -> (weights, trans, opacity, distance, colour, dl)."""
    R = packed.shape[0]
    S = sigmas.numel()
    dev = sigmas.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    w, T, op, dist, dl = f(S), f(S), f(R, 1), f(R, 1), f(R)
    col = f(R, 3) if rgbs is not None else None
    _call('perf_composite_distloss_fwd', _p(_f32(sigmas, 'sigmas')), _p(rgbs), _p(t_starts), _p(t_ends), _p(packed), R, _p(w),
          _p(T), _p(op), _p(dist), _p(col), _p(dl), _stream())
    return w, T, op, dist, col, dl


def composite_distloss_bwd(sigmas, t_starts, t_ends, packed, weights, trans, opacity, distance, g_opacity, g_distance,
                           scale=1.0, scale_dev=None):
    """d sigma of compositing with the distortion-loss gradient (scale * scale_dev[0] * d dl / d w) formed in the kernel."""
    ds = torch.empty(sigmas.numel(), dtype=torch.float32, device=sigmas.device)      # packed_info tiles [0, S): every sample is written
    _call('perf_composite_distloss_bwd', _p(sigmas), _p(t_starts), _p(t_ends), _p(packed), packed.shape[0], _p(weights), _p(trans),
          _p(opacity), _p(distance), _p(g_opacity), _p(g_distance), float(scale), _p(scale_dev), _p(ds), _stream())
    return ds


def composite_bwd(sigmas, t_starts, t_ends, packed, weights, trans, g_weights=None, g_opacity=None, g_distance=None,
                  g_color=None, g_trans=None, g_alphas=None, want_dsigma=True, want_drgb=False):
    R = packed.shape[0]
    S = sigmas.numel()
    dev = sigmas.device
    # (every sample of every ray is written by the kernel: no zero fill; rows beyond the live count of a capacity-sized batch
    #  stay uninitialised like everywhere else)
    ds = torch.empty(S, dtype=torch.float32, device=dev) if want_dsigma else None
    dr = torch.empty(S, 3, dtype=torch.float32, device=dev) if want_drgb else None
    _call('perf_composite_bwd', _p(sigmas), _p(t_starts), _p(t_ends), _p(packed), R, _p(weights), _p(trans),
              _p(g_weights), _p(g_trans), _p(g_alphas), _p(g_opacity), _p(g_distance), _p(g_color), _p(ds), _p(dr), _stream())
    return ds, dr


def render_finish_eval(opacity, distance=None, color=None, n_dev=None):
    """In place: distance += 5 (1 - opacity), color += 0.5 (1 - opacity) unless the batch (n_dev) holds no sample."""
    _call('perf_render_finish_eval', _p(_f32(opacity, 'opacity')), _p(distance), _p(color), opacity.shape[0], _nd(n_dev), _stream())


def accumulate_fwd(weights, values, packed):
    R = packed.shape[0]
    C = 1 if values is None else values.shape[-1]
    out = torch.empty(R, C, dtype=torch.float32, device=weights.device)
    _call('perf_accumulate_fwd', _p(_f32(weights, 'weights')), _p(values), _p(packed), R, C, _p(out), _stream())
    return out


def pack_info(ray_indices, n_rays):
    packed = torch.empty(n_rays, 2, dtype=torch.int32, device=ray_indices.device)
    _call('perf_pack_info', _p(ray_indices), ray_indices.numel(), n_rays, _p(packed), _stream())
    return packed


def distloss_fwd(w, t_starts, t_ends, packed):
    R = packed.shape[0]
    loss = torch.empty(R, dtype=torch.float32, device=w.device)
    _call('perf_distloss_fwd', _p(_f32(w, 'w')), _p(t_starts), _p(t_ends), _p(packed), R, _p(loss), _stream())
    return loss


def distloss_bwd(w, t_starts, t_ends, packed, scale, scale_dev=None):
    R = packed.shape[0]
    g = torch.zeros_like(w)
    _call('perf_distloss_bwd', _p(_f32(w, 'w')), _p(t_starts), _p(t_ends), _p(packed), R, float(scale), _p(scale_dev), _p(g), _stream())
    return g


_TICKETS = {}


def _ticket(device):
    """A zeroed device int32 per device for kernels that elect their last workgroup (left at zero by every call; the callers
    are serial on their stream)."""
    key = str(device)
    t = _TICKETS.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        if torch.cuda.is_current_stream_capturing():
            return t               # (created inside a capture: zero-filled by the graph itself on every replay)
        _TICKETS[key] = t
    return t


def geo_loss(opacity, distance, gt_distance, noise, distloss_per_ray, packed, global_batch, depth_weight, distortion_weight,
             ratio_dev, loss_scale):
    """-> (g_opacity [R,1], g_distance [R,1], scalars [3] = depth loss, distortion loss, distloss-backward scale)."""
    R = packed.shape[0]
    dev = opacity.device
    g_op = torch.empty(R, 1, dtype=torch.float32, device=dev); g_d = torch.empty(R, 1, dtype=torch.float32, device=dev)
    sc = torch.empty(_lib.LOSS_SCALARS, dtype=torch.float32, device=dev)
    _call('perf_geo_loss', _p(opacity), _p(distance), _p(_f32(gt_distance.contiguous(), 'gt')), _p(noise), _p(distloss_per_ray), _p(packed), R,
          int(global_batch), float(depth_weight), float(distortion_weight), _p(ratio_dev), float(loss_scale), _p(g_op), _p(g_d), _p(sc),
          _p(_ticket(dev)), _stream())
    return g_op, g_d, sc


def app_loss(opacity, color, bg_color, gt_color, global_batch, color_weight, loss_scale):
    """-> (g_color [R,3], scalars [1] = colour loss)."""
    R = opacity.shape[0]
    g_c = torch.empty(R, 3, dtype=torch.float32, device=opacity.device)
    sc = torch.empty(_lib.LOSS_SCALARS, dtype=torch.float32, device=opacity.device)
    _call('perf_app_loss', _p(opacity), _p(color), _p(bg_color), _p(_f32(gt_color.contiguous(), 'gt')), R, int(global_batch), float(color_weight),
          float(loss_scale), _p(g_c), _p(sc), _stream())
    return g_c, sc


def train_head_geo(sigmas, rgbs, t_starts, t_ends, packed, gt_distance, noise, global_batch, depth_weight, distortion_weight,
                   ratio_dev, loss_scale):
    """The ray head of the geometry step in ONE launch (perf_train_head_geo = composite_distloss_fwd -> geo_loss ->
    composite_distloss_bwd) -> dict(weights, trans, opacity, distance, color, depth_terms, distloss_per_ray, inv_n, d_sigma)."""
    R = packed.shape[0]
    S = sigmas.numel()
    dev = sigmas.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    o = dict(weights=f(S), trans=f(S), opacity=f(R, 1), distance=f(R, 1), color=f(R, 3) if rgbs is not None else None, depth_terms=f(R),
             distloss_per_ray=f(R), inv_n=f(1), d_sigma=f(S))
    _call('perf_train_head_geo', _p(_f32(sigmas, 'sigmas')), _p(rgbs), _p(t_starts), _p(t_ends), _p(packed), R, S,
          _p(_f32(gt_distance.contiguous(), 'gt')), _p(noise), int(global_batch), float(depth_weight), float(distortion_weight), _p(ratio_dev),
          float(loss_scale), _p(o['weights']), _p(o['trans']), _p(o['opacity']), _p(o['distance']), _p(o['color']), _p(o['depth_terms']),
          _p(o['distloss_per_ray']), _p(o['inv_n']), _p(o['d_sigma']), _stream())
    return o


def train_head_app(sigmas, rgbs, t_starts, t_ends, packed, bg_color, gt_color, global_batch, color_weight, loss_scale):
    """The ray head of the colour step in ONE launch (perf_train_head_app = composite_fwd -> app_loss -> composite_bwd) ->
    dict(weights, trans, opacity, distance, color, color_terms, d_rgb)."""
    R = packed.shape[0]
    S = sigmas.numel()
    dev = sigmas.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    o = dict(weights=f(S), trans=f(S), opacity=f(R, 1), distance=f(R, 1), color=f(R, 3), color_terms=f(R), d_rgb=f(S, 3))
    _call('perf_train_head_app', _p(_f32(sigmas, 'sigmas')), _p(_f32(rgbs, 'rgbs')), _p(t_starts), _p(t_ends), _p(packed), R, S, _p(bg_color),
          _p(_f32(gt_color.contiguous(), 'gt')), int(global_batch), float(color_weight), float(loss_scale), _p(o['weights']), _p(o['trans']),
          _p(o['opacity']), _p(o['distance']), _p(o['color']), _p(o['color_terms']), _p(o['d_rgb']), _stream())
    return o


def occ_splat(rays_o, rays_d, dist, res):
    n = rays_o.shape[0]
    occ = torch.zeros(res ** 3, dtype=torch.uint8, device=rays_o.device)
    _call('perf_occ_splat', _p(_f32(rays_o, 'rays_o')), _p(_f32(rays_d, 'rays_d')), _p(_f32(dist.reshape(-1), 'dist')), n,
              int(res), _p(occ), _stream())
    return occ


def gather_supervision(indices, o_all=None, d_all=None, color_all=None, dist_all=None, normal_all=None):
    """One-launch supervision batch: -> dict with 'o','d','color','normal' [n,3] and 'dist' [n,1] for the given sources."""
    n = indices.shape[0]
    dev = indices.device
    idx = indices.contiguous()
    assert idx.dtype == torch.int64
    out, src = {}, {}
    for key, all_, width in (('o', o_all, 3), ('d', d_all, 3), ('color', color_all, 3), ('dist', dist_all, 1), ('normal', normal_all, 3)):
        src[key] = None if all_ is None else _f32(all_, key)
        out[key] = None if all_ is None else torch.empty(n, width, dtype=torch.float32, device=dev)
    _call('perf_gather_supervision', _p(idx), n, _p(src['o']), _p(src['d']), _p(src['color']), _p(src['dist']), _p(src['normal']),
          _p(out['o']), _p(out['d']), _p(out['color']), _p(out['dist']), _p(out['normal']), _stream())
    return out


def draw_train_batch(seed, counter, pool_lo, pool_hi, n_local, first_global, o_all, d_all, color_all, dist_all, normal_all=None,
                     want_bg=False, want_indices=False):
    """One launch: the step's batch indices (uniform over [pool_lo, pool_hi)), the gathered supervision rows and the per-ray
    uniforms (jitter, noise[, bg]) from the counter-based generator (perf_draw_train_batch).  counter: device int64 [1],
    advanced by the launch.  -> dict(o, d, color, dist, normal, jitter [n], noise [n,1], bg [n,3] | None, indices | None)."""
    dev = o_all.device
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    out = {'o': f(n_local, 3), 'd': f(n_local, 3), 'color': f(n_local, 3), 'dist': f(n_local, 1),
           'normal': f(n_local, 3) if normal_all is not None else None, 'jitter': f(n_local), 'noise': f(n_local, 1),
           'bg': f(n_local, 3) if want_bg else None,
           'indices': torch.empty(n_local, dtype=torch.int64, device=dev) if want_indices else None}
    _call('perf_draw_train_batch', int(seed) & 0xFFFFFFFFFFFFFFFF, _nd(counter), _p(_ticket(dev)), int(pool_lo), int(pool_hi), int(n_local),
          int(first_global), _p(_f32(o_all, 'o')), _p(_f32(d_all, 'd')), _p(_f32(color_all, 'color')), _p(_f32(dist_all, 'dist')),
          _p(normal_all), _p(out['o']), _p(out['d']), _p(out['color']), _p(out['dist']), _p(out['normal']), _p(out['indices']),
          _p(out['jitter']), _p(out['noise']), _p(out['bg']), _stream())
    return out


def pdf_resample(s_in, cdf, n_out, tau=None):
    """s_in, cdf [R, n_in+1] -> s_out [R, n_out+1] (inverse-CDF resampling, see perf_pdf_resample)."""
    R, n1 = s_in.shape
    out = torch.empty(R, n_out + 1, dtype=torch.float32, device=s_in.device)
    _call('perf_pdf_resample', _p(_f32(s_in.contiguous(), 's_in')), _p(_f32(cdf.contiguous(), 'cdf')), _p(tau), R, n1 - 1,
          int(n_out), _p(out), _stream())
    return out


# ---- reprojection visibility tests ---------------------------------------------------------------------------------
def pano_reproject(pts, pose, distance_map, mask, mode, eps=1.0 / 256.0):
    """In place on mask [n] (float 0/1): fold one registered panorama's depth test into it (perf_pano_reproject)."""
    n = pts.shape[0]
    h, w = distance_map.shape
    pose_h = (ctypes.c_float * 16)(*[float(v) for v in torch.as_tensor(pose).detach().cpu().reshape(-1).tolist()])
    _call('perf_pano_reproject', _p(_f32(pts, 'pts')), n, pose_h, _p(_f32(distance_map, 'distance_map')), int(h), int(w), int(mode),
          float(eps), _p(_f32(mask, 'mask')), _stream())
    return mask


def morph_binary(img, element, op):
    """img [H,W] float 0/1, element: 2-D 0/1 array (rows <= 16, cols <= 32), op 'dilate' | 'erode' -> new [H,W] float."""
    h, w = img.shape
    rows, cols = int(element.shape[0]), int(element.shape[1])
    bits = (ctypes.c_uint32 * rows)(*[sum(1 << c for c in range(cols) if float(element[r][c]) > 0.5) for r in range(rows)])
    out = torch.empty_like(img)
    _call('perf_morph_binary', _p(_f32(img, 'img')), _p(out), int(h), int(w), bits, rows, cols, 0 if op == 'dilate' else 1, _stream())
    return out
