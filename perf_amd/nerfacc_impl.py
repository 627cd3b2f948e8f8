"""The part of nerfacc's surface PeRF uses, on the gfx950 kernels.

PeRF imports (modules/scene/nerf_renderer.py:5-7, modules/scene/nerf.py:24-25):
    from nerfacc import accumulate_along_rays, render_weight_from_density, render_transmittance_from_alpha
    from nerfacc.estimators.occ_grid import OccGridEstimator
    from nerfacc.estimators.prop_net import PropNetEstimator
Semantics follow nerfacc==0.5.3 as restated in SURVEY.md Appendix A.3/A.4 (parity unpinned: the package is
not in the reference tree); integer bookkeeping is bit-exact against oracle/perf_oracle.py.
"""
import math

import torch
import torch.nn as nn

from . import ops


def _packed_of(ray_indices, n_rays):
    """packed_info for sorted ray_indices; sampling() attaches it to the tensor it returns so the usual
    sampling -> render_weight_from_density -> accumulate_along_rays chain never recomputes it."""
    cached = getattr(ray_indices, '_perf_packed', None)
    if cached is not None and cached.shape[0] == n_rays:
        return cached
    packed = ops.pack_info(ray_indices.contiguous(), int(n_rays))
    try:
        ray_indices._perf_packed = packed
    except Exception:
        pass
    return packed


class _WeightsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas, t_starts, t_ends, packed):
        ctx.set_materialize_grads(False)
        w, T, al, _, _, _ = ops.composite_fwd(sigmas, None, t_starts, t_ends, packed)
        ctx.save_for_backward(sigmas, t_starts, t_ends, packed, w, T)
        return w, T, al

    @staticmethod
    def backward(ctx, g_w, g_T, g_al):
        sigmas, ts, te, packed, w, T = ctx.saved_tensors
        f = lambda g: None if g is None else g.contiguous().float()
        ds, _ = ops.composite_bwd(sigmas, ts, te, packed, w, T, g_weights=f(g_w), g_trans=f(g_T), g_alphas=f(g_al))
        return ds, None, None, None


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None,
                               prefix_trans=None):
    """-> (weights, trans, alphas), each [S].  Call site: nerf_renderer.py:170-171."""
    if prefix_trans is not None:
        raise NotImplementedError('prefix_trans is not used by PeRF')
    if t_starts.dim() != 1:
        raise NotImplementedError('only the packed (flattened) layout PeRF uses is supported')
    if packed_info is None:
        if ray_indices is None:
            raise ValueError('ray_indices or packed_info is required')
        if n_rays is None:
            n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
        packed_info = _packed_of(ray_indices, n_rays)
    elif packed_info.dtype != torch.int32:
        packed_info = packed_info.to(torch.int32).contiguous()
    return _WeightsFn.apply(sigmas.contiguous().float(), t_starts.contiguous(), t_ends.contiguous(), packed_info)


class _AccumulateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, values, ray_indices, packed):
        out = ops.accumulate_fwd(weights, values, packed)
        ctx.save_for_backward(weights, values if values is not None else torch.empty(0, device=weights.device), ray_indices)
        ctx.has_values = values is not None
        return out

    @staticmethod
    def backward(ctx, g_out):
        weights, values, ray_indices = ctx.saved_tensors
        g = g_out[ray_indices]                                   # [S, C] gather: elementwise from here on
        gw = gv = None
        if ctx.needs_input_grad[0]:
            gw = (g * values).sum(-1) if ctx.has_values else g[:, 0]
        if ctx.has_values and ctx.needs_input_grad[1]:
            gv = g * weights[:, None]
        return gw, gv, None, None


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    """-> [n_rays, C] (C = 1 when values is None).  Call sites: nerf_renderer.py:173,175,183."""
    if ray_indices is None:
        raise NotImplementedError('only the packed layout (ray_indices given) is supported')
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
    packed = _packed_of(ray_indices, n_rays)
    v = None if values is None else values.contiguous().float()
    return _AccumulateFn.apply(weights.contiguous().float(), v, ray_indices, packed)


def render_transmittance_from_alpha(*args, **kwargs):
    raise NotImplementedError('imported but never called by PeRF (nerf_renderer.py:5)')


def _sig_feat(res):
    """sigma_points_fn result -> (sigmas [n] fp32 contiguous, level-major features or None)."""
    sig, feat = res if isinstance(res, tuple) else (res, None)
    return sig.reshape(-1).float().contiguous(), feat


def _lib_default_lattice():
    from . import _lib
    return _lib.DEFAULT_LATTICE


# calls of OccGridEstimator.update_every_n_steps' warm-up branch in this process, per torch seed (the jitter counter): the
# reference's torch.rand stream continues across estimators and starts over with torch.manual_seed(another seed)
_UPDATE_CALLS = {}


class Samples:
    """Packed samples of one ray batch (see OccGridEstimator.sampling_ex)."""
    __slots__ = ('ray_indices', 't_starts', 't_ends', 'packed', 'sig', 'x01', 'sel', 'n_dev', 'n_marched_dev', 'feat')

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)


class OccGridEstimator(nn.Module):
    """nerfacc.estimators.occ_grid.OccGridEstimator for levels == 1 (PeRF: nerf.py:68,144)."""

    def __init__(self, roi_aabb, resolution=128, levels=1, **kwargs):
        super().__init__()
        if levels != 1:
            raise NotImplementedError('PeRF uses a single-level occupancy grid')
        if isinstance(resolution, int):
            resolution = [resolution] * 3
        if not (resolution[0] == resolution[1] == resolution[2]):
            raise NotImplementedError('cubic grids only')
        if not torch.is_tensor(roi_aabb):
            roi_aabb = torch.tensor(roi_aabb, dtype=torch.float32)
        res = int(resolution[0])
        self.levels = 1
        self.cells_per_lvl = res ** 3
        self.register_buffer('resolution', torch.tensor(resolution, dtype=torch.int32))
        self.register_buffer('aabbs', roi_aabb.detach().float().reshape(1, 6).clone())
        self.register_buffer('occs', torch.zeros(self.cells_per_lvl, dtype=torch.float32))
        self.register_buffer('binaries', torch.zeros(1, res, res, res, dtype=torch.bool))
        # host copies of the constants the launchers need (no device read-back on the hot path)
        self._res = res
        self._aabb_host = [float(v) for v in roi_aabb.detach().cpu().reshape(-1).tolist()]
        self._diag = math.sqrt(sum((self._aabb_host[3 + i] - self._aabb_host[i]) ** 2 for i in range(3)))
        self._bits = None
        self._coarse = None
        self._bits_version = None

    # -- occupancy bit field used by the marching kernel -------------------------------------------
    def occ_bits(self):
        key = (self.binaries.data_ptr(), self.binaries._version)
        if self._bits is None or self._bits_version != key:
            self._bits = ops.occ_pack_bits(self.binaries)
            self._coarse = ops.occ_build_coarse(self._bits, self._res)
            self._bits_version = key
        return self._bits

    def occ_coarse(self):
        self.occ_bits()
        return self._coarse

    def set_binaries(self, occ_flat):
        """Install a precomputed occupancy (x-major flat uint8/bool [res^3])."""
        res = self._res
        self.binaries = occ_flat.reshape(1, res, res, res).bool().to(self.binaries.device)
        self.occs = self.binaries.reshape(-1).float()
        self._bits = None

    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn=None, alpha_fn=None, near_plane=0.0, far_plane=1e10, t_min=None,
                 t_max=None, render_step_size=1e-3, early_stop_eps=1e-4, alpha_thre=0.0, stratified=False,
                 cone_angle=0.0):
        sm = self.sampling_ex(rays_o, rays_d, sigma_fn, near_plane, far_plane, render_step_size,
                              early_stop_eps, alpha_thre, stratified, cone_angle, alpha_fn, t_min, t_max)
        return sm.ray_indices, sm.t_starts, sm.t_ends

    @torch.no_grad()
    def sampling_ex(self, rays_o, rays_d, sigma_fn=None, near_plane=0.0, far_plane=1e10, render_step_size=1e-3,
                    early_stop_eps=1e-4, alpha_thre=0.0, stratified=False, cone_angle=0.0, alpha_fn=None,
                    t_min=None, t_max=None, jitter=None, max_steps=None, capacity=None, points_aabb=None,
                    sigma_points_fn=None, head_samples=None, lattice=None):
        """sampling() returning a Samples record: ray_indices, t_starts, t_ends, packed (packed_info), sig (sigmas of the
        kept samples from the visibility pass, or None), x01 / sel (sample positions normalised to points_aabb, or None),
        n_dev (device int64 [1]: number of live samples when the arrays are capacity-sized, else None), n_marched_dev.
        max_steps caps the number of lattice intervals per ray (fixed-count benchmark mode).
        capacity: sync-free mode (hipGraph-capturable): every array has `capacity` rows, the sample counts stay on the
          device (n_dev) and no host read-back happens; a batch that marches more than `capacity` samples is truncated ray
          by ray (n_marched_dev > capacity tells).  Needs points_aabb and, for the visibility pass, sigma_points_fn(x01, sel,
          n_dev) -> sigmas [capacity].
        points_aabb: the sample positions normalised to that box are produced by the marching kernel itself.
        head_samples (sync-free mode with a visibility pass): two-phase early termination -- the density pass first runs
          on the first `head_samples` samples of every ray and then only on the rest of the rays that are still alive (a
          trained scene terminates a ray after a sample or two); same samples, same sigmas as the one-phase path.
          n_marched_dev then counts the samples whose density was evaluated.
        sigma_points_fn may return (sigmas, feat): the level-major encoded features of its density pass then travel with the
          samples (Samples.feat) so that a gradient pass on the kept samples need not encode them again -- compacted along
          (two-phase sampler) or, from the one-phase sampler, as an ops.IndexedFeat: the uncompacted array plus the row of
          every kept sample, which ops.mlp_bwd reads in place (.materialize() gives the compacted copy).
        lattice: 'repeated' (t_{k+1} = fl(t_k + step), the default: None) or 'single' (t_k = fl(t0 + fl(k step))): PERF_LATTICE_*."""
        if cone_angle != 0.0 or alpha_fn is not None or t_min is not None or t_max is not None:
            raise NotImplementedError('PeRF samples with cone_angle=0 and a sigma_fn (nerf_renderer.py:145-155)')
        if alpha_thre != 0.0:
            raise NotImplementedError('alpha_thre > 0 is not used by PeRF')
        R = rays_o.shape[0]
        dev = rays_o.device
        rays_o = rays_o.contiguous().float(); rays_d = rays_d.contiguous().float()
        aabb = self._aabb_host
        span = min(float(far_plane) - float(near_plane), self._diag)
        anchor = None
        if float(far_plane) - float(near_plane) > self._diag:
            # nerfacc starts a ray's march where it enters the box; with a lattice that is shorter than [near, far] the
            # origin must be anchored there too, or rays that start outside the box lose their samples.  (PeRF's
            # far - near = 1.5 < diagonal with cameras inside the box never takes this branch.)
            lo, hi = self.aabbs[0, :3], self.aabbs[0, 3:]            # (device buffer: no host copy, capture safe)
            inv = 1.0 / rays_d
            t1 = (lo - rays_o) * inv; t2 = (hi - rays_o) * inv
            t_in = torch.fmin(t1, t2).nan_to_num(nan=-float('inf')).amax(-1)
            anchor = torch.clamp(t_in, min=float(near_plane)).nan_to_num(posinf=float(near_plane))
        # lattice origin near + u * step: formed inside the marching kernels from (u, step, near) -- see ops._origin; only the
        # anchored case (rays that start outside the box, never PeRF's) materialises it
        if stratified:
            u = torch.rand(R, device=dev) if jitter is None else jitter.contiguous()
            t0 = (u, float(render_step_size), float(near_plane)) if anchor is None else u * render_step_size + anchor
        else:
            t0 = (None, 0.0, float(near_plane)) if anchor is None else anchor
        if max_steps is None:
            max_steps = int(math.ceil(span / render_step_size)) + 1
        if isinstance(t0, tuple) and t0[0] is None:
            # every ray on the same lattice (no jitter): its points are computed once and read by the kernels with one load
            t0 = t0 + (self._shared_lattice(float(near_plane), float(render_step_size), int(max_steps), lattice, dev),)
        elif isinstance(t0, tuple) and (lattice or _lib_default_lattice()) == 'repeated':
            # per-ray origins on the repeated-addition lattice: every ray's table of runs from ONE launch that gives each ray a lane
            # (the marching kernels would spend a wavefront per ray on it)
            t0 = t0 + (ops.lattice_runs(t0, float(render_step_size), int(max_steps)),)
        res = self._res
        compacts = (sigma_fn is not None or sigma_points_fn is not None) and early_stop_eps > 0
        sm = Samples()
        if capacity is not None:
            if compacts and (sigma_points_fn is None or points_aabb is None):
                raise ValueError('sync-free sampling with a visibility pass needs points_aabb and sigma_points_fn')
            if compacts and head_samples:
                return self._sample_two_phase(sm, rays_o, rays_d, t0, float(far_plane), float(render_step_size), max_steps,
                                              int(capacity), points_aabb, sigma_points_fn, early_stop_eps, int(head_samples), lattice)
            out = ops.occ_march(rays_o, rays_d, t0, self.occ_bits(), res, aabb, float(far_plane), float(render_step_size),
                                max_steps, capacity=capacity, occ_coarse=self.occ_coarse(), points_aabb=points_aabb, lattice=lattice)
            ri, ts, te, packed, total = out[:5]
            x01, sel = out[5:] if points_aabb is not None else (None, None)
            sm.n_marched_dev = total
            sig = None
            if compacts:
                sig, feat = _sig_feat(sigma_points_fn(x01, sel, total))
                new_counts = ops.visibility_count(sig, ts, te, packed, early_stop_eps)
                out = ops.compact_prefix(packed, new_counts, ts, te, sig, capacity=capacity, x01=x01, sel=sel, feat=feat)
                ri, ts, te, sig, packed, total, x01, sel = out[:8]
                sm.feat = out[8] if feat is not None else None
            sm.n_dev = total
        else:
            out = ops.occ_march(rays_o, rays_d, t0, self.occ_bits(), res, aabb, float(far_plane), float(render_step_size), max_steps,
                                occ_coarse=self.occ_coarse(), points_aabb=None if compacts else points_aabb, lattice=lattice)
            ri, ts, te, packed = out[:4]
            x01, sel = out[4:] if (points_aabb is not None and not compacts) else (None, None)
            sig = None
            if compacts and ri.numel() > 0:
                ri._perf_packed = packed
                if sigma_points_fn is not None and points_aabb is not None:
                    x01, sel = ops.points_from_rays(rays_o, rays_d, ri, ts, te, points_aabb)
                    sig = sigma_points_fn(x01, sel, None)
                else:
                    sig = sigma_fn(ts, te, ri)
                    x01 = sel = None
                sig = sig.reshape(-1).float().contiguous()
                new_counts = ops.visibility_count(sig, ts, te, packed, early_stop_eps)
                if x01 is not None:
                    ri, ts, te, sig, packed, x01, sel = ops.compact_prefix(packed, new_counts, ts, te, sig, x01=x01, sel=sel)
                else:
                    ri, ts, te, sig, packed = ops.compact_prefix(packed, new_counts, ts, te, sig)
        ri._perf_packed = packed
        sm.ray_indices, sm.t_starts, sm.t_ends, sm.packed, sm.sig, sm.x01, sm.sel = ri, ts, te, packed, sig, x01, sel
        return sm

    def _shared_lattice(self, t0_base, step, max_steps, lattice, device):
        """The precomputed lattice table of jitter-free launches (ops.lattice_table), memoised per (origin, step, steps, lattice).
        (Created outside a capture when possible: a table first made INSIDE a captured render belongs to that graph's pool.)"""
        key = (t0_base, step, max_steps, lattice or _lib_default_lattice(), str(device))
        cache = self.__dict__.setdefault('_lattice_tables', {})
        tab = cache.get(key)
        if tab is None:
            tab = ops.lattice_table(t0_base, step, max_steps, lattice, device)
            if not torch.cuda.is_current_stream_capturing():
                cache[key] = tab
        return tab

    STRIDED_HEAD_MAX = 16          # heads of up to this many samples are written by the counting pass, K rows per ray

    def _sample_two_phase(self, sm, rays_o, rays_d, t0, far_plane, step, max_steps, capacity, points_aabb, sigma_points_fn,
                          early_stop_eps, K, lattice=None):
        """March once; density + visibility on the first K samples of every ray; then density on the remaining samples of
        the rays that are still alive; final visibility + compaction over (head, tail).  All counts stay on the device."""
        R = rays_o.shape[0]
        # ---- head: rank [0, K) of every ray, written by the counting pass itself to rows r*K..r*K+K-1 (rays with fewer
        #      samples leave padding rows with selector 0: their density is evaluated and ignored)
        if K <= self.STRIDED_HEAD_MAX:
            masks, counts, (ri_h, ts_h, te_h, pk_h, x_h, s_h) = ops.occ_march_count_head(
                rays_o, rays_d, t0, self.occ_bits(), self._res, self._aabb_host, far_plane, step, max_steps, self.occ_coarse(), K, points_aabb,
                lattice=lattice)
            total_h = None                      # R * K rows, a host constant: folded into the tail scan's biased total below
            sig_h, feat_h = _sig_feat(sigma_points_fn(x_h, s_h, None))
        else:       # a long head: packed rows (count clamp, scan over the rays, write pass) instead of K rows per ray
            masks, counts = ops.occ_march_count(rays_o, rays_d, t0, self.occ_bits(), self._res, self._aabb_host, far_plane, step,
                                                max_steps, self.occ_coarse(), lattice=lattice)
            ch = ops.head_tail_counts(counts, K)
            oh, total_h = ops.exclusive_scan_i32(ch)
            ri_h, ts_h, te_h, pk_h, x_h, s_h = ops.occ_march_write(t0, masks, ch, oh, R * K, step, max_steps, rays_o, rays_d, points_aabb,
                                                                   lattice=lattice)
            sig_h, feat_h = _sig_feat(sigma_points_fn(x_h, s_h, total_h))
        # ---- head decision and the tail counts (rank [K, count) of the rays whose whole head survived) in one launch
        kept_h, ct = ops.visibility_count(sig_h, ts_h, te_h, pk_h, early_stop_eps, march_counts=counts, head_samples=K)
        if total_h is None:
            ot, total_t, n_evaluated = ops.exclusive_scan_i32(ct, bias=R * K)
        else:
            ot, total_t = ops.exclusive_scan_i32(ct)
            n_evaluated = total_h + total_t
        ri_t, ts_t, te_t, pk_t, x_t, s_t = ops.occ_march_write(t0, masks, ct, ot, capacity, step, max_steps, rays_o, rays_d, points_aabb,
                                                               rank_lo=K, lattice=lattice)
        sig_t, feat_t = _sig_feat(sigma_points_fn(x_t, s_t, total_t))
        # ---- final decision and compaction over both sample sets
        head = (sig_h, ts_h, te_h, pk_h, x_h, s_h); tail = (sig_t, ts_t, te_t, pk_t, x_t, s_t)
        new_counts = ops.visibility_count2(head[:4], tail[:4], early_stop_eps)
        out = ops.compact_prefix2(head, tail, new_counts, capacity, feat_h, feat_t)
        ri, ts, te, sig, packed, total, x01, sel = out[:8]
        sm.feat = out[8] if feat_h is not None else None
        ri._perf_packed = packed
        sm.ray_indices, sm.t_starts, sm.t_ends, sm.packed, sm.sig, sm.x01, sm.sel = ri, ts, te, packed, sig, x01, sel
        sm.n_dev = total
        sm.n_marched_dev = n_evaluated                # rows whose density was evaluated (what must fit the capacity)
        return sm

    @torch.no_grad()
    def update_every_n_steps(self, step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16):
        """Follows nerfacc: while step < warmup_steps every cell is evaluated at (coord + U[0,1))/res mapped to
        the aabb; occs = max(occs*ema_decay, occ); binaries = occs > min(mean(occs), occ_thre).
        (PeRF: nerf.py:160-168 -- 256 warm-up calls with an occupancy look-up closure.)"""
        if not self.training:
            raise RuntimeError('update_every_n_steps() should only be called in training mode')
        if step % n != 0:
            return
        res = self._res
        dev = self.occs.device
        if step < warmup_steps:
            # Every cell: two launches around the caller's closure instead of ~20 element-wise torch passes over 16.7 M x 3
            # floats -- the jittered points (counter-based generator), then the moving maximum with its sum; one more
            # launch thresholds.  The cells go through the closure in chunks of 2 M, so that whatever element-wise work the
            # closure itself does (PeRF's look-up closure: ~10 passes, nerf.py:149-158) runs out of the Infinity Cache.
            if getattr(self, '_upd_seed', None) != int(torch.initial_seed()):
                self._upd_seed = int(torch.initial_seed())
            if getattr(self, '_upd_sum', None) is None:
                self._upd_sum = torch.zeros(1, dtype=torch.float64, device=dev)
            # the jitter stream continues ACROSS estimators, like the torch.rand stream of the reference does across episodes
            # (PeRF builds a fresh OccGridEstimator per fit, nerf.py:144): a process-wide call counter, not a per-instance one
            _UPDATE_CALLS[self._upd_seed] = _UPDATE_CALLS.get(self._upd_seed, 0) + 1
            self._upd_calls = _UPDATE_CALLS[self._upd_seed]
            self._upd_sum.zero_()
            n = self.cells_per_lvl
            occs = self.occs if self.occs.is_contiguous() else self.occs.contiguous()
            for lo in range(0, n, self.UPDATE_CHUNK):
                m = min(self.UPDATE_CHUNK, n - lo)
                x = ops.occ_jitter_points(self._upd_seed, self._upd_calls, lo, m, res, self._aabb_host, dev)
                occ = occ_eval_fn(x).reshape(-1).float().contiguous()
                ops.occ_ema_update(occs[lo:lo + m], occ, ema_decay, self._upd_sum)
            self.occs = occs
            self.binaries = ops.occ_threshold(occs, self._upd_sum, occ_thre).reshape(self.binaries.shape)
            self._bits = None
            return
        k = self.cells_per_lvl // 4
        uni = torch.randint(self.cells_per_lvl, (k,), device=dev)
        occ_idx = torch.nonzero(self.binaries.reshape(-1))[:, 0]
        if occ_idx.numel() > k:
            occ_idx = occ_idx[torch.randint(occ_idx.numel(), (k,), device=dev)]
        idx = torch.cat([uni, occ_idx])
        cz = idx % res; cy = (idx // res) % res; cx = idx // (res * res)
        coords = torch.stack([cx, cy, cz], -1).float()
        x = (coords + torch.rand_like(coords)) / res
        aabb = self.aabbs[0]
        x = aabb[:3] + x * (aabb[3:] - aabb[:3])
        occ = occ_eval_fn(x).reshape(-1).float()
        self.occs[idx] = torch.maximum(self.occs[idx] * ema_decay, occ)
        thre = torch.clamp(self.occs.mean(), max=occ_thre)
        self.binaries = (self.occs > thre).reshape(self.binaries.shape)
        self._bits = None

    UPDATE_CHUNK = 1 << 21


class PropNetEstimator(nn.Module):
    """nerfacc.estimators.prop_net.PropNetEstimator.sampling (forward): proposal-network guided hierarchical
    resampling.  Reachable in PeRF only through estimator_type == 'prop', which is dead in the
    reference (NameError at nerf_renderer.py:73; the proposal networks are never trained), so this follows the
    repo's own restatement (oracle/perf_oracle.py:prop_sampling) and supports inference only."""

    def __init__(self, optimizer=None, scheduler=None):
        super().__init__()
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.prop_cache = []

    @torch.no_grad()
    def sampling(self, prop_sigma_fns, prop_samples, num_samples, n_rays, near_plane, far_plane,
                 sampling_type='uniform', stratified=False, requires_grad=False, taus=None):
        """-> (t_starts, t_ends) [n_rays, num_samples].  taus: optional list of per-ray stratified draws (tests)."""
        if requires_grad:
            raise NotImplementedError('proposal-network training is not on any live PeRF path')
        if sampling_type != 'uniform':
            raise NotImplementedError("PeRF samples with sampling_type='uniform' (nerf_renderer.py:67)")
        dev = next(iter(self.buffers()), torch.empty(0, device='cuda')).device if False else torch.device('cuda', torch.cuda.current_device())
        s = torch.cat([torch.zeros(n_rays, 1, device=dev), torch.ones(n_rays, 1, device=dev)], -1)
        cdf = s.clone()
        levels = list(zip(prop_sigma_fns, prop_samples)) + [(None, num_samples)]
        for li, (fn, n_out) in enumerate(levels):
            tau = None
            if stratified:
                tau = taus[li] if taus is not None else torch.rand(n_rays, device=dev)
            s = ops.pdf_resample(s, cdf, n_out, tau)
            t_vals = near_plane + s * (far_plane - near_plane)
            t_starts, t_ends = t_vals[:, :-1].contiguous(), t_vals[:, 1:].contiguous()
            if fn is None:
                return t_starts, t_ends
            sigmas = fn(t_starts, t_ends)
            sd = sigmas * (t_ends - t_starts)
            trans = torch.exp(-(torch.cumsum(sd, -1) - sd))                      # exclusive: T_i = exp(-sum_{j<i} sd_j)
            cdf = 1.0 - torch.cat([trans, torch.zeros_like(trans[:, :1])], -1)
            cdf = torch.cummax(cdf.clamp(0, 1), -1).values.contiguous()          # sigma = inf tails make NaN-free, monotone CDFs


def render_weight_from_alpha(alphas):
    """Dense [R, n] alpha compositing: T_i = prod_{j<i} (1 - alpha_j), w = T * alpha (the function
    nerf_renderer.py:73 calls without importing it)."""
    T = torch.cumprod(torch.cat([torch.ones_like(alphas[:, :1]), 1.0 - alphas[:, :-1]], -1), -1)
    return T * alphas, T
