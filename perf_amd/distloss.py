"""torch_efficient_distloss surface used by PeRF (modules/scene/nerf.py:23,230) on the gfx950 kernels.
flatten_eff_distloss(w, m, interval, ray_id) = (1/3 sum d_i w_i^2 + 2 sum w_i (m_i W_i - WM_i)) / n_rays with
n_rays = ray_id.max()+1 and W, WM the per-ray exclusive prefixes; gradient w.r.t. w only."""
import torch

from . import ops
from .nerfacc_impl import _packed_of


class _DistLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, ts, te, packed, inv_n):
        per_ray = ops.distloss_fwd(w, ts, te, packed)
        ctx.save_for_backward(w, ts, te, packed, inv_n)
        return per_ray.sum() * inv_n

    @staticmethod
    def backward(ctx, g):
        w, ts, te, packed, inv_n = ctx.saved_tensors
        gw = ops.distloss_bwd(w, ts, te, packed, 1.0)
        return gw * (g * inv_n), None, None, None, None


def flatten_eff_distloss(w, m, interval, ray_id, packed_info=None, n_rays_total=None):
    """m = interval midpoints, interval = lengths: the kernels want (t_start, t_end) = m -+ interval/2."""
    if w.numel() == 0:
        return w.sum()
    half = interval * 0.5
    ts = (m - half).contiguous().float()
    te = (m + half).contiguous().float()
    last = ray_id[-1:].to(torch.float32) + 1.0                 # ray_id is sorted: max == last (stays on device)
    inv_n = 1.0 / last
    if packed_info is None:
        n = n_rays_total if n_rays_total is not None else int(ray_id[-1].item()) + 1
        packed_info = _packed_of(ray_id, n)
    return _DistLossFn.apply(w.contiguous().float(), ts, te, packed_info, inv_n)


def eff_distloss(w, m, interval):
    """Dense [R, n] variant (only reachable from PeRF's dead proposal-network branch, nerf.py:222)."""
    R, n = w.shape
    ray_id = torch.arange(R, device=w.device).repeat_interleave(n)
    return flatten_eff_distloss(w.reshape(-1), m.reshape(-1), interval.reshape(-1), ray_id)
