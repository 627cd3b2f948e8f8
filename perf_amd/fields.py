"""Host-side mirror of PeRF's modules/fields/ngp_nerf.py on the gfx950 kernels: same class names, method
names, argument meaning and state_dict keys (`aabb`, `geo_mlp.params`, `app_mlp.params`).

The arithmetic of the two tcnn networks is in perf_amd.tcnn; what this file adds is the reference's glue
(ngp_nerf.py:136-176): aabb normalisation, the 0<x<1 selector, trunc_exp on the density logit -- fused here
into the position and MLP-epilogue kernels (PERF_ACT_EXP + selector), so a density query is three launches.
"""
from typing import List, Union

import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import tcnn
from .tcnn import _DualFieldFn, field_apply

PER_LEVEL_SCALE = 1.4472692012786865


def _grid_cfg(n_levels=16, log2_hashmap_size=18, base_resolution=16, per_level_scale=PER_LEVEL_SCALE):
    return {"otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": 2,
            "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution,
            "per_level_scale": per_level_scale}


class _TruncExp(torch.autograd.Function):
    """exp forward, gradient exp(min(x, 15)) (ngp_nerf.py:24-40)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(torch.clamp(x, max=15))


trunc_exp = _TruncExp.apply


def contract_to_unisphere(x, aabb, eps: float = 1e-6):
    """Scene contraction of mip-NeRF 360 as used by ngp_nerf.py:43-65 for `unbounded=True` (never enabled by PeRF):
    points of the box map to the inner half of the unit ball, everything beyond to the shell between radius 1 and 2;
    the result is rescaled to [0, 1]^3.  (The reference's `derivative=True` branch has no caller and is not mirrored.)"""
    lo, hi = aabb[..., :3], aabb[..., 3:]
    u = (x - lo) / (hi - lo) * 2.0 - 1.0                      # box -> [-1, 1]^3
    r = torch.linalg.vector_norm(u, dim=-1, keepdim=True)
    outside = r > 1.0
    r_safe = torch.where(outside, r, torch.ones_like(r))
    u = torch.where(outside, (2.0 - 1.0 / r_safe) * (u / r_safe), u)
    return u * 0.25 + 0.5


class _DensityNet(tcnn.NetworkWithInputEncoding):
    """geo network whose kernel epilogue applies trunc_exp(y - shift) * selector."""

    def __init__(self, grid_cfg, exp_shift=0.0, seed=tcnn.DEFAULT_SEED, dtype=None):
        super().__init__(3, 1, grid_cfg, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                          "n_neurons": 64, "n_hidden_layers": 1}, seed=seed, dtype=dtype)
        self.mlp.output_activation = 'Exponential'
        self.mlp.exp_shift = exp_shift


class NGPNeRF(nn.Module):
    """Instant-NGP radiance field (ngp_nerf.py:68-198)."""

    def __init__(self, aabb: Union[torch.Tensor, List[float]], num_dim: int = 3, use_viewdirs: bool = False,
                 unbounded: bool = False, n_levels: int = 16, dtype=None, log2_hashmap_size: int = 18):
        super().__init__()
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32, device='cpu')
        self.register_buffer("aabb", aabb.float().cuda())
        self._aabb_host = [float(v) for v in aabb.reshape(-1).tolist()]
        self.num_dim = num_dim
        self.use_viewdirs = use_viewdirs
        self.unbounded = unbounded
        self.n_levels = n_levels
        self.dtype_name = dtype
        # (n_levels / log2_hashmap_size beyond the reference's 16 / 18: BASELINE config 5's tables sized to HBM)
        self.geo_mlp = _DensityNet(_grid_cfg(n_levels, log2_hashmap_size), dtype=dtype)
        self.app_mlp = tcnn.NetworkWithInputEncoding(
            3, 3, _grid_cfg(n_levels, log2_hashmap_size),
            {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid", "n_neurons": 64,
             "n_hidden_layers": 2}, dtype=dtype)

    # -- point queries (ngp_nerf.py:136-162).  Like the reference's, they normalise by the aabb whatever `unbounded` says: the
    #    reference stores that flag (ngp_nerf.py:90) and only NGPDensityField.forward (:251-252) ever contracts. ---------
    def query_density(self, x):
        shape = list(x.shape[:-1])
        x01, sel = ops.points_normalize(x.reshape(-1, 3).contiguous().float(), self._aabb_host)
        return field_apply(self.geo_mlp, x01, self.geo_mlp.params, sel).view(shape + [1])

    def query_rgb(self, x):
        shape = list(x.shape[:-1])
        x01, sel = ops.points_normalize(x.reshape(-1, 3).contiguous().float(), self._aabb_host)
        return field_apply(self.app_mlp, x01, self.app_mlp.params, sel).view(shape + [3])

    # -- ray-sample queries: positions o + d (t0+t1)/2 are formed in-kernel (nerf_renderer.py:125-127) --
    def sample_points(self, rays_o, rays_d, ray_indices, t_starts, t_ends):
        return ops.points_from_rays(rays_o, rays_d, ray_indices, t_starts, t_ends, self._aabb_host)

    def density_at(self, x01, sel, n_dev=None):
        return field_apply(self.geo_mlp, x01, self.geo_mlp.params, sel, n_dev)[:, 0]

    @torch.no_grad()
    def density_with_features(self, x01, sel, n_dev=None):
        """Density without gradient plus the level-major encoded features it was computed from -> (sigma [n], feat [L,n,2])."""
        net = self.geo_mlp
        sig, feat = ops.field_infer(net.grid, net.mlp, x01, sel, net.working_copy(), n_dev=n_dev, want_features=True)
        return sig[:, 0], feat

    def rgb_at(self, x01, sel, n_dev=None):
        return field_apply(self.app_mlp, x01, self.app_mlp.params, sel, n_dev)

    def density_rgb_at(self, x01, sel, geo_grad=True, app_grad=False):
        """sigma [n] and rgb [n,3] at the same points with ONE shared encode pass (both grids have the same geometry).
        geo_grad / app_grad = False detach the respective parameters (nerf_renderer.py:166-179 no_grad branches)."""
        pg = self.geo_mlp.params if geo_grad else self.geo_mlp.params.detach()
        pa = self.app_mlp.params if app_grad else self.app_mlp.params.detach()
        sig, rgb = _DualFieldFn.apply(x01, pg, pa, sel, self.geo_mlp, self.app_mlp)
        return sig[:, 0], rgb

    def forward(self, positions, directions=None, contract=None):
        if self.use_viewdirs and (directions is not None):
            assert positions.shape == directions.shape, f"{positions.shape} v.s. {directions.shape}"
        density = self.query_density(positions)
        rgb = self.query_rgb(positions)
        return rgb, density

    def reset_geo(self):
        """Fresh geometry network, identical initialisation every episode (ngp_nerf.py:178-197)."""
        self.geo_mlp = _DensityNet(_grid_cfg(16), dtype=self.dtype_name)


class InferenceNeRF:
    """NGPNeRF for inference only, for fields whose tables are sized to HBM (BASELINE config 5: L = 20, log2_hashmap_size
    28-30; SURVEY.md 8(e): "inference-only replicated fp16"): each network exists ONLY as its 16-bit working copy
    [MLP weights | table] -- no fp32 master, no optimizer state -- and is initialised on the device (tcnn's rule: Xavier-uniform
    MLP weights, tables U(-1e-4, 1e-4); the 10^9-entry tables are filled in chunks from a seeded device generator).  The duck
    type NeRFOCCRenderer uses (density_at / rgb_at / sample_points on kernel-made positions), like sharded.LevelShardedNeRF."""

    def __init__(self, aabb, n_levels=20, log2_hashmap_size=28, per_level_scale=PER_LEVEL_SCALE, dtype='fp16', seed=tcnn.DEFAULT_SEED,
                 table_scale=1e-4, density_bias=0.0, device=None, layout='tcnn', **layout_kw):
        """table_scale / density_bias (tests): tables U(-table_scale, table_scale) and sigma = exp(y + density_bias) instead of the
        fresh initialisation's near-constant sigma = exp(y ~ 0) -- a field with structure, dense enough for rays to terminate.
        layout: 'tcnn' (default) or the opt-in 'line_local' / 'line_overlap' table layouts of perf_amd.grid.GridConfig -- these fields exist for grids
        the reference never defines, no reference result constrains how their tables are laid out."""
        from .grid import GridConfig, MlpConfig
        import math
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32, device='cpu')
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.aabb = aabb.float().to(dev)
        self._aabb_host = [float(v) for v in aabb.reshape(-1).tolist()]
        self.training = False
        self.dtype_name = dtype
        self.grid = GridConfig(n_levels=n_levels, log2_hashmap_size=log2_hashmap_size, base_resolution=16, per_level_scale=per_level_scale,
                               layout=layout, **layout_kw)          # (layout_kw: sb_shift, local_min_res)
        t16 = ops.torch_dtype(dtype)
        self.nets = {}
        gen = torch.Generator(device=dev).manual_seed(seed)
        cg = torch.Generator(device='cpu').manual_seed(seed)
        for name, mlp in (('geo_mlp', MlpConfig(n_levels, 1, 1, 'Exponential', exp_shift=-float(density_bias))),
                          ('app_mlp', MlpConfig(n_levels, 2, 3, 'Sigmoid'))):
            n_net = mlp.n_params
            w16 = torch.empty(n_net + self.grid.n_params, dtype=t16, device=dev)
            parts = [(torch.rand(o * i, generator=cg, device='cpu') * 2 - 1) * math.sqrt(6.0 / (i + o)) for (o, i) in mlp.shapes]
            w16[:n_net].copy_(torch.cat(parts))
            chunk = 1 << 28
            for lo in range(n_net, w16.numel(), chunk):
                hi = min(lo + chunk, w16.numel())
                w16[lo:hi].copy_((torch.rand(hi - lo, device=dev, generator=gen) * 2 - 1) * table_scale)
            self.grid.canonicalize_(w16[n_net:])          # (line_overlap: the two copies of a run's shared vertex hold one value)
            self.nets[name] = (mlp, w16)

    def eval(self):
        return self

    def table_bytes(self):
        """Bytes of ONE encoder's 16-bit table."""
        return self.grid.n_params * 2

    @torch.no_grad()
    def density_at(self, x01, sel, n_dev=None):
        mlp, w16 = self.nets['geo_mlp']
        return ops.field_infer(self.grid, mlp, x01, sel, w16, n_dev=n_dev)[:, 0]

    @torch.no_grad()
    def rgb_at(self, x01, sel, n_dev=None):
        mlp, w16 = self.nets['app_mlp']
        return ops.field_infer(self.grid, mlp, x01, sel, w16, n_dev=n_dev)

    def sample_points(self, rays_o, rays_d, ray_indices, t_starts, t_ends):
        return ops.points_from_rays(rays_o, rays_d, ray_indices, t_starts, t_ends, self._aabb_host)


class NGPDensityField(nn.Module):
    """Proposal density field (ngp_nerf.py:200-265): sigma = trunc_exp(net(x) - 1) * selector."""

    def __init__(self, aabb, num_dim: int = 3, unbounded: bool = False, base_resolution: int = 16,
                 max_resolution: int = 128, n_levels: int = 5, log2_hashmap_size: int = 17, dtype=None):
        super().__init__()
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32, device='cpu')
        self.register_buffer("aabb", aabb.float().cuda())
        self._aabb_host = [float(v) for v in aabb.reshape(-1).tolist()]
        self.num_dim = num_dim
        self.unbounded = unbounded
        self.base_resolution = base_resolution
        self.max_resolution = max_resolution
        self.n_levels = n_levels
        self.log2_hashmap_size = log2_hashmap_size
        per_level_scale = np.exp((np.log(max_resolution) - np.log(base_resolution)) / (n_levels - 1)).tolist()
        self.mlp_base = _DensityNet(_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale),
                                    exp_shift=1.0, dtype=dtype)

    def forward(self, positions: torch.Tensor):
        shape = list(positions.shape[:-1])
        if self.unbounded:
            x01 = contract_to_unisphere(positions, self.aabb).reshape(-1, 3).contiguous().float()
            sel = ((x01 > 0.0) & (x01 < 1.0)).all(dim=-1).to(torch.uint8)
        else:
            x01, sel = ops.points_normalize(positions.reshape(-1, 3).contiguous().float(), self._aabb_host)
        return field_apply(self.mlp_base, x01, self.mlp_base.params, sel).view(shape + [1])
