"""A CPU stand-in for the three third-party packages tools/pin_upstream.py drives -- `tinycudann`, `nerfacc`,
`torch_efficient_distloss` -- with exactly the surface that script touches, backed by oracle/perf_oracle.py (test infrastructure).
The build container has none of the packages: running the pinning script over this stand-in proves the file format and the
checkers (tests/upstream_check.py) end to end; the vectors it yields are the oracle's own numbers, NOT a pin."""
import types

import numpy as np
import torch
import torch.nn as nn

from oracle import perf_oracle as O


def _levels(cfg):
    return O.grid_levels(n_levels=int(cfg['n_levels']), n_feat=int(cfg.get('n_features_per_level', 2)),
                         log2_hashmap_size=int(cfg['log2_hashmap_size']), base_resolution=int(cfg['base_resolution']),
                         per_level_scale=float(cfg['per_level_scale']))


class NetworkWithInputEncoding(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config):
        super().__init__()
        self.spec = O.FieldSpec(_levels(encoding_config), int(network_config['n_hidden_layers']), int(n_output_dims),
                                network_config.get('output_activation', 'None'))
        self.params = nn.Parameter(torch.zeros(self.spec.n_params))

    def forward(self, x):
        return O.network_with_encoding(x, self.params, self.spec)


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config):
        super().__init__()
        self.lv = _levels(encoding_config)
        self.interp = encoding_config.get('interpolation', 'Linear')
        self.params = nn.Parameter(torch.zeros(self.lv.n_params))

    def forward(self, x):
        return O.hashgrid_encode(x, self.params.view(self.lv.total, self.lv.n_feat), self.lv, interpolation=self.interp)


class OccGridEstimator(nn.Module):
    def __init__(self, roi_aabb, resolution=128, levels=1):
        super().__init__()
        self.res = int(resolution)
        self.register_buffer('aabbs', torch.as_tensor(roi_aabb, dtype=torch.float32).reshape(1, 6))
        self.register_buffer('occs', torch.zeros(self.res ** 3))
        self.register_buffer('binaries', torch.zeros(1, self.res, self.res, self.res, dtype=torch.bool))

    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn=None, near_plane=0.0, far_plane=1e10, render_step_size=1e-3, early_stop_eps=1e-4,
                 alpha_thre=0.0, stratified=False, cone_angle=0.0):
        assert not stratified and cone_angle == 0.0
        aabb = self.aabbs[0].numpy()
        ri, ts, te, packed = O.occ_march(rays_o.numpy(), rays_d.numpy(), self.binaries[0].numpy(), aabb, near_plane, far_plane,
                                         render_step_size)
        ri, ts, te = torch.from_numpy(ri), torch.from_numpy(ts), torch.from_numpy(te)
        if sigma_fn is not None and ri.numel():
            sig = sigma_fn(ts, te, ri)
            keep, _ = O.visibility_keep_mask(sig.numpy(), ts.numpy(), te.numpy(), packed, early_stop_eps)
            keep = torch.from_numpy(keep)
            ri, ts, te = ri[keep], ts[keep], te[keep]
        return ri, ts, te


def render_weight_from_density(t_starts, t_ends, sigmas, ray_indices=None, n_rays=None):
    packed = O.packed_info_from_ray_indices(ray_indices.numpy(), n_rays)
    return O.render_weight_from_density(t_starts, t_ends, sigmas, packed)


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    return O.accumulate_along_rays(weights, values, ray_indices, n_rays)


def modules():
    """(tinycudann, nerfacc, OccGridEstimator, torch_efficient_distloss) as tools/pin_upstream.py:main(modules=...) takes them."""
    tcnn = types.SimpleNamespace(NetworkWithInputEncoding=NetworkWithInputEncoding, Encoding=Encoding, __version__='stand-in')
    nerfacc = types.SimpleNamespace(render_weight_from_density=render_weight_from_density, accumulate_along_rays=accumulate_along_rays,
                                    __version__='stand-in')
    ted = types.SimpleNamespace(flatten_eff_distloss=O.flatten_eff_distloss)
    return tcnn, nerfacc, OccGridEstimator, ted
