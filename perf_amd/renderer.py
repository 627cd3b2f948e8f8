"""Host-side mirror of PeRF's NeRFOCCRenderer (modules/scene/nerf_renderer.py:105-209) on the gfx950 kernels.

Same class name, constructor and render() signature, same result dictionary.  Differences are fusions that do
not change results: sample positions and the selector are produced in-kernel; weights, opacity, distance and
colour come out of ONE compositing kernel (one wave per ray, no index_add_ atomics); in no-grad density paths
the sigmas evaluated for the visibility test inside sampling are reused instead of being recomputed
(the reference evaluates sigma_fn twice, :145-148 and :166-168 -- identical values).
"""
import torch
import torch.nn as nn

from . import ops
from .fields import NGPNeRF
from .nerfacc_impl import OccGridEstimator


class _VolumeRenderFn(torch.autograd.Function):
    """(sigmas, rgbs) -> (weights, trans, opacity, distance, colour); colour uses detached weights
    (nerf_renderer.py:183), so d colour / d sigma = 0 and d colour / d rgb_i = w_i."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, t_starts, t_ends, packed):
        ctx.set_materialize_grads(False)
        w, T, _, op, dist, col = ops.composite_fwd(sigmas, rgbs, t_starts, t_ends, packed)
        ctx.save_for_backward(sigmas, t_starts, t_ends, packed, w, T)
        return w, T, op, dist, col

    @staticmethod
    def backward(ctx, g_w, g_T, g_op, g_dist, g_col):
        sigmas, ts, te, packed, w, T = ctx.saved_tensors
        f = lambda g: None if g is None else g.contiguous().float()
        want_ds, want_dr = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        ds, dr = ops.composite_bwd(sigmas, ts, te, packed, w, T, g_weights=f(g_w), g_trans=f(g_T), g_opacity=f(g_op),
                                   g_distance=f(g_dist), g_color=f(g_col) if want_dr else None,
                                   want_dsigma=want_ds, want_drgb=want_dr)
        return ds, dr, None, None, None


volume_render = _VolumeRenderFn.apply


class NeRFOCCRenderer(nn.Module):
    def __init__(self, max_radius, bg_color):
        super().__init__()
        self.max_radius = max_radius
        self.bg_color = bg_color
        assert self.bg_color in ['rand_noise', 'black', 'white']
        # sampling constants hard-coded by the reference (nerf_renderer.py:149-154)
        self.near_plane = 0.
        self.far_plane = 1.5
        self.render_step_size = 5e-4
        self.early_stop_eps = 1e-4
        self.max_steps = None          # None: ceil((far-near)/step)+1 like the reference; an int fixes the count
        self.sample_capacity = None    # int: sync-free sampling -- arrays of that many rows, live counts on the device
        # two-phase early termination of the sync-free sampler (None = one phase): density on the first `head_samples`
        # samples of every ray, then on the rest of the rays still alive; identical results, far fewer density
        # evaluations once the scene is opaque (nerfacc_impl.OccGridEstimator.sampling_ex).  2 is the smallest head that can end
        # a ray (the first sample of a ray is always kept; the second is dropped iff the first made the ray opaque) and the
        # fastest on a trained scene (512x1024 frames/s: K=2 1140, 3 1070, 4 1010, 6 880; untrained scenes do not care)
        self.head_samples = 2
        # marching lattice: 'repeated' (t_{k+1} = fl(t_k + step): how nerfacc's traverse_grids is understood to march -- the
        # default since round 4, in the oracle too) or 'single' (t_k = fl(t0 + fl(k step)): one rounding per sample, rounds
        # 1-3); include/perf_hip.h PERF_LATTICE_*.  tools/pin_upstream.py lets a maintainer who holds nerfacc check it.
        self.lattice = 'repeated'

    # The render is cut in two stages so that a data-parallel trainer can overlap the gradient all-reduce of step k
    # with everything of step k+1 that does not depend on the parameters being updated (scene.py).
    def stage_sample(self, nerf: NGPNeRF, estimator: OccGridEstimator, rays_o, rays_d, rand=None, with_rgb=False,
                     keep_features=False):
        """Sampling (marching, the no-grad density pass and visibility compaction of nerf_renderer.py:145-155), sample
        positions and -- with_rgb -- the colour field without gradient.  Returns a dict consumed by stage_composite, or
        None when the batch has no sample.  With self.sample_capacity set every per-sample array has that many rows and
        st['n_dev'] (device int64 [1]) holds the live count: no host read-back, hipGraph-capturable.
        keep_features (sync-free mode): st['feat0'] = the density field's encoded features of the KEPT samples, compacted from
        the sampler's own density pass."""
        rays_o = rays_o.contiguous().float(); rays_d = rays_d.contiguous().float()
        rand = rand or {}

        def sigma_points_fn(x01, sel, n_dev):
            if keep_features:
                return nerf.density_with_features(x01, sel, n_dev)
            return nerf.density_at(x01, sel, n_dev)

        sm = estimator.sampling_ex(
            rays_o, rays_d, sigma_points_fn=sigma_points_fn, near_plane=self.near_plane, far_plane=self.far_plane,
            render_step_size=self.render_step_size, early_stop_eps=self.early_stop_eps, stratified=nerf.training,
            cone_angle=0., alpha_thre=0., jitter=rand.get('jitter'), max_steps=self.max_steps, capacity=self.sample_capacity,
            points_aabb=nerf._aabb_host, head_samples=self.head_samples, lattice=self.lattice)
        if sm.n_dev is None and sm.ray_indices.numel() <= 0:
            return None
        x01, sel = sm.x01, sm.sel
        if x01 is None:
            x01, sel = nerf.sample_points(rays_o, rays_d, sm.ray_indices, sm.t_starts, sm.t_ends)
        st = {'ray_indices': sm.ray_indices, 't_starts': sm.t_starts, 't_ends': sm.t_ends, 'packed': sm.packed, 'sig0': sm.sig,
              'x01': x01, 'sel': sel, 'n_rays': rays_o.shape[0], 'rgbs': None, 'n_dev': sm.n_dev,
              'n_marched_dev': sm.n_marched_dev, 'feat0': sm.feat}
        if with_rgb:
            with torch.no_grad():
                st['rgbs'] = nerf.rgb_at(x01, sel, sm.n_dev)
        return st

    def stage_composite(self, nerf: NGPNeRF, st, geo_inference=False, app_inference=False, rand=None):
        rand = rand or {}
        n_rays = st['n_rays']
        x01, sel, packed = st['x01'], st['sel'], st['packed']
        dev = x01.device
        grad_geo = torch.is_grad_enabled() and not geo_inference
        grad_app = torch.is_grad_enabled() and not app_inference
        # (A shared-index pass over both grids -- perf_hashgrid_fwd2 -- measured 2x SLOWER than two passes: the
        #  per-XCD working set doubles to 4 MiB = the whole L2.  The fields are therefore queried one after the other.)
        n_dev = st.get('n_dev')
        if grad_geo:
            sigmas = nerf.density_at(x01, sel, n_dev)
        elif st['sig0'] is not None:
            sigmas = st['sig0']                              # same values the reference recomputes under no_grad
        else:
            with torch.no_grad():
                sigmas = nerf.density_at(x01, sel, n_dev)
        if st['rgbs'] is not None and not grad_app:
            rgbs = st['rgbs']
        else:
            with torch.set_grad_enabled(grad_app):
                rgbs = nerf.rgb_at(x01, sel, n_dev)

        weights, trans, opacities, distances, colors = volume_render(sigmas, rgbs, st['t_starts'], st['t_ends'], packed)

        if nerf.training:
            # (the background colour is only read by the training composite: an eval frame does not draw / fill 3 floats per ray for nothing)
            if self.bg_color == 'rand_noise':
                bg_color = rand['bg'] if 'bg' in rand else torch.rand(n_rays, 3, device=dev)
            elif self.bg_color == 'white':
                bg_color = torch.ones(n_rays, 3, device=dev)
            else:
                bg_color = torch.zeros(n_rays, 3, device=dev)
            noise = rand['noise'] if 'noise' in rand else torch.rand_like(distances)
            distances = torch.relu(distances + (noise * 2. - 1.) * (1. - opacities))
            colors = colors + bg_color * (1. - opacities).detach()
        elif torch.is_grad_enabled():
            distances = distances + 5. * (1. - opacities).detach()
            colors = colors + .5 * (1. - opacities).detach()
        else:
            ops.render_finish_eval(opacities, distances, colors, n_dev)      # same arithmetic, one launch, in place

        # (capacity mode: per-sample arrays have sample_capacity rows, the first n_samples_dev of them are live)
        return {'is_valid': True, 'rgb': colors, 'distance': distances, 'weights': weights, 'opacities': opacities,
                'trans': trans, 't_starts': st['t_starts'], 't_ends': st['t_ends'], 'ray_indices': st['ray_indices'],
                'packed_info': packed, 'n_samples_dev': n_dev, 'n_marched_dev': st.get('n_marched_dev')}

    def render(self, nerf: NGPNeRF, estimator: OccGridEstimator, rays_o, rays_d, near, far,
               geo_inference=False, app_inference=False, rand=None):
        """`rand` (optional dict with 'jitter' [R], 'bg' [R,3], 'noise' [R,1]) injects the random draws of
        :152,:185,:193 for tests; by default they are drawn with torch.rand on the device in that order."""
        assert near.shape[-1] == 1 and len(near.shape) == 2
        n_rays = rays_o.shape[0]
        dev = rays_o.device
        st = self.stage_sample(nerf, estimator, rays_o, rays_d, rand)
        if st is None:
            return {'is_valid': False, 'rgb': torch.zeros(n_rays, 3, device=dev), 'distance': torch.zeros(n_rays, 1, device=dev),
                    'opacities': torch.zeros(n_rays, 1, device=dev)}
        return self.stage_composite(nerf, st, geo_inference, app_inference, rand)


class NeRFPropRenderer(nn.Module):
    """Mirror of modules/scene/nerf_renderer.py:10-102 (proposal-network renderer).  That path is dead in the
    reference (render_weight_from_alpha is never imported, :73); this is the inference-only counterpart built on the
    hierarchical resampling kernel: 128 -> 64 proposal samples, 64 final samples, near 1e-2, far 2, dense layout."""

    def __init__(self, max_radius, bg_color):
        super().__init__()
        self.max_radius = max_radius
        self.bg_color = bg_color
        assert self.bg_color in ['rand_noise', 'black', 'white']
        self.n_samples = 64
        self.n_samples_per_prop = [128, 64]

    @torch.no_grad()
    def render(self, nerf, prop_networks, estimator, rays_o, rays_d, near, far, sampling_requires_grad=False, taus=None):
        from .nerfacc_impl import render_weight_from_alpha
        n_rays = rays_o.shape[0]
        dev = rays_o.device

        def positions(t_starts, t_ends):
            return rays_o[:, None, :] + rays_d[:, None, :] * (t_starts + t_ends)[..., None] / 2.0

        def prop_sigma_fn(t_starts, t_ends, net):
            sig = net(positions(t_starts, t_ends)).squeeze(-1)
            sig[..., -1] = torch.inf                                     # nerf_renderer.py:43
            return sig

        t_starts, t_ends = estimator.sampling(
            prop_sigma_fns=[lambda a, b, p=p: prop_sigma_fn(a, b, p) for p in prop_networks],
            prop_samples=self.n_samples_per_prop, num_samples=self.n_samples, n_rays=n_rays, near_plane=1e-2, far_plane=2.,
            sampling_type='uniform', stratified=nerf.training, requires_grad=sampling_requires_grad, taus=taus)
        pos = positions(t_starts, t_ends)
        rgb, density = nerf(pos.reshape(-1, 3))
        rgb = rgb.reshape(n_rays, -1, 3); density = density.reshape(n_rays, -1)
        alphas = 1. - torch.exp(-density * (t_ends - t_starts))
        weights, trans = render_weight_from_alpha(alphas)
        colors = (weights[..., None] * rgb).sum(1)
        opacities = weights.sum(1, keepdim=True)
        mid = (t_starts + t_ends) / 2.0
        distances = (weights * mid).sum(1, keepdim=True)
        if self.bg_color == 'rand_noise':
            bg = torch.rand(n_rays, 3, device=dev)
        elif self.bg_color == 'white':
            bg = torch.ones(n_rays, 3, device=dev)
        else:
            bg = torch.zeros(n_rays, 3, device=dev)
        colors = colors + bg * (1.0 - opacities)
        distances = distances + torch.rand_like(distances) * (1. - opacities)
        return {'rgb': colors, 'distance': distances, 'weights': weights, 'opacities': opacities, 'trans': trans,
                't_starts': t_starts, 't_ends': t_ends, 'sampled_pts': pos}
