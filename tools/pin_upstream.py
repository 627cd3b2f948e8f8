#!/usr/bin/env python
"""Pin the third-party arithmetic of the hot path on the REAL packages.

PeRF's hash grid + fused MLP are tinycudann==1.7 (requirements.txt:34; call sites modules/fields/ngp_nerf.py:96-134,
modules/geo_predictors/pano_joint_predictor.py:30-41), its marching / compositing nerfacc==0.5.3 (requirements.txt:16;
modules/scene/nerf_renderer.py:145-183) and its distortion loss torch_efficient_distloss==0.1.3 (requirements.txt:36;
modules/scene/nerf.py:230).  None of them is in /root/reference or installable in the build container, so this repository's
oracle restates them from their published algorithms and every parity claim about them is "unpinned" (DESIGN.md 2).

A maintainer who HAS the packages (any CUDA box with the reference's environment) closes that gap with one run:

    python tools/pin_upstream.py                      # -> tests/golden/upstream_{tcnn,nerfacc,distloss}.npz

It imports whatever `tinycudann`, `nerfacc` and `torch_efficient_distloss` resolve to, feeds them SEEDED inputs -- flat
fp32 `params` in tcnn's own layout ([network | grid], which is also this repository's), points, rays, an occupancy grid,
densities -- and records their outputs and gradients.  Commit the three files: `pytest -m gpu tests/test_gpu_upstream.py`
then compares this repository's HIP path with them (while the files are absent the same tests RUN over vectors the script
produces from an oracle-backed stand-in, tests/upstream_standin.py: format, seeded inputs and checkers are exercised, nothing is
pinned by that), and
`tests/test_cpu_oracle.py::test_oracle_against_upstream_vectors` does the same for the oracle -- including WHICH marching
lattice upstream walks (include/perf_hip.h PERF_LATTICE_*).

Nothing here imports this repository: the script is self-contained on purpose (it has to run inside the reference's
environment).  `main(modules=...)` takes the three modules as arguments instead of importing them -- how the build
container proves the format on a stand-in (tests/upstream_standin.py, tests/test_cpu_oracle.py).
"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

FORMAT_VERSION = 1
AABB = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]

# modules/fields/ngp_nerf.py:94-134
PER_LEVEL_SCALE = 1.4472692012786865
ENC = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 18, "base_resolution": 16,
       "per_level_scale": PER_LEVEL_SCALE}
NETS = {
    'geo': (1, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}),
    'app': (3, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid", "n_neurons": 64, "n_hidden_layers": 2}),
}
# modules/geo_predictors/pano_joint_predictor.py:22-41 (n_levels 16, T 19, 16 -> 2048, Smoothstep)
ENC_SMOOTH = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
              "per_level_scale": float(np.exp(np.log(2048 / 16) / 15)), "interpolation": "Smoothstep"}


def seeded_params(n, n_net, seed):
    """The flat parameter vector both sides load: network part U(-0.25, 0.25), grid part U(-1, 1) (non-trivial features),
    from numpy's PCG64 stream -- reproducible from (n, n_net, seed) alone, so the multi-megabyte vector is not stored."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p = rng.random(n, dtype=np.float32) * 2.0 - 1.0
    p[:n_net] *= 0.25
    return p


def seeded_points(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((n, 3), dtype=np.float32) * 0.98 + 0.01


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _sparse(g):
    idx = np.nonzero(g)[0].astype(np.int64)
    return idx, g[idx].astype(np.float32)


def dump_tcnn(tcnn, device, n_points=1024):
    out = {'format': FORMAT_VERSION, 'n_points': n_points}
    for name, (n_out, net_cfg) in NETS.items():
        m = tcnn.NetworkWithInputEncoding(n_input_dims=3, n_output_dims=n_out, encoding_config=dict(ENC), network_config=dict(net_cfg))
        m = m.to(device)
        n = m.params.numel()
        n_grid = 2 * grid_entries(ENC)
        n_net = n - n_grid
        seed = {'geo': 101, 'app': 202}[name]
        p = seeded_params(n, n_net, seed)
        with torch.no_grad():
            m.params.copy_(torch.from_numpy(p).to(device))
        x = seeded_points(n_points, seed + 1)
        dy = np.random.Generator(np.random.PCG64(seed + 2)).standard_normal((n_points, n_out)).astype(np.float32)
        xt = torch.from_numpy(x).to(device)
        y = m(xt)
        (y.float() * torch.from_numpy(dy).to(device)).sum().backward()
        g = m.params.grad.detach().float().cpu().numpy()
        gi, gv = _sparse(g[n_net:])
        out.update({f'{name}_n_params': n, f'{name}_n_net': n_net, f'{name}_seed': seed, f'{name}_params_sha256': _sha(p),
                    f'{name}_x': x, f'{name}_dy': dy, f'{name}_y': y.detach().float().cpu().numpy(), f'{name}_y_dtype': str(y.dtype),
                    f'{name}_grad_net': g[:n_net].astype(np.float32), f'{name}_grad_grid_idx': gi, f'{name}_grad_grid_val': gv})
    # tcnn.Encoding, Smoothstep: value, input gradient, and the second-order path SphereDistanceField takes
    # (pano_joint_predictor.py:58-67: autograd.grad(..., create_graph=True) followed by a backward through it)
    e = tcnn.Encoding(n_input_dims=3, encoding_config=dict(ENC_SMOOTH)).to(device)
    n = e.params.numel()
    p = seeded_params(n, 0, 303)
    with torch.no_grad():
        e.params.copy_(torch.from_numpy(p).to(device))
    x = seeded_points(256, 304)
    xt = torch.from_numpy(x).to(device).requires_grad_(True)
    f = e(xt).float()
    wv = torch.from_numpy(np.random.Generator(np.random.PCG64(305)).standard_normal((256, f.shape[1])).astype(np.float32)).to(device)
    gx, = torch.autograd.grad((f * wv).sum(), xt, create_graph=True)
    out.update({'enc_n_params': n, 'enc_seed': 303, 'enc_params_sha256': _sha(p), 'enc_x': x, 'enc_w': wv.cpu().numpy(),
                'enc_y': f.detach().cpu().numpy(), 'enc_dx': gx.detach().float().cpu().numpy()})
    try:
        (gx.float() ** 2).sum().backward()
        g2 = e.params.grad.detach().float().cpu().numpy()
        gi, gv = _sparse(g2)
        out.update({'enc_dd_x': xt.grad.detach().float().cpu().numpy(), 'enc_dd_grid_idx': gi, 'enc_dd_grid_val': gv})
    except RuntimeError as err:          # (a tcnn build without double backward support)
        out['enc_dd_error'] = str(err)
    return out


def grid_entries(enc):
    """Entries of a tcnn HashGrid (public rule: scale_l = N_min b^l - 1, res_l = ceil(scale_l) + 1, size_l = min(align8(res_l^3), 2^T))."""
    total = 0
    log2_b = np.float32(np.log2(np.float32(enc['per_level_scale'])))
    for l in range(enc['n_levels']):
        s = np.float32(np.float32(np.exp2(np.float64(np.float32(l) * log2_b))) * np.float32(enc['base_resolution'])) - np.float32(1.0)
        r = int(np.ceil(float(s))) + 1
        n = min((min(r ** 3, 0x7fffffff) + 7) // 8 * 8, 1 << enc['log2_hashmap_size'])
        total += n
    return total


def _sigma_analytic(x):
    return 40.0 * torch.exp(-6.0 * (x * x).sum(-1))


def dump_nerfacc(nerfacc, OccGridEstimator, device, n_rays=64, res=32):
    rng = np.random.Generator(np.random.PCG64(404))
    occ = rng.random((res, res, res)) < 0.35
    o = (rng.random((n_rays, 3), dtype=np.float32) - 0.5) * 0.6
    d = rng.standard_normal((n_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    est = OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=res, levels=1).to(device)
    with torch.no_grad():
        est.binaries.copy_(torch.from_numpy(occ).reshape(est.binaries.shape).to(device))
        est.occs.copy_(torch.from_numpy(occ.reshape(-1).astype(np.float32)).to(device))
    est.eval()
    ot, dt = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    out = {'format': FORMAT_VERSION, 'res': res, 'binaries': np.packbits(occ.reshape(-1)), 'o': o, 'd': d, 'aabb': np.asarray(AABB, np.float32)}
    for tag, step in (('coarse', 4e-3), ('perf', 5e-4)):                  # PeRF passes 5e-4 (nerf_renderer.py:149)
        ri, ts, te = est.sampling(ot, dt, sigma_fn=None, near_plane=0.0, far_plane=1.5, render_step_size=step, stratified=False,
                                  cone_angle=0.0, alpha_thre=0.0)
        out.update({f'{tag}_step': np.float32(step), f'{tag}_ray_indices': ri.cpu().numpy().astype(np.int64),
                    f'{tag}_t_starts': ts.cpu().numpy(), f'{tag}_t_ends': te.cpu().numpy()})

    def sigma_fn(t_starts, t_ends, ray_indices):
        x = ot[ray_indices] + dt[ray_indices] * ((t_starts + t_ends)[:, None] / 2.0)
        return _sigma_analytic(x)
    ri, ts, te = est.sampling(ot, dt, sigma_fn=sigma_fn, near_plane=0.0, far_plane=1.5, render_step_size=4e-3, stratified=False,
                              cone_angle=0.0, alpha_thre=0.0)                  # default early_stop_eps = 1e-4, as PeRF leaves it
    sig = sigma_fn(ts, te, ri)
    w, T, al = nerfacc.render_weight_from_density(ts, te, sig, ray_indices=ri, n_rays=n_rays)
    vals = torch.from_numpy(rng.random((ri.numel(), 3), dtype=np.float32)).to(device)
    acc = nerfacc.accumulate_along_rays(w, vals, ray_indices=ri, n_rays=n_rays)
    opa = nerfacc.accumulate_along_rays(w, None, ray_indices=ri, n_rays=n_rays)
    out.update({'vis_ray_indices': ri.cpu().numpy().astype(np.int64), 'vis_t_starts': ts.cpu().numpy(), 'vis_t_ends': te.cpu().numpy(),
                'vis_sigmas': sig.cpu().numpy(), 'weights': w.cpu().numpy(), 'trans': T.cpu().numpy(), 'alphas': al.cpu().numpy(),
                'values': vals.cpu().numpy(), 'accumulated': acc.cpu().numpy(), 'opacity': opa.cpu().numpy()})
    return out


def dump_distloss(mod, device, n_rays=48):
    rng = np.random.Generator(np.random.PCG64(505))
    counts = rng.integers(0, 24, n_rays)
    counts[-1] = max(counts[-1], 1)                        # (n_rays = ray_id.max() + 1)
    ray_id = np.repeat(np.arange(n_rays), counts).astype(np.int64)
    n = ray_id.size
    interval = np.full(n, 4e-3, np.float32)
    first = rng.random(n_rays, dtype=np.float32) * 0.5
    m = np.concatenate([first[r] + 4e-3 * (np.arange(c) + 0.5) for r, c in enumerate(counts)]).astype(np.float32)
    w = (rng.random(n, dtype=np.float32) * 0.2).astype(np.float32)
    wt = torch.from_numpy(w).to(device).requires_grad_(True)
    loss = mod.flatten_eff_distloss(wt, torch.from_numpy(m).to(device), torch.from_numpy(interval).to(device), torch.from_numpy(ray_id).to(device))
    loss.backward()
    return {'format': FORMAT_VERSION, 'w': w, 'm': m, 'interval': interval, 'ray_id': ray_id, 'loss': np.float32(loss.item()),
            'grad_w': wt.grad.detach().cpu().numpy()}


def main(modules=None, out_dir=None, device=None, backend=None):
    if modules is None:
        import tinycudann as tcnn
        import nerfacc
        import torch_efficient_distloss as ted
        from nerfacc.estimators.occ_grid import OccGridEstimator
        backend = backend or 'upstream: tinycudann %s, nerfacc %s' % (getattr(tcnn, '__version__', '?'), getattr(nerfacc, '__version__', '?'))
    else:
        tcnn, nerfacc, OccGridEstimator, ted = modules
    device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
    out_dir = out_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    files = {}
    for name, data in (('upstream_tcnn', dump_tcnn(tcnn, device)), ('upstream_nerfacc', dump_nerfacc(nerfacc, OccGridEstimator, device)),
                       ('upstream_distloss', dump_distloss(ted, device))):
        data['backend'] = str(backend or 'unknown')
        path = os.path.join(out_dir, name + '.npz')
        np.savez_compressed(path, **data)
        files[name] = path
        print(f'{path}: {os.path.getsize(path)} bytes, {len(data)} arrays')
    return files


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--out', default=None, help='output directory (default: tests/golden of this checkout)')
    ap.add_argument('--device', default=None)
    a = ap.parse_args()
    main(out_dir=a.out, device=a.device)
    sys.exit(0)
