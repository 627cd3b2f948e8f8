"""Checkers of the vectors tools/pin_upstream.py writes (tests/golden/upstream_{tcnn,nerfacc,distloss}.npz): the oracle
(CPU) and the HIP path (GPU) evaluated on the recorded inputs and compared with the recorded outputs.  The tolerances depend
on who produced the vectors (`backend`): the oracle stand-in (fp32: its own numbers must come back exactly / to fp32
rounding), or the real packages (tiny-cuda-nn computes in fp16: 16-bit tolerances; stated where they are used)."""
import numpy as np
import torch

from oracle import perf_oracle as O
from tools import pin_upstream as PU


def is_upstream(npz):
    return str(npz['backend']).startswith('upstream')


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _params(npz, name, n_net_key=None):
    n = int(npz[f'{name}_n_params'])
    n_net = int(npz[n_net_key]) if n_net_key else 0
    p = PU.seeded_params(n, n_net, int(npz[f'{name}_seed']))
    assert PU._sha(p) == str(npz[f'{name}_params_sha256']), 'numpy reproduced a different parameter vector: regenerate the file here'
    return p, n_net


def _dense_grid_grad(npz, name, n):
    g = np.zeros(n, np.float32)
    g[npz[f'{name}_idx']] = npz[f'{name}_val']
    return g


# ---- the oracle against the vectors (CPU) ---------------------------------------------------------------------------------------
def oracle_vs_tcnn(npz):
    up = is_upstream(npz)
    rep = {}
    for name, (n_out, net_cfg) in PU.NETS.items():
        spec = O.geo_spec() if name == 'geo' else O.app_spec()
        p, n_net = _params(npz, name, f'{name}_n_net')
        assert n_net == spec.n_net and p.size == spec.n_params, 'tcnn lays its parameters out differently than SURVEY.md A.2 says'
        pt = torch.from_numpy(p).requires_grad_(True)
        x = torch.from_numpy(npz[f'{name}_x'])
        y = O.network_with_encoding(x, pt, spec, quant='fp16' if up else None)
        (y * torch.from_numpy(npz[f'{name}_dy'])).sum().backward()
        g = pt.grad.numpy()
        rep[name] = {'y': _rel(y.detach().numpy(), npz[f'{name}_y']), 'grad_net': _rel(g[:n_net], npz[f'{name}_grad_net']),
                     'grad_grid': _rel(g[n_net:], _dense_grid_grad(npz, f'{name}_grad_grid', p.size - n_net))}
        tol = 2e-2 if up else 1e-6                  # upstream: fp16 storage AND fp16 accumulation inside the fused MLP
        assert max(rep[name].values()) <= tol, (name, rep[name])
    lv = O.grid_levels(16, 2, 19, 16, PU.ENC_SMOOTH['per_level_scale'])
    p, _ = _params(npz, 'enc')
    pt = torch.from_numpy(p).requires_grad_(True)
    x = torch.from_numpy(npz['enc_x']).requires_grad_(True)
    f = O.hashgrid_encode(x, pt.view(lv.total, 2), lv, interpolation='Smoothstep', quant='fp16' if up else None)
    gx, = torch.autograd.grad((f * torch.from_numpy(npz['enc_w'])).sum(), x, create_graph=True)
    rep['enc'] = {'y': _rel(f.detach().numpy(), npz['enc_y']), 'dx': _rel(gx.detach().numpy(), npz['enc_dx'])}
    if 'enc_dd_x' in npz.files:
        (gx ** 2).sum().backward()
        rep['enc']['dd_x'] = _rel(x.grad.numpy(), npz['enc_dd_x'])
        rep['enc']['dd_grid'] = _rel(pt.grad.numpy(), _dense_grid_grad(npz, 'enc_dd_grid', p.size))
    assert max(rep['enc'].values()) <= (2e-2 if up else 1e-5), rep['enc']
    return rep


def _binaries(npz):
    res = int(npz['res'])
    return np.unpackbits(npz['binaries'])[:res ** 3].astype(bool).reshape(res, res, res)


def lattice_verdict(npz):
    """Which marching lattice the recorded samples walk: {'coarse' / 'perf': {'repeated': equal?, 'single': equal?}} -- equality of
    ray_indices AND bitwise equality of t_starts / t_ends."""
    occ = _binaries(npz)
    out = {}
    for tag in ('coarse', 'perf'):
        out[tag] = {}
        for lat in ('repeated', 'single'):
            ri, ts, te, _ = O.occ_march(npz['o'], npz['d'], occ, npz['aabb'], 0.0, 1.5, float(npz[f'{tag}_step']), lattice=lat)
            out[tag][lat] = bool(ri.size == npz[f'{tag}_ray_indices'].size and np.array_equal(ri, npz[f'{tag}_ray_indices'])
                                 and np.array_equal(ts, npz[f'{tag}_t_starts']) and np.array_equal(te, npz[f'{tag}_t_ends']))
    return out


def oracle_vs_nerfacc(npz):
    verdict = lattice_verdict(npz)
    # THE question the file answers: the default lattice must be the one upstream walks, to the bit
    assert verdict['coarse'][O.DEFAULT_LATTICE] and verdict['perf'][O.DEFAULT_LATTICE], \
        f'the recorded samples do not walk the default lattice ({O.DEFAULT_LATTICE}): {verdict} -- switch DEFAULT_LATTICE (oracle, perf_amd/_lib.py) if the other one matches'
    ri, ts, te, sig = (torch.from_numpy(npz[k]) for k in ('vis_ray_indices', 'vis_t_starts', 'vis_t_ends', 'vis_sigmas'))
    R = npz['o'].shape[0]
    occ = _binaries(npz)
    o, d = torch.from_numpy(npz['o']), torch.from_numpy(npz['d'])
    mri, mts, mte, packed = O.occ_march(npz['o'], npz['d'], occ, npz['aabb'], 0.0, 1.5, 4e-3)
    x = o[torch.from_numpy(mri)] + d[torch.from_numpy(mri)] * ((torch.from_numpy(mts) + torch.from_numpy(mte))[:, None] / 2.0)
    keep, _ = O.visibility_keep_mask(PU._sigma_analytic(x).numpy(), mts, mte, packed, 1e-4)
    assert np.array_equal(mri[keep], npz['vis_ray_indices']) and np.array_equal(mts[keep], npz['vis_t_starts']), 'early termination keeps another set'
    pk = O.packed_info_from_ray_indices(npz['vis_ray_indices'], R)
    w, T, al = O.render_weight_from_density(ts, te, sig, pk)
    acc = O.accumulate_along_rays(w, torch.from_numpy(npz['values']), ri, R)
    rep = {'weights': _rel(w.numpy(), npz['weights']), 'trans': _rel(T.numpy(), npz['trans']), 'alphas': _rel(al.numpy(), npz['alphas']),
           'accumulated': _rel(acc.numpy(), npz['accumulated']), 'opacity': _rel(O.accumulate_along_rays(w, None, ri, R).numpy(), npz['opacity'])}
    assert max(rep.values()) <= 2e-6, rep
    return {'lattice': verdict, **rep}


def oracle_vs_distloss(npz):
    w = torch.from_numpy(npz['w']).requires_grad_(True)
    loss = O.flatten_eff_distloss(w, torch.from_numpy(npz['m']), torch.from_numpy(npz['interval']), torch.from_numpy(npz['ray_id']))
    loss.backward()
    rep = {'loss': abs(float(loss) - float(npz['loss'])) / abs(float(npz['loss'])), 'grad_w': _rel(w.grad.numpy(), npz['grad_w'])}
    assert max(rep.values()) <= 1e-5, rep
    return rep


# ---- the HIP path against the vectors (GPU) ----------------------------------------------------------------------------------------
def hip_vs_tcnn(npz, dtype='fp16'):
    from perf_amd import tcnn
    up = is_upstream(npz)
    rep = {}
    for name, (n_out, net_cfg) in PU.NETS.items():
        m = tcnn.NetworkWithInputEncoding(3, n_out, dict(PU.ENC), dict(net_cfg), dtype=dtype)
        p, n_net = _params(npz, name, f'{name}_n_net')
        assert m.params.numel() == p.size and m.mlp.n_params == n_net
        with torch.no_grad():
            m.params.copy_(torch.from_numpy(p).cuda())
        y = m(torch.from_numpy(npz[f'{name}_x']).cuda())
        (y.float() * torch.from_numpy(npz[f'{name}_dy']).cuda()).sum().backward()
        g = m.params.grad.float().cpu().numpy()
        rep[name] = {'y': _rel(y.detach().float().cpu().numpy(), npz[f'{name}_y']), 'grad_net': _rel(g[:n_net], npz[f'{name}_grad_net']),
                     'grad_grid': _rel(g[n_net:], _dense_grid_grad(npz, f'{name}_grad_grid', p.size - n_net))}
        # What 16-bit storage of table, weights and activations costs on THESE inputs is measured, not guessed: the oracle's
        # 16-bit emulation (operands rounded to the type, fp32 accumulation -- what the MFMA kernels do) against the same
        # vectors.  The HIP path may be off by twice that (+ 2e-3); against the emulation itself it must be tight.
        spec = O.geo_spec() if name == 'geo' else O.app_spec()
        pt = torch.from_numpy(p).requires_grad_(True)
        yq = O.network_with_encoding(torch.from_numpy(npz[f'{name}_x']), pt, spec, quant=dtype)
        (yq * torch.from_numpy(npz[f'{name}_dy'])).sum().backward()
        gq = pt.grad.numpy()
        emu = {'y': _rel(yq.detach().numpy(), npz[f'{name}_y']), 'grad_net': _rel(gq[:n_net], npz[f'{name}_grad_net']),
               'grad_grid': _rel(gq[n_net:], _dense_grid_grad(npz, f'{name}_grad_grid', p.size - n_net))}
        tight = {'y': _rel(y.detach().float().cpu().numpy(), yq.detach().numpy()), 'grad_net': _rel(g[:n_net], gq[:n_net]),
                 'grad_grid': _rel(g[n_net:], gq[n_net:])}
        rep[name + '_emulation_vs_vectors'] = emu
        rep[name + '_hip_vs_emulation'] = tight
        for k in rep[name]:
            assert rep[name][k] <= 2.0 * emu[k] + 2e-3, (name, dtype, k, rep[name], emu)
            assert tight[k] <= {'fp16': 5e-3, 'bf16': 2e-2}[dtype], (name, dtype, k, tight)
    e = tcnn.Encoding(3, dict(PU.ENC_SMOOTH), dtype='fp32')
    p, _ = _params(npz, 'enc')
    with torch.no_grad():
        e.params.copy_(torch.from_numpy(p).cuda())
    x = torch.from_numpy(npz['enc_x']).cuda().requires_grad_(True)
    f = e(x).float()
    gx, = torch.autograd.grad((f * torch.from_numpy(npz['enc_w']).cuda()).sum(), x, create_graph=True)
    rep['enc'] = {'y': _rel(f.detach().cpu().numpy(), npz['enc_y']), 'dx': _rel(gx.detach().cpu().numpy(), npz['enc_dx'])}
    if 'enc_dd_x' in npz.files:
        (gx ** 2).sum().backward()
        rep['enc']['dd_x'] = _rel(x.grad.cpu().numpy(), npz['enc_dd_x'])
        rep['enc']['dd_grid'] = _rel(e.params.grad.cpu().numpy(), _dense_grid_grad(npz, 'enc_dd_grid', p.size))
    assert max(rep['enc'].values()) <= (2e-2 if up else 2e-5), rep['enc']
    return rep


def hip_vs_nerfacc(npz):
    from perf_amd import nerfacc_impl as N
    R = npz['o'].shape[0]
    res = int(npz['res'])
    est = N.OccGridEstimator(roi_aabb=torch.tensor(PU.AABB), resolution=res, levels=1).cuda()
    est.set_binaries(torch.from_numpy(_binaries(npz).reshape(-1)).cuda())
    est.eval()
    o, d = torch.from_numpy(npz['o']).cuda(), torch.from_numpy(npz['d']).cuda()
    for tag in ('coarse', 'perf'):
        ri, ts, te = est.sampling(o, d, sigma_fn=None, near_plane=0.0, far_plane=1.5, render_step_size=float(npz[f'{tag}_step']),
                                  stratified=False, cone_angle=0.0, alpha_thre=0.0)
        assert np.array_equal(ri.cpu().numpy(), npz[f'{tag}_ray_indices']), tag            # bit-exact bookkeeping (north_star)
        assert np.array_equal(ts.cpu().numpy(), npz[f'{tag}_t_starts']) and np.array_equal(te.cpu().numpy(), npz[f'{tag}_t_ends']), tag

    def sigma_fn(t_starts, t_ends, ray_indices):
        x = o[ray_indices] + d[ray_indices] * ((t_starts + t_ends)[:, None] / 2.0)
        return PU._sigma_analytic(x)
    ri, ts, te = est.sampling(o, d, sigma_fn=sigma_fn, near_plane=0.0, far_plane=1.5, render_step_size=4e-3, stratified=False,
                              cone_angle=0.0, alpha_thre=0.0)
    assert np.array_equal(ri.cpu().numpy(), npz['vis_ray_indices']) and np.array_equal(ts.cpu().numpy(), npz['vis_t_starts'])
    sig = torch.from_numpy(npz['vis_sigmas']).cuda()
    w, T, al = N.render_weight_from_density(ts, te, sig, ray_indices=ri, n_rays=R)
    acc = N.accumulate_along_rays(w, torch.from_numpy(npz['values']).cuda(), ray_indices=ri, n_rays=R)
    opa = N.accumulate_along_rays(w, None, ray_indices=ri, n_rays=R)
    rep = {'weights': _rel(w.cpu().numpy(), npz['weights']), 'trans': _rel(T.cpu().numpy(), npz['trans']), 'alphas': _rel(al.cpu().numpy(), npz['alphas']),
           'accumulated': _rel(acc.cpu().numpy(), npz['accumulated']), 'opacity': _rel(opa.cpu().numpy(), npz['opacity'])}
    assert max(rep.values()) <= 5e-6, rep
    return rep


def hip_vs_distloss(npz):
    from perf_amd.distloss import flatten_eff_distloss
    w = torch.from_numpy(npz['w']).cuda().requires_grad_(True)
    loss = flatten_eff_distloss(w, torch.from_numpy(npz['m']).cuda(), torch.from_numpy(npz['interval']).cuda(), torch.from_numpy(npz['ray_id']).cuda())
    loss.backward()
    rep = {'loss': abs(float(loss) - float(npz['loss'])) / abs(float(npz['loss'])), 'grad_w': _rel(w.grad.cpu().numpy(), npz['grad_w'])}
    assert max(rep.values()) <= 2e-5, rep
    return rep
