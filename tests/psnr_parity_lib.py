"""PSNR@iter parity harness (test infrastructure): the HIP path against the fp32 CPU oracle after EQUAL iterations on the
same synthetic panorama with identical batches and random draws (north_star: "PSNR within 0.1 dB of reference after equal
iterations"; the reference itself cannot run -- tinycudann / nerfacc are absent -- so the oracle stands in).

Used by tests/golden/make_psnr_curve.py (runs the oracle in the build container, commits its curve as a fixture),
tests/test_gpu_psnr.py (runs the HIP path on the GPU box against that fixture) and tools/psnr_parity.py."""
import hashlib

import numpy as np
import torch

from oracle import perf_oracle as O

AABB = [-1., -1, -1, 1, 1, 1]
GEO_MARKS = (30, 100, 300)      # geometry iterations at which the training depth loss is compared (window of 10)
CONF = dict(init_lr=0.0, peak_lr=1e-2, peak_at=0.2, lr_alpha=1e-2)      # configs/nerf.yaml:36-47


def psnr(a, b):
    return float(-10 * torch.log10(torch.mean((a.float() - b.float()) ** 2)))


def make_scene(h, w, scene='room'):
    """scene: 'room' (the oracle's own synthetic room) or one of the other families of perf_amd.synthetic ('doorway', 'pillars':
    pure torch functions of the ray directions, evaluated here on the CPU)."""
    o, d = O.pano_rays(torch.eye(4), h, w)
    o = o.reshape(-1, 3).contiguous(); d = d.reshape(-1, 3).contiguous()
    if scene == 'room':
        dist, rgb = O.synthetic_room(d)
    else:
        from perf_amd import synthetic
        dist, rgb = synthetic.SCENES[scene](d)
    occ = O.gen_occ_grid(o, d, dist, 256).reshape(256, 256, 256).bool().numpy()
    return o, d, dist, rgb, occ


def make_draws(n_pool, batch, n_steps, seed=0):
    """Batch indices and the three per-ray random draws of every step (torch CPU generator: identical on every box)."""
    g = torch.Generator().manual_seed(123 + seed)
    draws = []
    for _ in range(n_steps):
        draws.append({'idx': torch.randint(0, n_pool, (batch,), generator=g), 'jitter': torch.rand(batch, generator=g),
                      'bg': torch.rand(batch, 3, generator=g), 'noise': torch.rand(batch, 1, generator=g)})
    return draws


def draws_digest(draws):
    h = hashlib.sha256()
    for dr in draws[:4] + draws[-4:]:
        for k in ('idx', 'jitter', 'bg', 'noise'):
            h.update(dr[k].numpy().tobytes())
    return h.hexdigest()[:16]


def init_params(seed):
    return O.init_field_params(O.geo_spec(), 1337 + seed), O.init_field_params(O.app_spec(), 1337 + seed)


def run_oracle(scene, geo0, app0, draws, n_geo, n_app, marks, log=None, quant=None):
    """-> {'geo_end_depth_err', 'psnr@app<k>' for k in marks} and the geometry phase's learning curve: 'geo_depth_loss@<k>' (mean
    training depth loss of iterations k-10..k-1, k in GEO_MARKS: falls by two orders of magnitude while the field grows
    opaque -- unlike the eval depth error, which the occupancy shell fixes from the first iteration) and
    'geo_end_opacity' (mean eval opacity after the phase).
    quant ('bf16' | 'fp16' | None): the oracle's 16-bit emulation -- the forward passes see parameters and encoded features rounded to
    that type (what tcnn's and this build's 16-bit working copies do), everything else (master weights, Adam, compositing,
    losses) stays fp32: what the STORAGE TYPE ALONE does to the curve."""
    o, d, dist, rgb, occ = scene
    geo = geo0.clone().requires_grad_(True); app = app0.clone().requires_grad_(True)
    curve = {}
    mg = torch.zeros_like(geo); vg = torch.zeros_like(geo); ma = torch.zeros_like(app); va = torch.zeros_like(app)

    def render_eval():
        outs_rgb, outs_d, outs_o = [], [], []
        with torch.no_grad():
            for lo in range(0, o.shape[0], 16384):
                out = O.occ_render(o[lo:lo + 16384], d[lo:lo + 16384], geo, app, occ, AABB, training=False, quant=quant)
                outs_rgb.append(out['rgb']); outs_d.append(out['distance']); outs_o.append(out['opacities'])
        return torch.cat(outs_rgb), torch.cat(outs_d), torch.cat(outs_o)

    step_g = 0
    dls = []
    for i in range(n_geo):
        dr = draws[i]
        t0 = (dr['jitter'].numpy() * np.float32(5e-4)).astype(np.float32)
        out = O.occ_render(o[dr['idx']], d[dr['idx']], geo, app, occ, AABB, training=True, t0=t0, bg_color=dr['bg'], dist_noise=dr['noise'], quant=quant)
        if not out['is_valid']:
            continue
        loss, dl_i, _ = O.geo_step_loss(out, dist[dr['idx']], progress=i / n_app)
        dls.append(float(dl_i))
        geo.grad = None; loss.backward()
        step_g += 1
        with torch.no_grad():
            p, mg, vg = O.adam_step(geo, geo.grad, mg, vg, step_g, O.lr_schedule(i / n_geo, **CONF)); geo.copy_(p)
        if log and (i + 1) % 50 == 0:
            log(f'geo {i + 1}/{n_geo}')
    ev = render_eval()
    curve['geo_end_depth_err'] = float((ev[1] - dist).abs().mean())
    curve['geo_end_opacity'] = float(ev[2].mean())
    for k in GEO_MARKS:
        if k <= len(dls):
            curve[f'geo_depth_loss@{k}'] = float(np.mean(dls[k - 10:k]))
    step_a = 0
    for i in range(n_app):
        dr = draws[n_geo + i]
        t0 = (dr['jitter'].numpy() * np.float32(5e-4)).astype(np.float32)
        out = O.occ_render(o[dr['idx']], d[dr['idx']], geo, app, occ, AABB, training=True, t0=t0, bg_color=dr['bg'], dist_noise=dr['noise'],
                           geo_grad=False, app_grad=True, quant=quant)
        if out['is_valid']:
            loss, _ = O.app_step_loss(out, rgb[dr['idx']])
            app.grad = None; loss.backward()
            step_a += 1
            with torch.no_grad():
                p, ma, va = O.adam_step(app, app.grad, ma, va, step_a, O.lr_schedule(i / n_app, **CONF)); app.copy_(p)
        if (i + 1) in marks:
            curve[f'psnr@app{i + 1}'] = psnr(render_eval()[0], rgb)
            if log:
                log(f'app {i + 1}/{n_app}: {curve[f"psnr@app{i + 1}"]:.3f} dB')
    return curve


def run_hip(scene, geo0, app0, draws, n_geo, n_app, marks, dtype, accum='fixed', batch=None):
    """The same schedule on the HIP path (eager steps with the injected draws)."""
    from perf_amd import tcnn
    from perf_amd.scene import NeRFScene, Rays, SupInfoPool
    o, d, dist, rgb, occ = scene
    batch = batch or draws[0]['idx'].numel()
    tcnn.GRID_GRAD_ACCUM = accum
    sc = NeRFScene(dtype=dtype)
    pool = SupInfoPool(); pool.register_rays(o.cuda(), d.cuda(), rgb.cuda(), dist.cuda())
    sc.train_conf.pixel_loss_batch_size = batch
    sc.set_train()
    sc.estimator.set_binaries(torch.from_numpy(occ.reshape(-1)).cuda())
    sc.nerf.reset_geo()
    with torch.no_grad():
        sc.nerf.geo_mlp.params.copy_(geo0.cuda()); sc.nerf.app_mlp.params.copy_(app0.cuda())
    state = {'idx': None}
    pool.rand_ray_color_data = lambda bs, **kw: (Rays(pool.all_sup_rays.o[state['idx']], pool.all_sup_rays.d[state['idx']]),
                                                  pool.all_sup_colors[state['idx']], pool.all_sup_distances[state['idx']],
                                                  pool.all_sup_normals[state['idx']])
    rays = Rays(o.cuda(), d.cuda())
    curve = {}
    opt = sc.make_optimizer(sc.nerf.geo_mlp, 0.0)
    conf = sc.train_conf.geo_optimizer
    dls = []
    for i in range(n_geo):
        dr = draws[i]; state['idx'] = dr['idx'].cuda()
        sc.update_lr(opt, conf, i / n_geo)
        sc.train_one_step_geo(opt, pool, progress=i / n_app, rand={k: dr[k].cuda() for k in ('jitter', 'bg', 'noise')})
        dls.append(sc.last_losses['depth_loss'])
    dls = [float(v) for v in torch.stack(dls).cpu()]
    ev = sc.render(rays, ['rgb', 'distance', 'opacities'])
    curve['geo_end_depth_err'] = float((ev['distance'].cpu() - dist).abs().mean())
    curve['geo_end_opacity'] = float(ev['opacities'].mean())
    for k in GEO_MARKS:
        if k <= len(dls):
            curve[f'geo_depth_loss@{k}'] = float(np.mean(dls[k - 10:k]))
    sc.set_train()
    opt = sc.make_optimizer(sc.nerf.app_mlp, 0.0)
    for i in range(n_app):
        dr = draws[n_geo + i]; state['idx'] = dr['idx'].cuda()
        sc.update_lr(opt, conf, i / n_app)
        sc.train_one_step_app(opt, pool, progress=i / n_app, rand={k: dr[k].cuda() for k in ('jitter', 'bg', 'noise')})
        if (i + 1) in marks:
            curve[f'psnr@app{i + 1}'] = psnr(sc.render(rays, ['rgb'])['rgb'].cpu(), rgb); sc.set_train()
    return curve
