"""Per-kernel timings on the GPU box (HIP events on torch's current stream)."""
import sys, os, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perf_amd import ops
from perf_amd.grid import GridConfig, MlpConfig


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = 'cuda'
    cfg = GridConfig()
    res = {}
    for dt in ('bf16', 'fp16'):
        tdt = ops.torch_dtype(dt)
        table = (torch.rand(cfg.n_params, device=dev) * 2 - 1).to(tdt)
        geo = MlpConfig(16, 1, 1, 'Exponential'); app = MlpConfig(16, 2, 3, 'Sigmoid')
        wg = (torch.randn(geo.n_params, device=dev) * 0.2).to(tdt)
        wa = (torch.randn(app.n_params, device=dev) * 0.2).to(tdt)
        for n in (1 << 20, 1 << 22):
            # ray-coherent positions: 128 samples along each ray from the origin
            R = n // 128
            d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
            t = (torch.arange(128, device=dev) + 0.5) / 128
            xr = (d[:, None, :] * t[None, :, None]).reshape(-1, 3) * 0.5 + 0.5
            xu = torch.rand(n, 3, device=dev)
            for name, x in (('ray', xr), ('uniform', xu)):
                x = x.contiguous()
                tf = timeit(lambda: ops.hashgrid_fwd(cfg, x, table))
                feat = ops.hashgrid_fwd(cfg, x, table)
                dfeat = torch.randn(16, n, 2, device=dev)
                grad = torch.zeros(cfg.n_params, device=dev)
                tb = timeit(lambda: ops.hashgrid_bwd(cfg, x, dfeat, grad))
                res[f'{dt}/n{n}/{name}/hashgrid_fwd_Msps'] = n / tf / 1e6
                res[f'{dt}/n{n}/{name}/hashgrid_bwd_Msps'] = n / tb / 1e6
                amax = torch.zeros(16, device=dev); amax[:] = dfeat.abs().amax(dim=(1, 2))
                tbf = timeit(lambda: ops.hashgrid_bwd(cfg, x, dfeat, grad, level_absmax=amax))
                res[f'{dt}/n{n}/{name}/hashgrid_bwd_fixed_Msps'] = n / tbf / 1e6
            tg = timeit(lambda: ops.mlp_fwd(geo, wg, feat))
            ta = timeit(lambda: ops.mlp_fwd(app, wa, feat))
            dg = torch.randn(n, 1, device=dev); da = torch.randn(n, 3, device=dev)
            tgb = timeit(lambda: ops.mlp_bwd(geo, wg, feat, dg))
            tab = timeit(lambda: ops.mlp_bwd(app, wa, feat, da))
            res[f'{dt}/n{n}/mlp_fwd_geo_Msps'] = n / tg / 1e6
            res[f'{dt}/n{n}/mlp_fwd_app_Msps'] = n / ta / 1e6
            res[f'{dt}/n{n}/mlp_bwd_geo_Msps'] = n / tgb / 1e6
            res[f'{dt}/n{n}/mlp_bwd_app_Msps'] = n / tab / 1e6
    p32 = torch.randn(6644288, device=dev)
    res['cast_us'] = timeit(lambda: ops.cast_params(p32, 'bf16')) * 1e6
    m = torch.zeros_like(p32); v = torch.zeros_like(p32); g = torch.randn_like(p32); w16 = torch.empty_like(p32, dtype=torch.bfloat16)
    res['adam_us'] = timeit(lambda: ops.adam_step(p32, m, v, g, 1, 1e-3, w16=w16)) * 1e6
    for k, v_ in res.items():
        print(f'{k:48s} {v_:12.1f}')
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/microbench.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
