#!/usr/bin/env python
"""Forward-encode A/B: the statically dealt kernel against the kernel with rotating level groups and run de-duplication (hashgrid.hip), on three
sample distributions -- (train) 8192 rays x 128 lattice samples in ray order, as bench.py's step encodes them; (eval) the
2-row heads of neighbouring pixels of a 512x1024 frame, as the two-phase sampler of a trained scene encodes them; (random)
uniform points.  Every variant runs in its own process (the switches are read once); features must be bit-identical.

    python tools/exp/fwd_v2.py [--out gpurun_out/fwd_v2.json]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

VARIANTS = {
    'static (round 2 kernel)': {'PERF_FWD_V2': '0'},
    'rotate + dedup': {},
    'rotate only': {'PERF_FWD_NO_DEDUP': '1'},
    'dedup only (fixed pinning)': {'PERF_FWD_NO_ROTATE': '1'},
    'neither (new kernel body)': {'PERF_FWD_NO_ROTATE': '1', 'PERF_FWD_NO_DEDUP': '1'},
    'rotate + dedup, 512 looping workgroups per XCD': {'PERF_FWD_MAX_CHUNKS': '512'},
}


def samples(kind, dev):
    import math
    import torch
    g = torch.Generator().manual_seed(3)
    if kind == 'train':
        R, S = 8192, 128
        d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
        t = (torch.arange(S)[None, :] + torch.rand(R, 1, generator=g)) * (0.99 / S)
        x = 0.5 + 0.5 * d[:, None, :] * t[:, :, None]
        return x.reshape(-1, 3).contiguous().to(dev)
    if kind == 'eval':
        H, W, K = 512, 1024, 2
        i = (torch.arange(H) + 0.5) / H; j = (torch.arange(W) + 0.5) / W
        beta = -(i - 0.5) * math.pi; alpha = -(j - 0.5) * 2 * math.pi
        d = torch.stack([torch.cos(alpha)[None, :] * torch.cos(beta)[:, None], torch.sin(alpha)[None, :] * torch.cos(beta)[:, None],
                         torch.sin(beta)[:, None].expand(H, W)], -1).reshape(-1, 3)
        # box room of half extents (0.857, 0.667, 0.476): distance to the wall along d
        ext = torch.tensor([0.857, 0.667, 0.476])
        dist = (ext / d.abs().clamp_min(1e-9)).min(-1).values
        t = dist[:, None] - 0.004 + torch.arange(K)[None, :] * 5e-4
        x = 0.5 + 0.5 * d[:, None, :] * t[:, :, None]
        return x.reshape(-1, 3).contiguous().to(dev)
    import torch
    return torch.rand(1 << 20, 3, generator=g).to(dev)


def worker(out_path):
    import torch
    from perf_amd import ops
    from perf_amd.grid import GridConfig
    dev = torch.device('cuda', 0)
    cfg = GridConfig()
    g = torch.Generator().manual_seed(1)
    res = {}
    for dt in ('bf16',):
        table = ((torch.rand(cfg.n_params, generator=g) * 2 - 1) * 0.5).to(dev)
        t16 = ops.cast_params(table, dt)
        for kind in ('train', 'eval', 'random'):
            x = samples(kind, dev)
            feat = ops.hashgrid_fwd(cfg, x, t16)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 30
            for _ in range(5):
                ops.hashgrid_fwd(cfg, x, t16)
            a.record()
            for _ in range(reps):
                ops.hashgrid_fwd(cfg, x, t16)
            b.record(); torch.cuda.synchronize()
            res[kind] = {'n': x.shape[0], 'ms': a.elapsed_time(b) / reps,
                         'sha': hashlib.sha256(feat.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]}
    json.dump(res, open(out_path, 'w'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'fwd_v2.json'))
    ap.add_argument('--worker', default=None)
    a = ap.parse_args()
    if a.worker:
        return worker(a.worker)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    table = {}
    for name, env in VARIANTS.items():
        tmp = a.out + '.tmp'
        e = dict(os.environ, **env)
        for k in ('PERF_FWD_V2', 'PERF_FWD_NO_DEDUP', 'PERF_FWD_NO_ROTATE', 'PERF_FWD_MAX_CHUNKS'):
            if k not in env:
                e.pop(k, None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', tmp], env=e, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            table[name] = {'error': r.stderr[-1500:]}
            continue
        table[name] = json.load(open(tmp)); os.remove(tmp)
    ref = table.get('static (round 2 kernel)', {})
    for name, res in table.items():
        if 'error' in res:
            continue
        for kind, row in res.items():
            row['bit_identical_to_static'] = (row['sha'] == ref.get(kind, {}).get('sha'))
            row['Gsamples_per_s'] = round(row['n'] / row['ms'] / 1e6, 3)
            row['frac_of_hbm_peak_algorithmic'] = round(512 * row['n'] / (row['ms'] * 1e-3) / 8e12, 4)
    json.dump(table, open(a.out, 'w'), indent=1)
    for name, res in table.items():
        print(name, {k: (round(v['ms'], 4), v['bit_identical_to_static']) for k, v in res.items()} if 'error' not in res else res)


if __name__ == '__main__':
    main()
