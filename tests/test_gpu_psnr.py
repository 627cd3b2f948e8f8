"""PSNR@iter parity (north_star: "PSNR within 0.1 dB of reference after equal iterations"): the HIP path replays the
schedule of tests/golden/psnr_curve*.json -- the fp32 CPU oracle's curves, generated in the build container by
tests/golden/make_psnr_curve.py (256x512 panorama, 1024-ray batches, 300 geometry + 300 colour iterations, identical
batches and random draws).  The bounds -- the mean over seeds within 0.1 dB; every seed, as the mean of its one-ulp ensemble, within
max(0.1 dB, the oracle's own one-ulp spread) + what the storage type alone does to the oracle -- are stated and computed in
tests/psnr_bounds.py; tests/test_psnr_bounds.py holds the committed evidence (profiles/r06_psnr_ensemble.json) to the same checker.
profiles/r06_psnr_split.json has the HIP deviations split over {bf16, fp16} x {fixed-point, fp32 accumulation}: no combination is
systematically off, none is the cause of a single seed's deviation."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('scene_name', ['room', 'doorway', 'pillars'])
def test_psnr_at_iter_matches_the_oracle_curve(scene_name):
    """Three scene families (perf_amd/synthetic.py): the box room every constant of the fixed-point machinery was tuned on, two rooms
    joined by a doorway (thin wall, 3x depth discontinuities, a long free space behind an occluder) and a room with fourteen thin
    pillars (high-frequency occupancy, many short free spans) -- each against ITS oracle curve (tests/golden/psnr_curve[_<scene>].json).
    Per seed and 16-bit type THREE runs: the golden initialisation and that initialisation moved by one fp32 ulp up / down (the
    perturbation `oracle_spread` applies to the oracle): a 16-bit path moves by up to 0.2 dB under it (profiles/r06_psnr_ensemble.json),
    so the seed's statistic is the MEAN of the three; the members and their spread are printed."""
    from perf_amd import tcnn
    from tests import psnr_parity_lib as P
    name = 'psnr_curve.json' if scene_name == 'room' else f'psnr_curve_{scene_name}.json'
    golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', name)))
    cfg = golden['config']
    assert cfg.get('scene', 'room') == scene_name
    h, w = cfg['pano']
    scene = P.make_scene(h, w, scene_name)
    marks = [f'psnr@app{m}' for m in cfg['marks']]
    geo_marks = cfg.get('geo_marks', [])
    mode0 = tcnn.GRID_GRAD_ACCUM
    from tests import psnr_bounds as B
    other = 'fp16' if tcnn.DEFAULT_DTYPE == 'bf16' else 'bf16'          # fp16 = tcnn's own storage type when the default is bf16
    try:
        for dtype in (tcnn.DEFAULT_DTYPE, other):
            by_seed = {}
            depth, opacity = [], []
            geo_ratio = {k: [] for k in geo_marks}
            for row in golden['seeds']:
                sd = row['seed']
                geo0, app0 = P.init_params(sd)
                draws = P.make_draws(scene[0].shape[0], cfg['batch'], cfg['geo_iters'] + cfg['app_iters'], sd)
                assert P.draws_digest(draws) == row['draws_digest'], 'the CPU generator must reproduce the fixture draws'
                members = []
                for towards in (None, float('inf'), -float('inf')):
                    g, a = geo0, app0
                    if towards is not None:
                        g = torch.nextafter(geo0, torch.full_like(geo0, towards)); a = torch.nextafter(app0, torch.full_like(app0, towards))
                    members.append(P.run_hip(scene, g, a, draws, cfg['geo_iters'], cfg['app_iters'], tuple(cfg['marks']), dtype, mode0))
                got = members[0]
                by_seed[sd] = members
                depth.append(got['geo_end_depth_err'] / row['oracle']['geo_end_depth_err'])
                for k in geo_marks:
                    geo_ratio[k].append(got[f'geo_depth_loss@{k}'] / row['oracle'][f'geo_depth_loss@{k}'])
                if 'geo_end_opacity' in row['oracle']:
                    opacity.append(got['geo_end_opacity'] - row['oracle']['geo_end_opacity'])
            B.check_family(golden, dtype, by_seed)
            print('depth-error ratio:', [round(v, 3) for v in depth])
            assert 0.85 <= float(np.mean(depth)) <= 1.15, depth
            # the geometry phase's LEARNING curve (the eval depth error above is fixed by the occupancy shell from the first
            # iteration and carries no information about learning): mean training depth loss of iterations k-10..k-1 -- 0.41 at
            # k = 30, 0.005 at k = 100 for the oracle -- must follow the oracle's on every seed, and the field must end as opaque
            print('HIP / oracle training depth loss:', {k: [round(v, 5) for v in vs] for k, vs in geo_ratio.items()}, 'opacity delta:', [round(v, 5) for v in opacity])
            for k, vs in geo_ratio.items():
                assert 0.93 <= float(np.mean(vs)) <= 1.07 and all(0.85 <= v <= 1.15 for v in vs), (dtype, k, vs)
            assert all(abs(v) < 5e-3 for v in opacity), opacity
    finally:
        tcnn.GRID_GRAD_ACCUM = mode0


def test_data_parallel_psnr_at_iter_matches_the_oracle_curve(tmp_path):
    """`north_star` asks for scaling AND "PSNR within 0.1 dB after equal iterations".  The data-parallel DEFAULT (sharded
    exchange with lagged fixed-point units, perf_amd/dp.py) is not bit-identical to the single process, so it replays the
    oracle's schedule itself: two ranks on this box's one GPU (gloo), each taking its half of every golden batch and of the
    batch's random draws -- three seeds of tests/golden/psnr_curve.json (a step costs ~50 ms through gloo's host copies): the
    mean of (data-parallel HIP - fp32 oracle) within 0.1 dB at both marks, single seeds within the single-process bound (tests/psnr_bounds.py), the geometry phase's
    learning curve on the oracle's, no step skipped by the job-wide gate."""
    import subprocess
    import sys
    golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'psnr_curve.json')))
    cfg = golden['config']
    seeds = [row['seed'] for row in golden['seeds']][:3]
    out = str(tmp_path / 'dp_psnr.json')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT)
    env.pop('PERF_DP_UNITS', None)
    import socket
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'psnr_dp_worker.py'), out] + [str(s) for s in seeds]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert res['world'] == 2 and sorted(res['curves']) == sorted(str(s) for s in seeds)
    rows = {str(row['seed']): row['oracle'] for row in golden['seeds']}
    deltas = {f'psnr@app{m}': [res['curves'][str(s)][f'psnr@app{m}'] - rows[str(s)][f'psnr@app{m}'] for s in seeds] for m in cfg['marks']}
    print('data-parallel (2 ranks, lagged units) HIP - oracle PSNR [dB]:', {k: [round(v, 3) for v in vs] for k, vs in deltas.items()})
    from perf_amd import tcnn as _tc
    from tests import psnr_bounds as B
    B.check_family(golden, _tc.DEFAULT_DTYPE, {s: [res['curves'][str(s)]] + list(res['curves'][str(s)]['one_ulp_members']) for s in seeds})
    for k in cfg.get('geo_marks', []):
        ratio = [res['curves'][str(s)][f'geo_depth_loss@{k}'] / rows[str(s)][f'geo_depth_loss@{k}'] for s in seeds]
        assert 0.93 <= float(np.mean(ratio)) <= 1.07 and all(0.85 <= v <= 1.15 for v in ratio), (k, ratio)
    for s in seeds:
        c = res['curves'][str(s)]
        assert c['skipped_for_overflow'] == 0 and c['skipped_for_truncation'] == 0, (s, c['skipped_for_overflow'], c['steps_skipped_for_overflow'])
        assert all(m['skipped_for_overflow'] == 0 and m['skipped_for_truncation'] == 0 for m in c['one_ulp_members']), (s, c['one_ulp_members'])
        assert abs(c['geo_end_opacity'] - rows[str(s)]['geo_end_opacity']) < 5e-3
