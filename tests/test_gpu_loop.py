"""PeRF's progressive multi-panorama loop in miniature (tools/mini_perf_loop.py): episode -> new position -> rendered
distance -> get_pano_visibility_mask -> geo_check -> register_sup_info (PanoSupInfo's rules) -> occupancy rebuild -> next
episode, on the synthetic room with an occluding box.  Pins the flow end to end on the GPU: supervision pools of several
origins, rays that do not start at the centre, the reprojection / morphology kernels on rendered distances."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_room_with_box_reduces_to_the_room():
    """Without the box and from the centre, room_with_box is room() up to the normalisation constant (room() normalises by
    the largest distance of the sampled directions, room_with_box by the corner distance)."""
    from perf_amd import synthetic
    from perf_amd.scene import gen_pano_rays
    rays = gen_pano_rays(torch.eye(4), 64, 128)
    d0, c0 = synthetic.room(rays.d)
    d1, c1 = synthetic.room_with_box(rays.o, rays.d, box_h=(0., 0., 0.))
    ratio = d1 / d0
    assert float(ratio.max() - ratio.min()) < 1e-5 and 0.98 < float(ratio.mean()) < 1.0
    assert float((c0 - c1).abs().max()) < 1e-5
    d2, _ = synthetic.room_with_box(rays.o, rays.d)
    covered = float((d2 < d1 - 1e-6).float().mean())
    assert 0.01 < covered < 0.2                                  # the box hides part of the walls from the centre


def test_progressive_loop_runs_end_to_end():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'mini_perf_loop.py'), '--views', '3', '--geo', '150', '--app', '100', '--height', '128'],
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"config"')][-1])
    eps = out['episodes']
    assert [e['panoramas'] for e in eps] == [1, 2, 3]
    assert all(e['skipped_steps'] == 0 for e in eps)
    assert eps[0]['supervision_rays'] <= eps[1]['supervision_rays'] <= eps[2]['supervision_rays'] <= 3 * 128 * 256
    for e in eps[:2]:
        assert 0.5 < e['new_view_seen_fraction'] <= 1.0 and 0.0 <= e['geo_check_ok_fraction'] <= 1.0
        assert e['new_rays_registered'] >= 0
    assert all(e['first_pano']['psnr_dB'] > 20.0 and e['held_out']['psnr_dB'] > 15.0 for e in eps)
    assert all(e['first_pano']['mean_abs_distance_err'] < 0.06 for e in eps)     # (32 k supervision points leave holes in the 256^3 occupancy shell at this size)
