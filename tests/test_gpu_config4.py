"""BASELINE config 4: the render_dense traverse (core_exp_runner.py:223-246) as hipGraph-captured fp16 eval frames --
512x1024 rays per frame in 16 batches of 32,768 (the reference's hard-coded eval batch, nerf.py:86), rays generated from a
device-resident pose, variable sample counts decided on the device."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def trained_scene():
    from perf_amd import synthetic
    from perf_amd.scene import NeRFScene, SupInfoPool, gen_pano_rays
    torch.manual_seed(0); np.random.seed(0)
    scene = NeRFScene(dtype='fp16')
    rays = gen_pano_rays(torch.eye(4), 256, 512)
    dist, rgb = synthetic.room(rays.d)
    pool = SupInfoPool(); pool.register_rays(rays.o, rays.d, rgb, dist)
    scene.train_conf.pixel_loss_batch_size = 4096
    scene.train_one_episode(pool, 150, 100)                     # (hipGraph-replayed steps after the eager head)
    return scene, pool, dist, rgb


def _pose(tx, ty, tz):
    p = torch.eye(4)
    p[0, 3], p[1, 3], p[2, 3] = tx, ty, tz
    return p


def test_graphed_frame_equals_eager_render(trained_scene):
    from perf_amd.scene import gen_pano_rays
    scene, pool, dist, rgb = trained_scene
    H, W = 512, 1024
    frame = scene.make_graphed_render(H, W, ('rgb', 'distance', 'opacities'), batch_size=32768)
    for pose in (_pose(0, 0, 0), _pose(0.12, -0.07, 0.03), _pose(-0.2, 0.15, -0.05)):
        got = {k: v.clone() for k, v in frame(pose).items()}
        rays = gen_pano_rays(pose, H, W)
        ref = scene.render(rays, ['rgb', 'distance', 'opacities'], batch_size=32768, sync_free=False)   # the synced reference path
        for k in ref:
            assert got[k].shape == ref[k].shape == (H, W, ref[k].shape[-1])
            assert torch.equal(got[k], ref[k]), k
        # size-independent properties of a frame
        assert torch.isfinite(got['rgb']).all() and torch.isfinite(got['distance']).all()
        assert float(got['opacities'].min()) >= 0.0 and float(got['opacities'].max()) <= 1.0 + 1e-4
        assert float(got['rgb'].min()) >= 0.0 and float(got['rgb'].max()) <= 1.0 + 1e-3
        # a replay is deterministic
        again = frame(pose)
        assert torch.equal(again['rgb'], got['rgb'])
    # the trained room is seen from the origin roughly as it was supervised
    got = frame(_pose(0, 0, 0))
    small = torch.nn.functional.interpolate(got['distance'].permute(2, 0, 1)[None], size=(256, 512), mode='area')[0, 0]
    assert float((small - dist[..., 0]).abs().mean()) < 0.05


def test_graphed_frame_recaptures_when_the_capacity_is_too_small(trained_scene):
    from perf_amd.scene import gen_pano_rays
    scene, pool, dist, rgb = trained_scene
    H, W = 128, 256
    pose = _pose(0.05, 0.02, 0.0)
    ref = scene.render(gen_pano_rays(pose, H, W), ['rgb', 'distance'], batch_size=8192, sync_free=False)
    frame = scene.make_graphed_render(H, W, ('rgb', 'distance'), batch_size=8192, samples_per_ray=1)
    got = frame(pose)
    assert frame.state['per_ray'] > 1
    for k in ref:
        assert torch.equal(got[k], ref[k]), k


def test_render_dense_traverse_runs_on_graphed_frames(trained_scene):
    """perf_amd.traverse.render_dense: pose samplers on the host, every frame one graph replay."""
    from perf_amd.pose_sampler import CirclePoseSampler
    from perf_amd.traverse import render_dense
    scene, pool, dist, rgb = trained_scene
    sparse = CirclePoseSampler(dist.reshape(256, 512).cpu(), traverse_ratios=[.2, .4, .6], n_anchors_per_ratio=[8, 8, 8])
    sums = []
    render_dense(scene, sparse, n_poses=24, height=128, width=256, on_frame=lambda i, pose, res: sums.append(float(res['rgb'].sum())),
                 max_frames=6)
    assert len(sums) == 6 and all(np.isfinite(sums)) and len(set(sums)) > 1          # the camera moves
