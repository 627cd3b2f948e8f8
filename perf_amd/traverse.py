"""Dense fly-through rendering (core_exp_runner.py:223-246 `CoreRunner.render_dense`, BASELINE config 4) without the
image/video IO: a DenseTravelPoseSampler trajectory through the anchor poses, one 512x1024 panorama per pose
(rotation reset to identity as the reference does for cam_type='pano'), rays generated in-kernel."""
import torch

from .pose_sampler import DenseTravelPoseSampler
from .scene import Rays, gen_pano_rays


@torch.no_grad()
def render_dense(scene, pose_sampler, n_poses=180, height=512, width=1024, query_keys=('rgb', 'distance'),
                 on_frame=None, max_frames=None):
    """Returns the list of per-frame result dicts (or calls on_frame(i, pose, result) and keeps nothing)."""
    dense = DenseTravelPoseSampler(pose_sampler, n_dense_poses=n_poses)
    frames = []
    n = dense.n_poses if max_frames is None else min(dense.n_poses, max_frames)
    for i in range(n):
        pose = dense.sample_pose(i).clone()
        pose[:3, :3] = torch.eye(3)
        rays = gen_pano_rays(pose, height, width)
        res = scene.render(Rays(rays.o, rays.d), query_keys=list(query_keys))
        if on_frame is not None:
            on_frame(i, pose, res)
        else:
            frames.append(res)
    return frames
