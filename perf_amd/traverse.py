"""Dense fly-through rendering (core_exp_runner.py:223-246 `CoreRunner.render_dense`, BASELINE config 4) without the
image/video IO: a DenseTravelPoseSampler trajectory through the anchor poses, one 512x1024 panorama per pose
(rotation reset to identity as the reference does for cam_type='pano').  Every frame is ONE replay of a hipGraph that
holds ray generation from a device-resident pose plus the 32,768-ray eval batches of NeRFScene.render
(NeRFScene.make_graphed_render): no per-batch host work, sample counts stay on the device."""
import torch

from .pose_sampler import DensePoseFuture, DenseTravelPoseSampler


@torch.no_grad()
def render_dense(scene, pose_sampler, n_poses=180, height=512, width=1024, query_keys=('rgb', 'distance'),
                 on_frame=None, max_frames=None, batch_size=32768, graphed=True, dense=None):
    """Returns the list of per-frame result dicts (or calls on_frame(i, pose, result) and keeps nothing; the tensors
    handed to on_frame belong to the graph and are overwritten by the next frame).
    dense: a DenseTravelPoseSampler, or the DensePoseFuture of DenseTravelPoseSampler.start(pose_sampler, n_poses) issued
    earlier (e.g. before the scene was trained): the 10,000-step tour annealing has then run beside the GPU work and the
    frame loop starts at once.  Without it the trajectory is started here and the frame graph is captured meanwhile."""
    if dense is None:
        dense = DenseTravelPoseSampler.start(pose_sampler, n_dense_poses=n_poses)
    frame_fn = scene.make_graphed_render(height, width, tuple(query_keys), batch_size=batch_size) if graphed else None
    if isinstance(dense, DensePoseFuture):
        dense = dense.result()
    frames = []
    n = dense.n_poses if max_frames is None else min(dense.n_poses, max_frames)
    for i in range(n):
        pose = dense.sample_pose(i).clone().float()
        pose[:3, :3] = torch.eye(3, device=pose.device)                                # core_exp_runner.py:232
        if frame_fn is not None:
            res = frame_fn(pose)
        else:
            from .scene import Rays, gen_pano_rays
            rays = gen_pano_rays(pose, height, width)
            res = scene.render(Rays(rays.o, rays.d), query_keys=list(query_keys), batch_size=batch_size)
        if on_frame is not None:
            on_frame(i, pose, res)
        else:
            frames.append({k: v.clone() for k, v in res.items()} if frame_fn is not None else res)
    return frames
