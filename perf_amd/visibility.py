"""Reprojection visibility tests around the renderer: NeRFScene.get_pano_visibility_mask
(modules/scene/nerf.py:321-358) and SupInfoPool.geo_check (modules/dataset/sup_info.py:261-302).  They sit directly
either side of a full-panorama render: back-project the rendered distance, look the point up in every registered
panorama's distance map (bilinear grid_sample, border padding) and clean the binary result with elliptical
morphology.  On the device both steps are HIP kernels (perf_pano_reproject: one launch per registered panorama instead of
~15 torch kernels; perf_morph_binary); the torch formulation below is kept as their test reference (`use_kernels=False`).
The structuring elements restate OpenCV's MORPH_ELLIPSE rasterisation and kornia's geodesic-border dilation/erosion (both
packages are absent here: the elements and border rules are checked against scipy.ndimage on the CPU, and
direction_to_img_coord against the reference's golden vector)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def direction_to_img_coord(dirs):
    """utils/camera_utils.py:134-151: unit direction -> (row, col) image coordinates in [0,1]."""
    d = dirs / torch.linalg.norm(dirs, 2, -1, True)
    beta = torch.arcsin(d[..., 2])
    alpha = torch.atan2(d[..., 1], d[..., 0])
    return torch.stack([-beta / np.pi + .5, -(alpha / (2. * np.pi)) + .5], -1)


def img_coord_to_sample_coord(coords):
    """utils/camera_utils.py:180-181: (row, col) in [0,1] -> grid_sample's (x, y) in [-1,1]."""
    return torch.stack([coords[..., 1], coords[..., 0]], -1) * 2. - 1.


def ellipse_kernel(rows, cols, device=None):
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (cols, rows)): row i covers columns c-dx .. c+dx with
    dx = round(c * sqrt(1 - (i-r)^2 / r^2)), r = rows // 2, c = cols // 2."""
    r, c = rows // 2, cols // 2
    k = torch.zeros(rows, cols, device=device)
    for i in range(rows):
        dy = i - r
        if abs(dy) <= r:
            dx = int(round(c * math.sqrt(max(r * r - dy * dy, 0) / float(max(r * r, 1)))))
            k[i, max(c - dx, 0): min(c + dx + 1, cols)] = 1.
    return k


def dilate(mask, kernel):
    """Binary dilation, [1,1,H,W] float mask; outside the image counts as background (kornia geodesic border)."""
    ph, pw = kernel.shape[0] // 2, kernel.shape[1] // 2
    hit = F.conv2d(mask, kernel.flip(0, 1)[None, None], padding=(ph, pw))
    return (hit > 0.5).float()


def erode(mask, kernel):
    """Binary erosion; outside the image counts as foreground (borders do not erode)."""
    return 1.0 - dilate(1.0 - mask, kernel.flip(0, 1))


def _lookup_distance(pts, info):
    """Distance stored in panorama `info` along the direction of pts, and the distance of pts from its centre."""
    pose = info['pose']
    local = torch.matmul(pose[:3, :3].T, (pts - pose[:3, 3])[..., None])[..., 0]
    dist = torch.linalg.norm(local, 2, -1, True)
    coords = img_coord_to_sample_coord(direction_to_img_coord(local / dist))
    dmap = (info['distance_map'] * info['mask'].float()).permute(2, 0, 1)[None]
    proj = F.grid_sample(dmap, coords[None], padding_mode='border', align_corners=False)[0].permute(1, 2, 0)
    return dist, proj


def _kernel_path(pts, sup_infos, mode, small, large):
    from . import ops
    h, w = pts.shape[:2]
    flat = pts.reshape(-1, 3).contiguous().float()
    mask = torch.full((h * w,), 0.0 if mode == 0 else 1.0, device=pts.device)
    for info in sup_infos:
        dmap = (info['distance_map'] * info['mask'].float())[..., 0].contiguous().float()
        ops.pano_reproject(flat, info['pose'], dmap, mask, mode)
    m = mask.view(h, w)
    m = ops.morph_binary(m, ellipse_kernel(*small).flip(0, 1), 'dilate')
    return ops.morph_binary(m, ellipse_kernel(*large), 'erode')


def pano_visibility_mask(rays_o, rays_d, distance, sup_infos, use_kernels=True):
    """nerf.py:321-358: 1 where the rendered surface point is seen (not occluded) by at least one registered panorama
    (distance < stored + 1/256), then dilate 5x5 / erode 9x9 ellipses.  rays [H,W,3], distance [H,W] -> [H,W]."""
    h, w = distance.shape
    pts = rays_o + rays_d * distance[..., None]
    if use_kernels and pts.is_cuda:
        return _kernel_path(pts, sup_infos, 0, (5, 5), (9, 9))
    mask = torch.zeros(h, w, 1, device=distance.device)
    for info in sup_infos:
        dist, proj = _lookup_distance(pts, info)
        mask = torch.maximum(mask, (dist < proj + 1 / 256.).float())
    m = (mask.permute(2, 0, 1)[None] > 0.5).float()
    m = dilate(m, ellipse_kernel(5, 5, m.device))
    m = erode(m, ellipse_kernel(9, 9, m.device))
    return m[0, 0]


def geo_check(rays_o, rays_d, distances, sup_infos, use_kernels=True):
    """sup_info.py:261-302: 1 = consistent, 0 = the point lies in front of a surface some panorama already observed
    (stored distance >= distance of the point); dilate 3x3 / erode 9x9."""
    h, w = distances.shape[:2]
    pts = rays_o + rays_d * distances.reshape(h, w)[..., None]
    if use_kernels and pts.is_cuda:
        return _kernel_path(pts, sup_infos, 1, (3, 3), (9, 9))
    mask = torch.ones(h, w, 1, device=pts.device)
    for info in sup_infos:
        dist, proj = _lookup_distance(pts, info)
        mask = torch.minimum(mask, (proj < dist).float())
    m = (mask.permute(2, 0, 1)[None] > 0.5).float()
    m = dilate(m, ellipse_kernel(3, 3, m.device))
    m = erode(m, ellipse_kernel(9, 9, m.device))
    return m[0, 0]
