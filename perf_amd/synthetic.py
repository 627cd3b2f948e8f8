"""Synthetic panorama of SURVEY.md section 8(d) (the reference's example_data/kitchen/image.png is absent):
an axis-aligned box room seen from the origin, analytic ray->wall distance normalised by max*1.05 as
modules/dataset/dataset.py:97-101, procedural sinusoidal wall albedo.  Pure torch, runs on any device."""
import torch


def room(d: torch.Tensor, half=(0.9, 0.7, 0.5)):
    """d [..., 3] unit directions -> (distance [..., 1], rgb [..., 3])."""
    dev = d.device
    h = torch.tensor(half, dtype=torch.float32, device=dev)
    t = h / d.abs().clamp_min(1e-12)
    dist, axis = t.min(-1)
    p = d * dist[..., None]
    uv = torch.tensor([[1, 2], [0, 2], [0, 1]], device=dev)[axis]
    u = torch.gather(p, -1, uv[..., :1])[..., 0]
    v = torch.gather(p, -1, uv[..., 1:])[..., 0]
    k = torch.tensor([8., 16., 32.], device=dev)[axis]
    base = 0.5 + 0.5 * torch.sin(k * u) * torch.sin(k * v)
    tint = torch.tensor([[1.0, 0.6, 0.4], [0.4, 1.0, 0.6], [0.5, 0.6, 1.0]], device=dev)[axis]
    sgn = torch.gather(torch.sign(d), -1, axis[..., None])[..., 0]
    rgb = (base[..., None] * tint) * (0.75 + 0.25 * sgn[..., None])
    scale = dist.max() * 1.05
    return (dist / scale)[..., None], rgb.clamp(0, 1)
