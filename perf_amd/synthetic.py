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


ROOM_SCALE = 1.05 * (0.9 ** 2 + 0.7 ** 2 + 0.5 ** 2) ** 0.5      # room() normalises distances by (largest distance) * 1.05


def room_with_box(o: torch.Tensor, d: torch.Tensor, half=(0.9, 0.7, 0.5), box_c=(0.55, -0.35, -0.15), box_h=(0.12, 0.15, 0.3)):
    """The room of room() in NORMALISED units (the units of the registered distances and of pose translations) seen from
    arbitrary origins inside it, with an axis-aligned box standing in it -- so that views from other positions see surfaces
    the first panorama does not (disocclusion: what PeRF's visibility masks are about).  o, d [..., 3] (d unit) ->
    (distance [..., 1], rgb [..., 3]).  Same wall texture as room(); the box is textured with a finer pattern."""
    dev = d.device
    h = torch.tensor(half, dtype=torch.float32, device=dev) / ROOM_SCALE
    dd = torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
    t_wall = ((torch.sign(dd) * h - o) / dd)
    dist_w, axis_w = t_wall.min(-1)
    bc = torch.tensor(box_c, dtype=torch.float32, device=dev) / ROOM_SCALE
    bh = torch.tensor(box_h, dtype=torch.float32, device=dev) / ROOM_SCALE
    t1 = (bc - bh - o) / dd; t2 = (bc + bh - o) / dd
    tn, tf = torch.minimum(t1, t2), torch.maximum(t1, t2)
    t_in, axis_b = tn.max(-1)
    hit = (t_in < tf.min(-1).values) & (t_in > 1e-6)
    dist = torch.where(hit, t_in, dist_w)
    axis = torch.where(hit, axis_b, axis_w)
    p = (o + d * dist[..., None]) * ROOM_SCALE
    uv = torch.tensor([[1, 2], [0, 2], [0, 1]], device=dev)[axis]
    u = torch.gather(p, -1, uv[..., :1])[..., 0]
    v = torch.gather(p, -1, uv[..., 1:])[..., 0]
    k = torch.where(hit, torch.full_like(dist, 48.), torch.tensor([8., 16., 32.], device=dev)[axis])
    base = 0.5 + 0.5 * torch.sin(k * u) * torch.sin(k * v)
    tint = torch.tensor([[1.0, 0.6, 0.4], [0.4, 1.0, 0.6], [0.5, 0.6, 1.0]], device=dev)[axis]
    tint = torch.where(hit[..., None], tint.flip(-1), tint)
    sgn = torch.gather(torch.sign(d), -1, axis[..., None])[..., 0]
    rgb = (base[..., None] * tint) * (0.75 + 0.25 * sgn[..., None])
    return dist[..., None], rgb.clamp(0, 1)


# ---- two more scene families (camera at the origin, one panorama): what the room does not exercise -------------------------
def _room_and_boxes(d: torch.Tensor, half, boxes):
    """The box room of room() with axis-aligned boxes (centre, half size) standing in it, seen from the origin.  d [..., 3] unit
    directions -> (distance [..., 1] normalised by max * 1.05 as dataset.py:97-101, rgb [..., 3]); walls textured like room()'s,
    boxes with a finer pattern and the tints flipped."""
    dev = d.device
    h = torch.tensor(half, dtype=torch.float32, device=dev)
    dd = torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
    dist, axis = (h / dd.abs()).min(-1)
    hit_any = torch.zeros_like(dist, dtype=torch.bool)
    for c, hs in boxes:
        c = torch.tensor(c, dtype=torch.float32, device=dev); hs = torch.tensor(hs, dtype=torch.float32, device=dev)
        t1 = (c - hs) / dd; t2 = (c + hs) / dd
        tn, tf = torch.minimum(t1, t2), torch.maximum(t1, t2)
        t_in, ax = tn.max(-1)
        hit = (t_in < tf.min(-1).values) & (t_in > 1e-6) & (t_in < dist)
        dist = torch.where(hit, t_in, dist)
        axis = torch.where(hit, ax, axis)
        hit_any = torch.where(hit, torch.ones_like(hit_any), hit_any)
    p = d * dist[..., None]
    uv = torch.tensor([[1, 2], [0, 2], [0, 1]], device=dev)[axis]
    u = torch.gather(p, -1, uv[..., :1])[..., 0]
    v = torch.gather(p, -1, uv[..., 1:])[..., 0]
    k = torch.where(hit_any, torch.full_like(dist, 48.), torch.tensor([8., 16., 32.], device=dev)[axis])
    base = 0.5 + 0.5 * torch.sin(k * u) * torch.sin(k * v)
    tint = torch.tensor([[1.0, 0.6, 0.4], [0.4, 1.0, 0.6], [0.5, 0.6, 1.0]], device=dev)[axis]
    tint = torch.where(hit_any[..., None], tint.flip(-1), tint)
    sgn = torch.gather(torch.sign(d), -1, axis[..., None])[..., 0]
    rgb = (base[..., None] * tint) * (0.75 + 0.25 * sgn[..., None])
    return (dist / (dist.max() * 1.05))[..., None], rgb.clamp(0, 1)


def doorway(d: torch.Tensor):
    """Two rooms joined by a doorway: the camera stands in the smaller one, a partition wall at x = 0.30 (3 cm thick) has a door
    opening (0.30 wide, from the floor to z = 0.2) through which rays run on into the second room -- a long thin free space behind
    an occluder, depth discontinuities of 3x along the door frame, a thin wall seen edge-on from grazing rays."""
    part = [((0.315, -0.425, 0.0), (0.015, 0.275, 0.5)), ((0.315, 0.425, 0.0), (0.015, 0.275, 0.5)), ((0.315, 0.0, 0.35), (0.015, 0.15, 0.15))]
    return _room_and_boxes(d, (0.95, 0.7, 0.5), part)


def pillars(d: torch.Tensor):
    """The room with fourteen thin square pillars (6 cm wide, floor to ceiling) on two rings around the camera: high-frequency
    occupancy, many short free spans, most rays pass several pillar edges within a few occupancy cells."""
    import math
    boxes = []
    for ring, (rad, n, ph) in enumerate(((0.32, 6, 0.2), (0.55, 8, 0.55))):
        for i in range(n):
            a = ph + 2 * math.pi * i / n
            x, y = rad * math.cos(a), rad * 0.78 * math.sin(a)
            boxes.append(((x, y, 0.0), (0.03, 0.03, 0.5)))
    return _room_and_boxes(d, (0.9, 0.7, 0.5), boxes)


SCENES = {'room': room, 'doorway': doorway, 'pillars': pillars}
