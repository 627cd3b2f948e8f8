"""CPU oracle (numpy / torch-CPU, fp32) for PeRF's panoramic-NeRF hot path.

TEST INFRASTRUCTURE ONLY -- never imported by perf_amd/ (the product path).

Parity status
-------------
* PINNED against the reference's own code (golden vectors made by importing
  /root/reference, see tests/golden/make_fixtures.py): pano/pers ray generation
  (utils/camera_utils.py:113-147,229-241), trunc_exp and contract_to_unisphere
  (modules/fields/ngp_nerf.py:24-65), the occupancy pre-grid splat and the batch
  sampler (modules/dataset/sup_info.py:236-259,304-330), the LR schedule
  (modules/scene/nerf.py:300-311) and the renderer glue
  (modules/scene/nerf_renderer.py:112-209, run on top of this oracle's operators).
* PARITY UNPINNED for the arithmetic that lives in third-party packages that are
  not in /root/reference and cannot be installed here: tinycudann==1.7
  (requirements.txt:34), nerfacc==0.5.3 (requirements.txt:16),
  torch_efficient_distloss==0.1.3 (requirements.txt:36).  Their published
  algorithms are restated below; the reference holds no
  test or golden vector for them.

Where this restatement DEFINES a rounding order that upstream plausibly does differently
(not checkable here; it would change ray_indices / t_starts against the real packages):
* (grid position: follows tiny-cuda-nn -- ONE rounding, fmaf(scale, x, 0.5), see grid_pos; rounds 1-2 of this
  repository used fl(fl(x*scale) + 0.5), under which a position within half an ulp of a cell boundary could land in
  the neighbouring cell.)
* marching lattice: ONE global lattice per ray anchored at the near plane (+ stratified jitter), an interval being
  emitted iff its midpoint lies in an occupied cell (occ_march).  DEFAULT_LATTICE = 'repeated':
  t_0 = t0, t_{k+1} = fl(t_k + step) -- nerfacc's traverse_grids advances by REPEATED float addition (t_last += dt, in
  occupied and in empty cells alike, as v0.5.3 is recalled: "march until t_mid is right after t_traverse"), so its t_k
  carry the accumulated rounding of k additions (<= k/2 ulp, ~1e-6 after 3,000 steps of 5e-4).  Rounds 1-3 of this
  repository defaulted to 'single', t_k = fl(t0 + fl(k*step)) (one rounding; still selectable, bit-exact tests for
  both).  Whether upstream keeps stepping on the same lattice through empty cells or restarts its intervals at the
  cell entry cannot be checked here: under the first reading (ours) sample counts agree and, with 'repeated',
  t_starts agree to the bit; under the second, counts per occupied span agree to +-1 and positions differ by < step.
  tools/pin_upstream.py lets a maintainer who holds nerfacc decide it with one run.
* early termination: thresholded on the canonical-order exclusive sum (ex <= -ln eps) instead of on
  T = exp(-ex) >= eps (identical decision up to the rounding of exp).

Conventions
-----------
All bookkeeping arithmetic (lattice times, cell indices, scans that feed a threshold) is defined with
*unfused* IEEE fp32 multiplies and adds so that numpy and the HIP kernels (built with explicit
__fmul_rn/__fadd_rn) agree bit for bit; the one fused operation is the grid position (grid_pos).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

F32 = np.float32
DEFAULT_LATTICE = 'repeated'       # see the header; 'single' = t_k = fl(t0 + fl(k*step))
U32_MASK = 0xFFFFFFFF
PRIME_Y = 2654435761
PRIME_Z = 805459861


# --------------------------------------------------------------------------------------
# a1  ray generation  (reference: utils/camera_utils.py:113-147, 229-241)
# --------------------------------------------------------------------------------------
def pano_rays(pose: torch.Tensor, height: int, width: int):
    """Equirectangular rays.  Follows gen_pano_rays (camera_utils.py:229-234):
    pixel centres (i+.5)/H,(j+.5)/W  (img_coord_from_hw :113-117) ->
    beta=-(y-.5)pi, alpha=-(x-.5)2pi (img_to_pano_coord :120-126) ->
    d=(cos a cos b, sin a cos b, sin b) (pano_coord_to_direction :142-147) ->
    d <- R d (apply_rot :44-46); o <- pose[:3,3] broadcast."""
    pose = pose.to(torch.float32)
    i = torch.linspace(.5 / height, 1. - .5 / height, height)
    j = torch.linspace(.5 / width, 1. - .5 / width, width)
    beta = -(i - .5) * np.pi
    alpha = -(j - .5) * 2. * np.pi
    cb, sb = torch.cos(beta)[:, None], torch.sin(beta)[:, None]
    ca, sa = torch.cos(alpha)[None, :], torch.sin(alpha)[None, :]
    d = torch.stack([ca * cb, sa * cb, sb.expand(height, width)], -1)
    d = torch.matmul(pose[:3, :3], d[..., None])[..., 0]
    o = pose[:3, 3].expand(height, width, 3).contiguous()
    return o, d


def pers_rays(pose: torch.Tensor, fov: float, res: int):
    """Perspective rays, OpenCV style.  Follows gen_pers_rays (camera_utils.py:237-241)
    with cam_rays_cam_space (:55-76): x,y = linspace(-tan(fov/2), tan(fov/2), res)."""
    pose = pose.to(torch.float32)
    span = np.tan(fov * .5)
    y = torch.linspace(-span, span, res)
    x = torch.linspace(-span, span, res)
    yy, xx = torch.meshgrid(y, x, indexing='ij')
    xyz = torch.stack([xx, yy, torch.ones_like(xx)], -1)
    d = xyz / torch.linalg.norm(xyz, 2, -1, True)
    o = torch.zeros_like(d) + pose[:3, 3]
    d = torch.matmul(pose[:3, :3], d[..., None])[..., 0]
    return o, d


# --------------------------------------------------------------------------------------
# a2/a3  hash grid + bias-free MLP  (tinycudann==1.7 semantics, SURVEY.md A.1/A.2;
#        call sites modules/fields/ngp_nerf.py:96-134)
# --------------------------------------------------------------------------------------
@dataclass
class GridLevels:
    n_levels: int
    n_feat: int
    scale: np.ndarray      # f32 [L]
    res: np.ndarray        # u32 [L]
    size: np.ndarray       # u32 [L]  entries in the level
    offset: np.ndarray     # u32 [L]  first entry of the level
    hashed: np.ndarray     # bool [L]
    total: int             # entries in all levels
    layout: str = 'tcnn'   # 'tcnn' | 'line_local' | 'line_overlap' (see grid_levels)
    sb_shift: tuple = (5, 6, 8)
    local: np.ndarray = None   # bool [L]  level stored line-local
    nsx: np.ndarray = None     # u32 [L]   dense line-local levels: super-blocks per row
    nsxy: np.ndarray = None    # u32 [L]   ... per z-slice

    @property
    def n_params(self):
        return self.total * self.n_feat


LOCAL_MIN_RES = 64          # line_local grids: levels of at least this resolution are stored line-local (default)
SB_SHIFT = {'tcnn': (5, 6, 8), 'line_local': (5, 6, 8), 'line_overlap': (7, 5, 7)}       # default super-block shapes (perf_amd.grid.SB_SHIFT)


def grid_levels(n_levels=16, n_feat=2, log2_hashmap_size=18, base_resolution=16,
                per_level_scale=1.4472692012786865, layout='tcnn', sb_shift=None, local_min_res=LOCAL_MIN_RES) -> GridLevels:
    """Per-level geometry of a tcnn HashGrid (A.1): scale_l = N_min*b^l - 1 (fp32),
    res_l = ceil(scale_l)+1, size_l = min(align8(res_l^3), 2^T), offsets = prefix sums.

    layout='line_local' (NOT tcnn's; an opt-in table layout of this build for grids the reference never defines -- BASELINE
    config 5's L = 20 tables sized to HBM, inference only; no reference result exists to stay compatible with): levels with
    res >= local_min_res store the vertices of a 4 x 4 x 2 block as ONE 128-byte line (entry x%4 + 4 (y%4) + 16 (z%2)),
    the blocks of a 2^sx x 2^sy x 2^sz-vertex SUPER-BLOCK contiguously (x-major; default 32 x 64 x 256 vertices = 2 MiB), and
    address the super-block either densely (index sx + sy*nx + sz*nx*ny when all n_d = (res + 2^s_d) >> s_d super-blocks per
    dimension fit 2^T entries) or through tcnn's prime-XOR hash OF THE SUPER-BLOCK COORDINATES modulo the number of
    super-blocks in 2^T entries.  A sample's eight corners then lie in (1+1/4)(1+1/4)(1+1/2) = 2.3 lines of ONE page instead
    of four lines on four pages.  A line-local level starts at a multiple of the super-block size (the entries between the levels
    are padding).  Coarser levels keep tcnn's rule.  Same scale / res / interpolation as tcnn's grid.

    layout='line_overlap': line_local whose 16-byte x runs OVERLAP by one vertex, so that the x corner pair of a cell never straddles
    two runs: the pair of CELL gx lives in run gx // 3 of its row at positions gx % 3, gx % 3 + 1 -- storage coordinates
    X = gx + gx // 3 and X + 1 of the line_local rule.  Position 3 of a run repeats position 0 of the next run of the same
    super-block (whoever writes the table keeps the two equal: canonical_overlap_fill); the last cell of a super-block row takes its
    second corner from the NEXT super-block's first run (storage X + 2), so no vertex is stored in two super-blocks.  A sample's
    eight corners lie in (1+1/4)(1+1/2) = 1.9 lines; a super-block row holds 3/4 as many cells (hashed levels: 3/4 as many distinct
    vertices in the same 2^T entries; dense levels grow by up to 4/3)."""
    assert layout in ('tcnn', 'line_local', 'line_overlap')
    sb_shift = tuple(SB_SHIFT[layout] if sb_shift is None else sb_shift)
    per_sb = 1 << sum(sb_shift)
    log2_b = F32(np.log2(F32(per_level_scale)))
    scale = np.zeros(n_levels, F32)
    res = np.zeros(n_levels, np.uint32)
    size = np.zeros(n_levels, np.uint32)
    offset = np.zeros(n_levels, np.uint64)
    hashed = np.zeros(n_levels, bool)
    local = np.zeros(n_levels, bool)
    nsx = np.zeros(n_levels, np.uint32)
    nsxy = np.zeros(n_levels, np.uint32)
    total = 0
    for l in range(n_levels):
        e = F32(np.exp2(np.float64(F32(l) * log2_b)))
        s = F32(F32(e * F32(base_resolution)) - F32(1.0))
        r = int(math.ceil(float(s))) + 1
        if layout != 'tcnn' and r >= local_min_res:
            nd = [(r + (1 << sh)) >> sh for sh in sb_shift]          # super-blocks per dimension: vertices 0..res
            if layout == 'line_overlap':                             # cells 0..res-1 at storage x = gx + gx // 3 (+ 2 for a row's last cell)
                nd[0] = (((r - 1) + (r - 1) // 3 + 2) >> sb_shift[0]) + 1
            full = nd[0] * nd[1] * nd[2] * per_sb
            n = min(full, 1 << log2_hashmap_size)
            assert n >= per_sb, f'{layout}: 2^log2_hashmap_size must hold at least one super-block'
            local[l], nsx[l], nsxy[l] = True, nd[0], nd[0] * nd[1]
            total = -(-total // per_sb) * per_sb             # the level starts on a super-block boundary (its 128-byte blocks are cache lines)
        else:
            full = r ** 3
            n = min(full, U32_MASK // 2)
            n = (n + 7) // 8 * 8
            n = min(n, 1 << log2_hashmap_size)
        scale[l], res[l], size[l], offset[l] = s, r, n, total
        hashed[l] = full > n
        total += n
    return GridLevels(n_levels, n_feat, scale, res, size, offset, hashed, total, layout, tuple(sb_shift), local, nsx, nsxy)


def _quant(t: torch.Tensor, quant):
    if quant is None:
        return t
    dt = {'bf16': torch.bfloat16, 'fp16': torch.float16}[quant]
    return t.to(dt).to(torch.float32)


def grid_pos(x: np.ndarray, scale) -> np.ndarray:
    """pos = fl32(x*scale + 0.5) with ONE rounding: tiny-cuda-nn's pos_fract computes fmaf(scale, input, 0.5f).
    numpy has no fma: the product of two fp32 is exact in fp64 and the sum is formed in fp64; rounding that to fp32
    differs from rounding the exact sum only if the fp64 sum sits exactly half way between two fp32 neighbours (its low
    29 mantissa bits are 1000...0) while the fp64 addition was inexact -- those (rare) entries are resolved by the sign of
    the addition's rounding error (TwoSum)."""
    p = np.asarray(x, F32).astype(np.float64) * np.float64(F32(scale))
    s = p + 0.5
    r = s.astype(F32)
    cand = np.flatnonzero((s.view(np.int64) & 0x1FFFFFFF) == 0x10000000)
    if cand.size:
        pc, sc = p.reshape(-1)[cand], s.reshape(-1)[cand]
        bb = sc - pc
        e = (pc - (sc - bb)) + (0.5 - bb)                 # sc + e == pc + 0.5 exactly
        rc = sc.astype(F32)
        d = sc - rc.astype(np.float64)                    # +- half an fp32 gap (a tie, rounded to even)
        nb = np.nextafter(rc, np.where(d > 0, np.inf, -np.inf).astype(F32))
        r = r.copy()
        r.reshape(-1)[cand] = np.where(e * d > 0, nb, rc)
    return r


def grid_corner_indices(x: np.ndarray, lv: GridLevels, level: int):
    """Bit-exact integer bookkeeping of one level: returns (idx u32 [N,8], frac f32 [N,3]).
    pos = fma(x, scale, 0.5) (grid_pos); g = floor(pos); corner c: bit0->x, bit1->y, bit2->z.
    dense: (gx + gy*res + gz*res^2) mod 2^32 mod size; hashed: gx ^ gy*P1 ^ gz*P2 (uint32) mod size."""
    x = np.ascontiguousarray(x, F32)
    s = lv.scale[level]
    pos = grid_pos(x, s)
    g = np.floor(pos)
    frac = (pos - g).astype(F32)
    gi = g.astype(np.int64) & U32_MASK
    r = int(lv.res[level]); n = int(lv.size[level])
    idx = np.zeros((x.shape[0], 8), np.uint32)
    for c in range(8):
        cx = (gi[:, 0] + (c & 1)) & U32_MASK
        cy = (gi[:, 1] + ((c >> 1) & 1)) & U32_MASK
        cz = (gi[:, 2] + ((c >> 2) & 1)) & U32_MASK
        if lv.local is not None and lv.local[level]:
            shx, shy, shz = lv.sb_shift
            per_sb = 1 << (shx + shy + shz)
            if lv.layout == 'line_overlap':                          # storage x of the CELL's corner (see grid_levels)
                x0 = gi[:, 0] + gi[:, 0] // 3
                m = (1 << shx) - 1
                cx = (x0 if (c & 1) == 0 else np.where((x0 & m) == m - 1, x0 + 2, x0 + 1)) & U32_MASK
            sx, sy, sz = cx >> shx, cy >> shy, cz >> shz
            if lv.hashed[level]:
                slot = (sx ^ ((sy * PRIME_Y) & U32_MASK) ^ ((sz * PRIME_Z) & U32_MASK)) % (n // per_sb)
            else:
                slot = sx + sy * int(lv.nsx[level]) + sz * int(lv.nsxy[level])
            blk = ((cx >> 2) & ((1 << (shx - 2)) - 1)) + (((cy >> 2) & ((1 << (shy - 2)) - 1)) << (shx - 2)) \
                + (((cz >> 1) & ((1 << (shz - 1)) - 1)) << (shx - 2 + shy - 2))
            idx[:, c] = (slot * per_sb + (blk << 5) + (cx & 3) + ((cy & 3) << 2) + ((cz & 1) << 4)).astype(np.uint32)
            continue
        if lv.hashed[level]:
            h = cx ^ ((cy * PRIME_Y) & U32_MASK) ^ ((cz * PRIME_Z) & U32_MASK)
        else:
            h = (cx + cy * r + cz * r * r) & U32_MASK
        idx[:, c] = (h % n).astype(np.uint32)
    return idx, frac


def canonical_overlap_fill(table: np.ndarray, lv: GridLevels) -> np.ndarray:
    """layout='line_overlap': make a table [total, F] a valid one -- position 3 of every 16-byte x run := position 0 of the next run
    of the same super-block row (the two entries are ONE vertex); the last run of a row keeps its (never read) last entry.
    Restated by perf_amd.fields.canonical_overlap_fill_ on the device."""
    assert lv.layout == 'line_overlap'
    out = np.array(table, copy=True)
    shx = lv.sb_shift[0]
    runs = 1 << (shx - 2)                                            # x runs (= blocks along x) of a super-block row
    for l in range(lv.n_levels):
        if not lv.local[l]:
            continue
        lo, n = int(lv.offset[l]), int(lv.size[l])
        v = out[lo:lo + n].reshape(n // (32 * runs), runs, 8, 4, -1)  # [row of blocks, block along x, (y, z) of the block, x in run, F]
        v[:, :-1, :, 3] = v[:, 1:, :, 0]
    return out


def hashgrid_encode(x: torch.Tensor, table: torch.Tensor, lv: GridLevels,
                    interpolation: str = 'Linear', quant=None) -> torch.Tensor:
    """x [N,3] in grid coordinates, table [total, F] fp32 -> features [N, L*F] fp32.
    Differentiable w.r.t. table and x (autograd supplies first and second order).  All levels share one
    gather (one index_put in backward)."""
    N = x.shape[0]
    tq = _quant(table, quant)
    xn = x.detach().cpu().numpy().astype(F32)
    idx_all, w_all = [], []
    for l in range(lv.n_levels):
        idx_np, frac_np = grid_corner_indices(xn, lv, l)
        idx_all.append(torch.from_numpy(idx_np.astype(np.int64)) + int(lv.offset[l]))
        s = float(lv.scale[l])
        f = torch.from_numpy(frac_np) + (x - x.detach()) * s        # value: the fma's fractional part; d/dx = scale
        if interpolation == 'Smoothstep':
            f = f * f * (3.0 - 2.0 * f)
        ws = []
        for c in range(8):
            wx = f[:, 0] if (c & 1) else 1.0 - f[:, 0]
            wy = f[:, 1] if (c & 2) else 1.0 - f[:, 1]
            wz = f[:, 2] if (c & 4) else 1.0 - f[:, 2]
            ws.append((wx * wy) * wz)
        w_all.append(torch.stack(ws, -1))
    idx = torch.stack(idx_all, 1)                       # [N, L, 8]
    w = torch.stack(w_all, 1)                           # [N, L, 8]
    vals = tq[idx.reshape(-1)].view(N, lv.n_levels, 8, lv.n_feat)
    return (w[..., None] * vals).sum(2).reshape(N, lv.n_levels * lv.n_feat)


def mlp_shapes(n_in: int, n_hidden_layers: int, width: int = 64, n_out_padded: int = 16):
    shapes = [(width, n_in)] + [(width, width)] * (n_hidden_layers - 1) + [(n_out_padded, width)]
    return shapes


def mlp_forward(x: torch.Tensor, net_params: torch.Tensor, n_in: int, n_hidden_layers: int,
                n_out: int, output_activation: str = 'None', quant=None, width: int = 64):
    """Bias-free fully connected net (A.2): row-major [out,in] matrices, ReLU hidden,
    output rows padded to 16, only the first n_out are returned.  With quant set, operands
    of every matmul are rounded to the 16-bit type and products accumulate in fp32, which
    is what the MFMA kernels do."""
    h = _quant(x, quant)
    off = 0
    shapes = mlp_shapes(n_in, n_hidden_layers, width)
    for li, (o, i) in enumerate(shapes):
        W = _quant(net_params[off:off + o * i].view(o, i), quant)
        off += o * i
        h = h @ W.t()
        if li < len(shapes) - 1:
            h = _quant(torch.relu(h), quant)
    y = h[:, :n_out]
    if output_activation == 'Sigmoid':
        y = torch.sigmoid(y)
    return y


def n_mlp_params(n_in, n_hidden_layers, width=64):
    return sum(o * i for o, i in mlp_shapes(n_in, n_hidden_layers, width))


class _TruncExp(torch.autograd.Function):
    """exp forward; backward multiplies by exp(min(x,15)) (ngp_nerf.py:24-40)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(torch.clamp(x, max=15))


trunc_exp = _TruncExp.apply


def contract_to_unisphere(x, aabb):
    """ngp_nerf.py:43-65 (forward branch)."""
    lo, hi = aabb[:3], aabb[3:]
    x = (x - lo) / (hi - lo)
    x = x * 2 - 1
    mag = x.norm(dim=-1, keepdim=True)
    mask = mag.squeeze(-1) > 1
    x = x.clone()
    x[mask] = (2 - 1 / mag[mask]) * (x[mask] / mag[mask])
    return x / 4 + 0.5


@dataclass
class FieldSpec:
    lv: GridLevels
    n_hidden_layers: int
    n_out: int
    output_activation: str

    @property
    def n_in(self):
        """Input width of the network: the encoding's output padded to a multiple of 16 (tcnn pads the encoding to the
        network's alignment; the padded columns see zeros).  32 for PeRF's 16 levels, 48 for a 20-level grid."""
        return (self.lv.n_levels * self.lv.n_feat + 15) // 16 * 16

    @property
    def n_net(self):
        return n_mlp_params(self.n_in, self.n_hidden_layers)

    @property
    def n_params(self):
        return self.n_net + self.lv.n_params


def geo_spec():
    return FieldSpec(grid_levels(), 1, 1, 'None')


def app_spec():
    return FieldSpec(grid_levels(), 2, 3, 'Sigmoid')


def init_field_params(spec: FieldSpec, seed: int = 1337) -> torch.Tensor:
    """Flat fp32 params = [network | grid] (A.2): Xavier-uniform matrices, grid U(-1e-4,1e-4).
    (tcnn's own RNG stream is not reproducible here; the same tensor is fed to both sides.)"""
    g = torch.Generator().manual_seed(seed)
    parts = []
    for (o, i) in mlp_shapes(spec.n_in, spec.n_hidden_layers):
        s = math.sqrt(6.0 / (i + o))
        parts.append((torch.rand(o * i, generator=g) * 2 - 1) * s)
    parts.append((torch.rand(spec.lv.n_params, generator=g) * 2 - 1) * 1e-4)
    return torch.cat(parts)


def network_with_encoding(x01: torch.Tensor, params: torch.Tensor, spec: FieldSpec, quant=None):
    """tcnn.NetworkWithInputEncoding.forward: x in [0,1]^3 -> [N, n_out] (fp32 here)."""
    n_net = spec.n_net
    table = params[n_net:].view(spec.lv.total, spec.lv.n_feat)
    feat = hashgrid_encode(x01, table, spec.lv, quant=quant)
    if spec.n_in > feat.shape[1]:
        feat = torch.cat([feat, feat.new_zeros(feat.shape[0], spec.n_in - feat.shape[1])], 1)
    return mlp_forward(feat, params[:n_net], spec.n_in,
                       spec.n_hidden_layers, spec.n_out, spec.output_activation, quant=quant)


def query_density(x: torch.Tensor, params, spec, aabb, quant=None, shift: float = 0.0):
    """NGPNeRF.query_density (ngp_nerf.py:136-150): normalise to the aabb, selector 0<x<1,
    sigma = trunc_exp(net(x) - shift) * selector.  shift=1 gives NGPDensityField (:247-264)."""
    lo, hi = aabb[:3], aabb[3:]
    x01 = (x - lo) / (hi - lo)
    sel = ((x01 > 0.0) & (x01 < 1.0)).all(dim=-1)
    y = network_with_encoding(x01, params, spec, quant)
    return trunc_exp(y - shift) * sel[:, None]


def query_rgb(x: torch.Tensor, params, spec, aabb, quant=None):
    """NGPNeRF.query_rgb (ngp_nerf.py:152-162)."""
    lo, hi = aabb[:3], aabb[3:]
    x01 = (x - lo) / (hi - lo)
    sel = ((x01 > 0.0) & (x01 < 1.0)).all(dim=-1)
    return network_with_encoding(x01, params, spec, quant) * sel[:, None]


# --------------------------------------------------------------------------------------
# a6  occupancy-grid marching (nerfacc==0.5.3 OccGridEstimator.sampling, SURVEY.md A.3;
#     call site modules/scene/nerf_renderer.py:145-155)
# --------------------------------------------------------------------------------------
def ray_aabb(o: np.ndarray, d: np.ndarray, aabb: np.ndarray):
    """Slab test in fp32 with fminf/fmaxf NaN semantics.  Returns (tmin, tmax)."""
    o = o.astype(F32); d = d.astype(F32); aabb = aabb.astype(F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = (F32(1.0) / d).astype(F32)
        t1 = ((aabb[None, :3] - o) * inv).astype(F32)
        t2 = ((aabb[None, 3:] - o) * inv).astype(F32)
    lo = np.fmin(t1, t2); hi = np.fmax(t1, t2)
    tmin = np.fmax(np.fmax(lo[:, 0], lo[:, 1]), lo[:, 2])
    tmax = np.fmin(np.fmin(hi[:, 0], hi[:, 1]), hi[:, 2])
    return tmin.astype(F32), tmax.astype(F32)


def lattice_t(t0: np.ndarray, k: np.ndarray, step) -> np.ndarray:
    """t_k = fl(t0 + fl(k*step)) -- the oracle's definition of the marching lattice."""
    return (t0 + (k.astype(F32) * F32(step)).astype(F32)).astype(F32)


def lattice_table_repeated(t0: np.ndarray, n: int, step) -> np.ndarray:
    """The alternative lattice of a marcher that advances by repeated addition: T[r, 0] = t0[r], T[r, k+1] = fl(T[r, k] + step),
    k < n -- np.add.accumulate is strictly sequential, every partial sum rounded to fp32.  [R, n + 1]."""
    t0 = np.asarray(t0, F32).reshape(-1)
    seq = np.concatenate([t0[:, None], np.full((t0.shape[0], n), F32(step), F32)], axis=1)
    return np.add.accumulate(seq, axis=1, dtype=F32)


def occ_cell_index(p: np.ndarray, aabb: np.ndarray, res: int) -> np.ndarray:
    """x-major linear cell index (nerf.py:154-156) of points p, clamped to the grid."""
    aabb = aabb.astype(F32)
    inv = (F32(1.0) / (aabb[3:] - aabb[:3])).astype(F32)
    u = (((p - aabb[None, :3]).astype(F32) * inv[None]).astype(F32) * F32(res)).astype(F32)
    c = np.clip(np.floor(u), 0, res - 1).astype(np.int64)
    return c[:, 0] * res * res + c[:, 1] * res + c[:, 2]


def occ_march(o, d, binaries, aabb, near, far, step, t0=None, max_steps=None, lattice=None):
    """Sampling of fixed-step intervals whose midpoint lies in an occupied cell (A.3).

    o,d [R,3] f32; binaries bool [res,res,res] (x-major); aabb [6]; t0 [R] = lattice origin
    (near, plus U[0,1)*step when stratified).  Interval k is [t_k, t_{k+1}] on the lattice `lattice`
    (None: DEFAULT_LATTICE; 'repeated': t_{k+1} = fl(t_k + step), 'single': t_k = fl(t0 + fl(k*step))); kept iff lo <= mid <= hi where mid = fl(fl(t_k+t_{k+1})*0.5),
    [lo,hi] = [max(tmin_aabb, t0), min(tmax_aabb, far)], and the cell of o+d*mid
    (unfused) is occupied.  Returns (ray_indices i64 [S], t_starts f32 [S], t_ends f32 [S],
    packed_info i32 [R,2] = (start, count)); sorted by ray then t."""
    lattice = lattice or DEFAULT_LATTICE
    assert lattice in ('single', 'repeated')
    o = np.ascontiguousarray(o, F32); d = np.ascontiguousarray(d, F32)
    aabb = np.asarray(aabb, F32)
    R = o.shape[0]
    res = binaries.shape[0]
    flat = np.ascontiguousarray(binaries).reshape(-1).astype(bool)
    if t0 is None:
        t0 = np.full(R, near, F32)
    t0 = np.asarray(t0, F32)
    tmin, tmax = ray_aabb(o, d, aabb)
    lo = np.fmax(tmin, t0)
    hi = np.fmin(tmax, F32(far))
    K = int(max_steps) if max_steps is not None else int(math.ceil((float(far) - float(near)) / float(step))) + 1
    keep_cols = []
    table = lattice_table_repeated(t0, K, step) if lattice == 'repeated' else None      # (lattice='repeated': see the header)
    for k0 in range(0, K, 256):
        ks = np.arange(k0, min(K, k0 + 256))
        if table is not None:
            ta, tb = table[:, ks], table[:, ks + 1]
        else:
            ta = lattice_t(t0[:, None], ks[None, :], step)
            tb = lattice_t(t0[:, None], ks[None, :] + 1, step)
        mid = ((ta + tb).astype(F32) * F32(0.5)).astype(F32)
        ok = (mid >= lo[:, None]) & (mid <= hi[:, None])
        rr, cc = np.nonzero(ok)
        if rr.size:
            p = (o[rr] + (d[rr] * mid[rr, cc][:, None]).astype(F32)).astype(F32)
            occ = flat[occ_cell_index(p, aabb, res)]
            rr, cc = rr[occ], cc[occ]
            keep_cols.append((rr, ks[cc], ta[rr, cc], tb[rr, cc]))
    if keep_cols:
        rr = np.concatenate([c[0] for c in keep_cols]); kk = np.concatenate([c[1] for c in keep_cols])
        ts = np.concatenate([c[2] for c in keep_cols]); te = np.concatenate([c[3] for c in keep_cols])
        order = np.lexsort((kk, rr))
        rr, ts, te = rr[order], ts[order], te[order]
    else:
        rr = np.zeros(0, np.int64); ts = np.zeros(0, F32); te = np.zeros(0, F32)
    counts = np.bincount(rr, minlength=R).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    packed = np.stack([starts, counts], -1).astype(np.int32)
    return rr.astype(np.int64), ts.astype(F32), te.astype(F32), packed


def packed_info_from_ray_indices(ray_indices: np.ndarray, n_rays: int) -> np.ndarray:
    counts = np.bincount(ray_indices, minlength=n_rays).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    return np.stack([starts, counts], -1).astype(np.int32)


# --------------------------------------------------------------------------------------
# a6  packed scans, weights, accumulation  (nerfacc scan.cu / volrend.py)
# --------------------------------------------------------------------------------------
def packed_exclusive_sum_canonical(v: np.ndarray, packed: np.ndarray) -> np.ndarray:
    """Per-ray exclusive prefix sum in the *canonical association order* that the HIP
    wave scan uses, so thresholds on it are bit-exact: each ray is cut in chunks of 64
    consecutive samples; inside a chunk an inclusive Kogge-Stone scan (offsets 1,2,..,32,
    x_i += x_{i-off} for i>=off); carry = running fp32 sum of chunk totals (left to right);
    exclusive_i = fl(carry + inclusive_{i-1}) with inclusive_{-1} = 0 (so exclusive_0 = carry)."""
    v = np.asarray(v, F32)
    out = np.zeros_like(v)
    R = packed.shape[0]
    counts = packed[:, 1].astype(np.int64)
    starts = packed[:, 0].astype(np.int64)
    if v.size == 0:
        return out
    maxc = int(counts.max())
    nch = (maxc + 63) // 64
    carry = np.zeros(R, F32)
    lane = np.arange(64)
    for ch in range(nch):
        pos = ch * 64 + lane[None, :]
        valid = pos < counts[:, None]
        src = np.where(valid, starts[:, None] + pos, 0)
        x = np.where(valid, v[src], F32(0)).astype(F32)
        for off in (1, 2, 4, 8, 16, 32):
            sh = np.zeros_like(x)
            sh[:, off:] = x[:, :-off]
            x = (x + sh).astype(F32)
        prev = np.zeros_like(x)
        prev[:, 1:] = x[:, :-1]
        ex = (carry[:, None] + prev).astype(F32)
        out[src[valid]] = ex[valid]
        carry = (carry + x[:, 63]).astype(F32)
    return out


def visibility_keep_mask(sigmas, t_starts, t_ends, packed, early_stop_eps=1e-4, alpha_thre=0.0):
    """render_visibility_from_density: keep iff T >= eps (and alpha >= alpha_thre when > 0),
    T = exp(-exclusive_sum(sigma*delta)) with the canonical scan order.  The exp itself is
    evaluated by the caller's library, so a handful of samples within 1 ulp of the threshold
    may differ between libm and the GPU; tests compare on the scan value instead."""
    sd = (np.asarray(sigmas, F32) * (np.asarray(t_ends, F32) - np.asarray(t_starts, F32)).astype(F32)).astype(F32)
    ex = packed_exclusive_sum_canonical(sd, packed)
    # T >= eps  <=>  ex <= -log(eps): thresholding on the scan keeps the decision exact.
    keep = ex <= F32(-math.log(early_stop_eps))
    if alpha_thre > 0:
        keep &= (1.0 - np.exp(-sd)) >= alpha_thre
    return keep, ex


def _to_dense(v: torch.Tensor, packed: np.ndarray, fill=0.0):
    """[S] packed -> [R, Lmax] dense plus the gather index, for differentiable scans."""
    R = packed.shape[0]
    counts = torch.from_numpy(packed[:, 1].astype(np.int64))
    starts = torch.from_numpy(packed[:, 0].astype(np.int64))
    Lmax = int(counts.max()) if R else 0
    pos = torch.arange(max(Lmax, 1))[None, :]
    valid = pos < counts[:, None]
    src = torch.where(valid, starts[:, None] + pos, torch.zeros_like(pos))
    dense = torch.where(valid, v[src], torch.full((), fill, dtype=v.dtype))
    return dense, valid, src


def packed_exclusive_sum(v: torch.Tensor, packed: np.ndarray) -> torch.Tensor:
    """Differentiable per-ray exclusive sum (plain left-to-right association)."""
    if v.numel() == 0:
        return v.clone()
    dense, valid, src = _to_dense(v, packed)
    inc = torch.cumsum(dense, 1)
    exc = inc - dense
    out = torch.zeros_like(v)
    out = out.index_put((src[valid],), exc[valid])
    return out


def render_weight_from_density(t_starts, t_ends, sigmas, packed):
    """alpha = 1-exp(-sigma*delta); T = exp(-exclusive_sum(sigma*delta)); w = T*alpha (A.4)."""
    sd = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sd)
    trans = torch.exp(-packed_exclusive_sum(sd, packed))
    return trans * alphas, trans, alphas


def accumulate_along_rays(weights, values, ray_indices, n_rays):
    """zeros(n_rays, C).index_add_(0, ray_indices, w[:,None]*v)  (A.4)."""
    src = weights[:, None] if values is None else weights[:, None] * values
    out = torch.zeros(n_rays, src.shape[-1], dtype=src.dtype)
    return out.index_add(0, ray_indices, src)


def flatten_eff_distloss(w, m, interval, ray_id):
    """torch_efficient_distloss.flatten_eff_distloss (A.5):
    (1/3 sum d_i w_i^2 + 2 sum w_i (m_i W_i - WM_i)) / n_rays, n_rays = ray_id.max()+1,
    W, WM per-ray exclusive prefixes; gradient flows to w only."""
    n_rays = int(ray_id.max().item()) + 1
    packed = packed_info_from_ray_indices(ray_id.numpy(), n_rays)
    W = packed_exclusive_sum(w, packed)
    WM = packed_exclusive_sum(w * m, packed)
    uni = (interval * w * w).sum() / 3.0
    bi = 2.0 * (w * (m * W - WM)).sum()
    return (uni + bi) / n_rays


def pdf_resample(s_in: np.ndarray, cdf: np.ndarray, n_out: int, tau=None) -> np.ndarray:
    """Inverse-CDF resampling of intervals (nerfacc importance_sampling as used by PropNetEstimator.sampling,
    nerf_renderer.py:60-70; A.6).  The path is dead in the reference and the package is absent: this is the repo's own
    definition.  edges u_j = fl(fl(j + tau_r) / (n_out+1)), j = 0..n_out (tau_r = 0.5 unless stratified);
    k = last interval with cdf_k <= u_j; t_j = s_k + (u_j-cdf_k)/(cdf_{k+1}-cdf_k) * (s_{k+1}-s_k), unfused fp32."""
    s_in = np.asarray(s_in, F32); cdf = np.asarray(cdf, F32)
    R, n1 = s_in.shape
    n_in = n1 - 1
    tau = np.full(R, 0.5, F32) if tau is None else np.asarray(tau, F32)
    j = np.arange(n_out + 1, dtype=F32)[None, :]
    u = ((j + tau[:, None]).astype(F32) / F32(n_out + 1)).astype(F32)
    out = np.zeros((R, n_out + 1), F32)
    for r in range(R):
        k = np.clip(np.searchsorted(cdf[r], u[r], side='right') - 1, 0, n_in - 1)
        c0, c1, s0, s1 = cdf[r][k], cdf[r][k + 1], s_in[r][k], s_in[r][k + 1]
        den = (c1 - c0).astype(F32)
        with np.errstate(divide='ignore', invalid='ignore'):
            frac = ((u[r] - c0).astype(F32) / den).astype(F32)
            v = (s0 + (frac * (s1 - s0).astype(F32)).astype(F32)).astype(F32)
        v = np.where(den > 0, v, s0)
        out[r] = np.minimum(np.maximum(v, s0), s1)
    return out


def prop_sampling(sigma_fns, prop_samples, num_samples, n_rays, near, far, taus=None):
    """PropNetEstimator.sampling restated (A.6): start from CDF [0,1]; per proposal level resample n edges, map
    s -> t uniformly on [near, far], query sigma at the intervals, cdf = 1 - [T, 0]; final level draws num_samples.
    sigma_fns take (t_starts, t_ends) numpy [R, n] and return sigma [R, n]."""
    s = np.concatenate([np.zeros((n_rays, 1), F32), np.ones((n_rays, 1), F32)], 1)
    cdf = s.copy()
    levels = list(zip(sigma_fns, prop_samples)) + [(None, num_samples)]
    for li, (fn, n_out) in enumerate(levels):
        s = pdf_resample(s, cdf, n_out, None if taus is None else taus[li])
        t = (F32(near) + (s * F32(far - near)).astype(F32)).astype(F32)
        ts, te = t[:, :-1], t[:, 1:]
        if fn is None:
            return ts, te
        sd = (fn(ts, te).astype(F32) * (te - ts).astype(F32)).astype(F32)
        ex = np.cumsum(sd, -1, dtype=F32) - sd
        trans = np.exp(-ex).astype(F32)
        cdf = (F32(1.0) - np.concatenate([trans, np.zeros((n_rays, 1), F32)], 1)).astype(F32)
        cdf = np.maximum.accumulate(np.clip(cdf, 0, 1), axis=1).astype(F32)


# --------------------------------------------------------------------------------------
# a6  the renderer  (modules/scene/nerf_renderer.py:112-209)
# --------------------------------------------------------------------------------------
def occ_render(o, d, geo_params, app_params, binaries, aabb, training, t0=None,
               bg_color=None, dist_noise=None, near=0.0, far=1.5, step=5e-4,
               early_stop_eps=1e-4, quant=None, geo_grad=True, app_grad=False, max_steps=None,
               kept_counts=None, return_pre=False, lattice=None):
    """NeRFOCCRenderer.render restated on the oracle's operators.  o,d torch [R,3].
    bg_color [R,3] and dist_noise [R,1] are the torch.rand draws of :185,:193 (caller
    supplies them so both sides see the same numbers).
    kept_counts (int [R], tests): keep exactly that many leading samples of every ray instead of applying the
    early-stop threshold -- a sample whose scan value sits within rounding of the threshold is kept by one
    implementation and dropped by the other; the test first checks (with return_pre) that every disagreement is such
    a sample, then compares everything else on the SAME sample set.
    return_pre: add 'pre' = the marched samples before compaction with their canonical exclusive sums."""
    gs, as_ = geo_spec(), app_spec()
    aabb_t = torch.as_tensor(aabb, dtype=torch.float32)
    R = o.shape[0]
    ri, ts, te, packed = occ_march(o.detach().numpy(), d.detach().numpy(), binaries, np.asarray(aabb, F32),
                                   near, far, step, t0, max_steps, lattice=lattice)
    def positions(ri_t, ts_t, te_t):
        return o[ri_t] + d[ri_t] * ((ts_t + te_t)[:, None] / 2.0)
    ri_t = torch.from_numpy(ri); ts_t = torch.from_numpy(ts); te_t = torch.from_numpy(te)
    pre = None
    if early_stop_eps > 0 and ri.size:
        with torch.no_grad():
            sig0 = query_density(positions(ri_t, ts_t, te_t), geo_params, gs, aabb_t, quant)[:, 0]
        keep, ex = visibility_keep_mask(sig0.numpy(), ts, te, packed, early_stop_eps)
        if return_pre:
            pre = {'ray_indices': ri, 't_starts': ts, 't_ends': te, 'packed_info': packed, 'exsum': ex, 'keep': keep,
                   'threshold': F32(-math.log(early_stop_eps)), 'sigmas': sig0.numpy()}
        if kept_counts is not None:
            pos_in_ray = np.arange(ri.size) - packed[ri, 0]
            keep = pos_in_ray < np.asarray(kept_counts)[ri]
        ri, ts, te = ri[keep], ts[keep], te[keep]
        packed = packed_info_from_ray_indices(ri, R)
        ri_t = torch.from_numpy(ri); ts_t = torch.from_numpy(ts); te_t = torch.from_numpy(te)
    if ri.size == 0:
        return {'is_valid': False, 'rgb': torch.zeros(R, 3), 'distance': torch.zeros(R, 1),
                'opacities': torch.zeros(R, 1), 'pre': pre}
    pos = positions(ri_t, ts_t, te_t)
    with torch.set_grad_enabled(geo_grad and torch.is_grad_enabled()):
        sig = query_density(pos, geo_params, gs, aabb_t, quant)[:, 0]
    weights, trans, alphas = render_weight_from_density(ts_t, te_t, sig, packed)
    opac = accumulate_along_rays(weights, None, ri_t, R)
    tmid = ((ts_t + te_t) / 2.0)[:, None]
    dist = accumulate_along_rays(weights, tmid, ri_t, R)
    with torch.set_grad_enabled(app_grad and torch.is_grad_enabled()):
        rgbs = query_rgb(pos, app_params, as_, aabb_t, quant)
    col = accumulate_along_rays(weights.detach(), rgbs, ri_t, R)
    if training:
        dist = torch.relu(dist + (dist_noise * 2. - 1.) * (1. - opac))
        col = col + bg_color * (1. - opac).detach()
    else:
        dist = dist + 5. * (1. - opac).detach()
        col = col + .5 * (1. - opac).detach()
    return {'is_valid': True, 'rgb': col, 'distance': dist, 'weights': weights, 'opacities': opac,
            'trans': trans, 't_starts': ts_t, 't_ends': te_t, 'ray_indices': ri_t,
            'packed_info': packed, 'sigmas': sig, 'rgbs': rgbs, 'pre': pre}


# --------------------------------------------------------------------------------------
# a9  training arithmetic  (modules/scene/nerf.py:186-311)
# --------------------------------------------------------------------------------------
def lr_schedule(progress, init_lr, peak_lr, peak_at, lr_alpha):
    """NeRFScene.update_lr (nerf.py:300-311): linear warm-up to peak_at, then cosine to lr_alpha*peak."""
    if progress < peak_at:
        lp = progress / peak_at
        return peak_lr * lp + init_lr * (1. - lp)
    lp = (progress - peak_at) / (1. - peak_at)
    return peak_lr * ((np.cos(lp * np.pi) + 1.) * .5 * (1. - lr_alpha) + lr_alpha)


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update (the optimiser of nerf.py:171,180), step>=1.
    Returns new (p, m, v)."""
    m = m * beta1 + g * (1 - beta1)
    v = v * beta2 + g * g * (1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def geo_step_loss(out, gt_dist, progress, depth_w=1.0, dist_w=0.1, loss_scale=128.0):
    """Loss of train_one_step_geo (nerf.py:208-252): smooth-L1(beta=1e-2, mean) on distance +
    dist_w*min(2*progress,1)*distortion; multiplied by the GradScaler's 2^7 (never unscaled)."""
    dl = torch.nn.functional.smooth_l1_loss(out['distance'], gt_dist, beta=1e-2, reduction='mean')
    mid = (out['t_ends'] + out['t_starts']) * .5
    sec = out['t_ends'] - out['t_starts']
    distl = flatten_eff_distloss(out['weights'], mid, sec, out['ray_indices'])
    ratio = min(progress * 2., 1.)
    loss = dl * depth_w + distl * dist_w * ratio
    return loss * loss_scale, dl, distl


def app_step_loss(out, gt_rgb, color_w=1.0, loss_scale=128.0):
    """Loss of train_one_step_app (nerf.py:281-293): smooth-L1(beta=5e-2, mean) on colour."""
    cl = torch.nn.functional.smooth_l1_loss(out['rgb'], gt_rgb, beta=5e-2, reduction='mean')
    return cl * color_w * loss_scale, cl


# --------------------------------------------------------------------------------------
# a10  supervision pool pieces  (modules/dataset/sup_info.py:236-259, 304-330)
# --------------------------------------------------------------------------------------
def gen_occ_grid(o: torch.Tensor, d: torch.Tensor, dist: torch.Tensor, res: int):
    """SupInfoPool.gen_occ_grid: splat o+d*dist with the 27 offsets in {-1/res,0,1/res}^3;
    index x*res^2+y*res+z of int64((clip(p,+-.999)*.5+.5)*res).  Returns uint8 [res^3]."""
    pts = o + d * dist.squeeze()[..., None]
    occ = torch.zeros(res ** 3, dtype=torch.uint8)
    shift = 1. / res
    lin = torch.linspace(-shift, shift, 3)
    for sx in lin:
        for sy in lin:
            for sz in lin:
                sh = torch.stack([sx, sy, sz])[None, :] + pts
                c = ((sh.clip(-0.999, 0.999) * .5 + .5) * res).to(torch.int64)
                occ[c[..., 0] * res * res + c[..., 1] * res + c[..., 2]] = 1
    return occ


def rand_ray_indices(n_pool: int, batch: int, generator=None):
    """rand_ray_color_data's index draw (sup_info.py:256): torch.randint(0, n, (batch,))."""
    return torch.randint(0, n_pool, (batch,), generator=generator)


# --------------------------------------------------------------------------------------
# synthetic scene of SURVEY.md section 8(d) (the kitchen panorama is absent)
# --------------------------------------------------------------------------------------
def synthetic_room(d: torch.Tensor, half=(0.9, 0.7, 0.5)):
    """Axis-aligned box room seen from the origin.  d [...,3] unit rays ->
    (distance [...,1] normalised by max*1.05 as dataset.py:97-101, rgb [...,3])."""
    h = torch.tensor(half, dtype=torch.float32)
    t = (h / d.abs().clamp_min(1e-12))
    dist, axis = t.min(-1)
    p = d * dist[..., None]
    uv_idx = torch.tensor([[1, 2], [0, 2], [0, 1]])[axis]
    u = torch.gather(p, -1, uv_idx[..., :1])[..., 0]
    v = torch.gather(p, -1, uv_idx[..., 1:])[..., 0]
    k = torch.tensor([8., 16., 32.])[axis]
    base = 0.5 + 0.5 * torch.sin(k * u) * torch.sin(k * v)
    tint = torch.tensor([[1.0, 0.6, 0.4], [0.4, 1.0, 0.6], [0.5, 0.6, 1.0]])[axis]
    sgn = torch.gather(torch.sign(d), -1, axis[..., None])[..., 0]
    rgb = (base[..., None] * tint) * (0.75 + 0.25 * sgn[..., None])
    scale = dist.max() * 1.05
    return (dist / scale)[..., None], rgb.clamp(0, 1)
