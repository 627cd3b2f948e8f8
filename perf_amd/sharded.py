"""BASELINE config 5 on several GPUs: hash tables cut by LEVEL over the ranks (last row).

A field whose tables are sized to HBM (L = 20 levels up to resolution 8192, log2_hashmap_size 28-31: 9-57 GiB per
encoder in 16 bits, 7x that with fp32 master + Adam state) is replicated for inference when it fits and sharded when it
has to be trained.  The shard axis is the level: a level's table lives on exactly one rank, so

  * forward: every rank all-gathers the batch's sample positions (12 B/sample), encodes ALL samples at ITS levels with the
    same gfx950 kernel (its descriptor simply lists fewer levels), and ONE all-to-all returns to every rank the features
    of its own samples at all levels (4 B per sample and level); the 64-wide MLP then runs locally on replicated weights;
  * backward: the transpose -- one all-to-all of the feature gradients, then the local grid backward; table gradients,
    fp32 masters and Adam state never leave their rank (no gradient all-reduce for 99.9 % of the parameters).

Per sample and pass a rank moves (W-1)/W x (12 + 4 L) bytes over xGMI: 80 B at L = 20, i.e. ~56 GB/s per GPU at the
7e8 ray-samples/s a single GPU reaches on such tables -- an eighth of the 7 x 64 GB/s a GPU can inject.  The exchange is
an all-to-all, not a ring: every pair of GPUs has its own xGMI link, so all 7 links carry 1/7 of the traffic each.

Levels are assigned greedily by table size (largest first to the least loaded rank) so that the shards are balanced in
bytes AND in gather work (every level costs one 8-corner gather per sample whatever its size).
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from . import _lib, ops
from .grid import GridConfig


@dataclass
class GridSlice:
    """The levels `levels` of `full` as a grid of their own (what one rank holds): same per-level geometry, offsets
    relative to the local table.  Duck-types GridConfig for perf_amd.ops (n_levels, n_params, total, desc())."""
    full: GridConfig
    levels: List[int]
    offset: np.ndarray = field(init=False, repr=False)
    total: int = field(init=False)

    def __post_init__(self):
        per_sb = 1 << sum(int(v) for v in self.full.sb_shift)
        offs, total = [], 0
        for l in self.levels:
            if int(self.full.local[l]):
                total = -(-total // per_sb) * per_sb             # (GridConfig's rule: a line-local level starts on a super-block boundary)
            offs.append(total)
            total += int(self.full.size[l])
        self.offset = np.asarray(offs, np.uint64)
        self.total = int(total)

    @property
    def n_levels(self):
        return len(self.levels)

    @property
    def n_params(self):
        return self.total * 2

    def pack(self, level_tables, dtype, device='cpu'):
        """The slice's table from its levels' tables (flat, 2 features per entry, in slice order): each at its offset, zeros between
        (the padding in front of a line-local level)."""
        out = torch.zeros(self.total * 2, dtype=dtype, device=device)
        for k, t in enumerate(level_tables):
            lo = 2 * int(self.offset[k])
            out[lo: lo + t.numel()] = t.to(device=device, dtype=dtype)
        return out

    @property
    def interpolation(self):
        return self.full.interpolation

    def desc(self):
        if getattr(self, '_desc', None) is not None:
            return self._desc
        d = _lib.GridDesc()
        d.n_levels = len(self.levels)
        d.interpolation = _lib.INTERP_SMOOTHSTEP if self.full.interpolation == 'Smoothstep' else _lib.INTERP_LINEAR
        for k, l in enumerate(self.levels):
            d.scale[k] = float(self.full.scale[l]); d.res[k] = int(self.full.res[l]); d.size[k] = int(self.full.size[l])
            d.offset[k] = int(self.offset[k]); d.hashed[k] = int(self.full.hashed[l])
            d.local[k] = int(self.full.local[l]); d.nsx[k] = int(self.full.nsx[l]); d.nsxy[k] = int(self.full.nsxy[l])
        d.layout = {'tcnn': _lib.LAYOUT_TCNN, 'line_local': _lib.LAYOUT_LINE_LOCAL, 'line_overlap': _lib.LAYOUT_LINE_OVERLAP}[self.full.layout]   # (a level keeps its layout on its rank)
        for k in range(3):
            d.sb_shift[k] = int(self.full.sb_shift[k])
        self._desc = d
        return d


def assign_levels(grid: GridConfig, world: int) -> List[List[int]]:
    """Levels per rank: largest table first to the rank that holds the fewest levels, ties by bytes (deterministic)."""
    order = sorted(range(grid.n_levels), key=lambda l: (-int(grid.size[l]), l))
    held = [[] for _ in range(world)]
    load = [0] * world
    for l in order:
        r = min(range(world), key=lambda q: (len(held[q]), load[q], q))
        held[r].append(l); load[r] += int(grid.size[l])
    return [sorted(h) for h in held]


def exchange_volume(grid: GridConfig, world: int, n_per_rank: int, feat_bytes: int = 2, training: bool = False):
    """Bytes a rank moves over xGMI for ONE encode of n_per_rank samples per rank through the level-sharded grid (and, with
    training, for the transposed exchange of the fp32 feature gradients): per rank and PER LINK -- the exchange is an
    all-gather of positions plus an all-to-all of features, every pair of GPUs has its own link, so a rank's traffic to
    peer q uses link (rank, q) only.  -> dict(levels_per_rank, table_GiB_per_rank (16-bit), per_link_bytes_out [W][W],
    per_rank_bytes_out, per_rank_bytes_in, worst_link_bytes)."""
    assign = assign_levels(grid, world)
    rows = [len(a) for a in assign]
    link = [[0] * world for _ in range(world)]                # link[s][q]: bytes s sends to q
    for s_ in range(world):
        for q in range(world):
            if q == s_:
                continue
            b = 12 * n_per_rank                               # positions of s's samples (all-gather)
            b += rows[s_] * n_per_rank * 2 * feat_bytes       # features of q's samples at s's levels (all-to-all)
            if training:
                b += rows[q] * n_per_rank * 2 * 4             # fp32 gradients of s's samples at q's levels (transposed all-to-all)
            link[s_][q] = b
    out = [sum(r) for r in link]
    inn = [sum(link[s_][q] for s_ in range(world)) for q in range(world)]
    return {'levels_per_rank': assign, 'table_GiB_per_rank': [round(sum(int(grid.size[l]) for l in a) * 2 * feat_bytes / 2 ** 30, 3) for a in assign],
            'per_link_bytes_out': link, 'per_rank_bytes_out': out, 'per_rank_bytes_in': inn,
            'worst_link_bytes': max(max(r) for r in link) if world > 1 else 0}


def _group():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


class LevelShardedEncoder:
    """One rank's share of a level-sharded hash grid (16-bit table for encoding; `table32` when it is trained)."""

    def __init__(self, grid: GridConfig, dtype='fp16', table16=None, seed=1337):
        self.dist, self.rank, self.world = _group()
        self.grid = grid
        self.assignment = assign_levels(grid, self.world)
        self.local = GridSlice(grid, self.assignment[self.rank])
        self.dtype = ops.torch_dtype(dtype)
        dev = torch.device('cuda', torch.cuda.current_device())
        if table16 is None:
            # every rank draws the FULL table's random stream level by level and keeps its own levels: the union equals the
            # unsharded initialisation (tests compare against it); huge tables would be initialised shard-locally instead
            g = torch.Generator(device='cpu').manual_seed(seed)
            parts = []
            for l in range(grid.n_levels):
                t = (torch.rand(int(grid.size[l]) * 2, generator=g, device='cpu') * 2 - 1) * 1e-4
                if l in self.local.levels:
                    parts.append(t)
            table16 = self.local.pack(parts, self.dtype)
        self.table16 = table16.to(dev)

    # ---- collectives (all-to-all on RCCL; gloo -- tests on one GPU -- has none, so it is composed from all_gather) ------
    def _all_gather(self, t):
        if self.dist is None or self.world == 1:
            return t
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1)) if self.dist.get_backend() == 'nccl' else \
            self.dist.all_gather(list(out.unbind(0)), t.contiguous())
        return out

    def _exchange(self, send, recv_rows):
        """send [W, rows_mine, n, C]: block q goes to rank q.  Returns a list over source ranks s of [rows_s, n, C]."""
        W = self.world
        n, C = send.shape[2], send.shape[3]
        if self.dist.get_backend() == 'nccl':
            out = torch.empty(sum(recv_rows) * n * C, dtype=send.dtype, device=send.device)
            self.dist.all_to_all_single(out, send.contiguous().view(-1), output_split_sizes=[r * n * C for r in recv_rows],
                                        input_split_sizes=[send.shape[1] * n * C] * W)
            return [p.view(r, n, C) for p, r in zip(out.split([r * n * C for r in recv_rows]), recv_rows)]
        # gloo: everybody gathers everything (padded to the largest block) and keeps what was addressed to it
        rmax = max(recv_rows)
        pad = torch.zeros(W, rmax, n, C, dtype=send.dtype, device=send.device)
        pad[:, :send.shape[1]] = send
        everything = [torch.empty_like(pad) for _ in range(W)]
        self.dist.all_gather(everything, pad)
        return [everything[s][self.rank, :recv_rows[s]] for s in range(W)]

    # ---- forward / backward -----------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x01):
        """x01 [n,3] (this rank's samples; the same n on every rank) -> feat [L, n, 2] 16-bit, level major, global order."""
        n = x01.shape[0]
        if self.world == 1:
            return ops.hashgrid_fwd(self.local, x01, self.table16)
        x_all = self._all_gather(x01).view(-1, 3)                                   # [W*n, 3]
        f = ops.hashgrid_fwd(self.local, x_all, self.table16) if self.local.n_levels else \
            torch.zeros(0, x_all.shape[0], 2, dtype=self.dtype, device=x01.device)
        send = f.view(self.local.n_levels, self.world, n, 2).permute(1, 0, 2, 3)     # block q = my levels at q's samples
        rows = [len(a) for a in self.assignment]
        parts = self._exchange(send, rows)
        feat = torch.empty(self.grid.n_levels, n, 2, dtype=self.dtype, device=x01.device)
        for s, p in enumerate(parts):
            if rows[s]:
                feat[torch.tensor(self.assignment[s], device=x01.device)] = p
        self._x_all = x_all                                                          # kept for table_gradient()
        return feat

    @torch.no_grad()
    def table_gradient(self, dfeat, level_absmax=None, out=None):
        """dfeat [L, n, 2] f32 (this rank's samples, global level order) -> fp32 gradient of THIS rank's table
        [local.n_params]; positions are the ones of the preceding encode()."""
        n = dfeat.shape[1]
        if self.world == 1:
            raise RuntimeError('single rank: use ops.hashgrid_bwd')
        rows = [len(a) for a in self.assignment]
        dev = dfeat.device
        rmax = max(rows)
        send = torch.zeros(self.world, rmax, n, 2, dtype=torch.float32, device=dev)   # block q = dfeat at q's levels
        for q in range(self.world):
            if rows[q]:
                send[q, :rows[q]] = dfeat[torch.tensor(self.assignment[q], device=dev)]
        mine = self.local.n_levels
        parts = self._exchange(send, [rmax] * self.world)
        d_all = torch.stack([p[:mine] for p in parts], 1).reshape(mine, self.world * n, 2).contiguous()   # [L_r, W*n, 2]
        if mine == 0:
            return torch.zeros(0, dtype=torch.float32, device=dev)
        return ops.hashgrid_bwd(self.local, self._x_all, d_all, out=out)


class LevelShardedNeRF:
    """Inference-side counterpart of NGPNeRF (modules/fields/ngp_nerf.py:68-176) for tables cut by level over the ranks
    (BASELINE config 5 on several GPUs): the duck type NeRFOCCRenderer uses -- density_at / rgb_at on kernel-made sample
    positions -- with the two encodes going through LevelShardedEncoder (positions all-gathered, ONE all-to-all of features
    per field) and the 64-wide MLPs running locally on replicated weights.  Every rank must render batches of the same
    capacity (sync-free mode: renderer.sample_capacity), so that the collectives match.

    Built from an unsharded NGPNeRF whose parameters every rank can afford to hold once (tests), or from per-rank tables."""

    def __init__(self, nerf, dtype=None):
        self.training = False
        self._aabb_host = nerf._aabb_host
        self.aabb = nerf.aabb
        self.dtype_name = dtype or (nerf.dtype_name if hasattr(nerf, 'nets') else nerf.geo_mlp.dtype_name)
        self.nets = {}
        for name in ('geo_mlp', 'app_mlp'):
            if hasattr(nerf, 'nets'):              # fields.InferenceNeRF: 16-bit tables only, either table layout
                mlp, w16 = nerf.nets[name]
                grid = nerf.grid
            else:
                net = getattr(nerf, name)
                mlp, w16, grid = net.mlp, net.working_copy(), net.grid
            n_net = mlp.n_params
            dist, rank, world = _group()
            levels = assign_levels(grid, world)[rank]
            local = GridSlice(grid, levels)
            mine = local.pack([w16[n_net + 2 * int(grid.offset[l]): n_net + 2 * int(grid.offset[l] + grid.size[l])] for l in levels], w16.dtype, w16.device)
            enc = LevelShardedEncoder(grid, dtype=self.dtype_name, table16=mine)
            assert enc.local.levels == local.levels
            self.nets[name] = (enc, mlp, w16[:n_net].clone())

    def eval(self):
        self.training = False
        return self

    def _field(self, name, x01, sel, n_dev):
        enc, mlp, w_net = self.nets[name]
        feat = enc.encode(x01)                      # every row of the capacity-sized batch (rows beyond n_dev hold garbage)
        return ops.mlp_fwd(mlp, w_net, feat, sel, n_dev=n_dev)

    @torch.no_grad()
    def density_at(self, x01, sel, n_dev=None):
        return self._field('geo_mlp', x01, sel, n_dev)[:, 0]

    @torch.no_grad()
    def rgb_at(self, x01, sel, n_dev=None):
        return self._field('app_mlp', x01, sel, n_dev)

    def sample_points(self, rays_o, rays_d, ray_indices, t_starts, t_ends):
        return ops.points_from_rays(rays_o, rays_d, ray_indices, t_starts, t_ends, self._aabb_host)
