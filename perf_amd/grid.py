"""Host-side geometry of the multiresolution hash grid and the 64-wide MLP (product code).

Mirrors the configuration dictionaries PeRF hands to tinycudann
(modules/fields/ngp_nerf.py:96-134,230-245; modules/geo_predictors/pano_joint_predictor.py:30-41):
per level  scale = N_min * b^l - 1 (fp32),  res = ceil(scale) + 1,
size = min(align8(res^3), 2^T), offsets = running sum.
"""
import math
from dataclasses import dataclass, field

import numpy as np

from . import _lib

_F = np.float32
LOCAL_MIN_RES = 64          # line_local grids: levels of at least this resolution are stored line-local (default)
# default super-block shapes (log2 vertices along x, y, z; 2 MiB each), by measurement on BASELINE config 5's panorama: z-deep for the rays
# near the poles; with overlapping runs x-wide as well (the last cell of a super-block row still needs a second request: 1 cell in 96)
SB_SHIFT = {'tcnn': (5, 6, 8), 'line_local': (5, 6, 8), 'line_overlap': (7, 5, 7)}


@dataclass
class GridConfig:
    n_levels: int = 16
    n_features_per_level: int = 2
    log2_hashmap_size: int = 18
    base_resolution: int = 16
    per_level_scale: float = 1.4472692012786865
    interpolation: str = 'Linear'
    # Table layout.  'tcnn': tiny-cuda-nn's (dense x + y res + z res^2, or the prime-XOR hash of the vertex) -- the only layout of
    # every grid the reference defines.  'line_local' (opt-in, inference only; BASELINE config 5's L = 20 tables sized to HBM, for
    # which no reference result exists): levels with res >= local_min_res store a 4 x 4 x 2 block of vertices as one 128-byte
    # line, the blocks of a 2^sb_shift-vertex super-block (default 32 x 64 x 256 = 2 MiB) contiguously, and hash (or densely
    # index) the SUPER-BLOCK: a sample's eight corners lie in ~2.3 lines of one page.  'line_overlap': line_local whose 16-byte x
    # runs overlap by one vertex (a cell's x corner pair never straddles two runs: ~1.9 lines per sample and level, one request
    # per (y, z) corner pair; a super-block row holds 3/4 as many cells; see canonicalize_).  include/perf_hip.h PERF_LAYOUT_*.
    layout: str = 'tcnn'
    sb_shift: tuple = None              # default: SB_SHIFT[layout]
    local_min_res: int = LOCAL_MIN_RES
    scale: np.ndarray = field(init=False, repr=False)
    res: np.ndarray = field(init=False, repr=False)
    size: np.ndarray = field(init=False, repr=False)
    offset: np.ndarray = field(init=False, repr=False)
    hashed: np.ndarray = field(init=False, repr=False)
    local: np.ndarray = field(init=False, repr=False)
    nsx: np.ndarray = field(init=False, repr=False)
    nsxy: np.ndarray = field(init=False, repr=False)
    total: int = field(init=False)

    def __post_init__(self):
        if self.n_features_per_level != 2:
            raise ValueError('only n_features_per_level == 2 is supported by the gfx950 kernels')
        if not (1 <= self.n_levels <= _lib.MAX_LEVELS):
            raise ValueError(f'n_levels must be in [1, {_lib.MAX_LEVELS}]')
        if self.interpolation not in ('Linear', 'Smoothstep'):
            raise ValueError(f'unsupported interpolation {self.interpolation!r}')
        L = self.n_levels
        log2_b = _F(np.log2(_F(self.per_level_scale)))
        self.scale = np.zeros(L, _F)
        self.res = np.zeros(L, np.uint32)
        self.size = np.zeros(L, np.uint32)
        self.offset = np.zeros(L, np.uint64)           # 64-bit: tables beyond 2^32 entries (tcnn's own offsets are uint32)
        self.hashed = np.zeros(L, np.uint32)
        self.local = np.zeros(L, np.uint32)
        self.nsx = np.zeros(L, np.uint32)
        self.nsxy = np.zeros(L, np.uint32)
        if self.layout not in ('tcnn', 'line_local', 'line_overlap'):
            raise ValueError(f'unsupported table layout {self.layout!r}')
        self.sb_shift = tuple(int(v) for v in (self.sb_shift if self.sb_shift is not None else SB_SHIFT[self.layout]))
        if len(self.sb_shift) != 3 or self.sb_shift[0] < 2 or self.sb_shift[1] < 2 or self.sb_shift[2] < 1 or sum(self.sb_shift) > 24:
            raise ValueError(f'sb_shift {self.sb_shift}: a super-block holds at least one 4 x 4 x 2 block and at most 2^24 entries')
        per_sb = 1 << sum(self.sb_shift)
        total = 0
        for l in range(L):
            growth = _F(np.exp2(np.float64(_F(l) * log2_b)))
            s = _F(_F(growth * _F(self.base_resolution)) - _F(1.0))
            r = int(math.ceil(float(s))) + 1
            if self.layout != 'tcnn' and r >= self.local_min_res:
                nd = [(r + (1 << sh)) >> sh for sh in self.sb_shift]          # super-blocks per dimension (vertices 0..res)
                if self.layout == 'line_overlap':
                    # x runs overlap by one vertex: cell gx sits at storage coordinate gx + gx // 3 (a super-block row holds 3/4 as many cells);
                    # the last cell (res - 1) may take its second corner from storage coordinate + 2
                    nd[0] = (((r - 1) + (r - 1) // 3 + 2) >> self.sb_shift[0]) + 1
                cells = nd[0] * nd[1] * nd[2] * per_sb
                n = min(cells, 1 << self.log2_hashmap_size)
                if n < per_sb:
                    raise ValueError(f'{self.layout}: 2^{self.log2_hashmap_size} entries hold no {self.sb_shift} super-block')
                self.local[l], self.nsx[l], self.nsxy[l] = 1, nd[0], nd[0] * nd[1]
                total = -(-total // per_sb) * per_sb         # a line-local level starts on a super-block boundary: its 128-byte blocks are cache lines
            else:
                cells = r ** 3
                n = min(cells, 0xFFFFFFFF // 2)
                n = -(-n // 8) * 8
                n = min(n, 1 << self.log2_hashmap_size)
            self.scale[l], self.res[l], self.size[l], self.offset[l] = s, r, n, total
            self.hashed[l] = 1 if cells > n else 0
            total += n
        self.total = total

    @classmethod
    def from_tcnn(cls, cfg: dict) -> 'GridConfig':
        otype = cfg.get('otype', 'HashGrid')
        if otype not in ('HashGrid', 'Grid'):
            raise ValueError(f'unsupported encoding otype {otype!r}')
        return cls(n_levels=int(cfg.get('n_levels', 16)),
                   n_features_per_level=int(cfg.get('n_features_per_level', 2)),
                   log2_hashmap_size=int(cfg.get('log2_hashmap_size', 19)),
                   base_resolution=int(cfg.get('base_resolution', 16)),
                   per_level_scale=float(cfg.get('per_level_scale', 2.0)),
                   interpolation=str(cfg.get('interpolation', 'Linear')))

    @property
    def n_params(self) -> int:
        return self.total * 2

    @property
    def n_output_dims(self) -> int:
        return self.n_levels * 2

    def canonicalize_(self, table):
        """layout='line_overlap': make `table` (a tensor of total * 2 features, any dtype / device, modified in place) a VALID table --
        position 3 of every 16-byte x run := position 0 of the next run of the same super-block row (the two entries hold ONE vertex;
        oracle/perf_oracle.py:canonical_overlap_fill).  Whoever writes such a table calls this afterwards; other layouts: no-op."""
        if self.layout != 'line_overlap':
            return table
        runs = 1 << (self.sb_shift[0] - 2)
        flat = table.view(-1)
        for l in range(self.n_levels):
            if not self.local[l]:
                continue
            lo, n = int(self.offset[l]) * 2, int(self.size[l]) * 2
            v = flat[lo:lo + n].view(n // (64 * runs), runs, 8, 4, 2)     # [row of blocks, block along x, (y, z) in block, x in run, feature]
            v[:, :-1, :, 3].copy_(v[:, 1:, :, 0].clone())
        return table

    def desc(self) -> '_lib.GridDesc':
        """The C-ABI descriptor (perf_grid_desc).  Built once per configuration: filling the ctypes arrays costs ~30 us of host
        time, and an eager step passes it to half a dozen entry points (the library only reads it)."""
        key = (self.n_levels, self.interpolation, self.log2_hashmap_size, self.base_resolution, self.per_level_scale, self.layout, self.sb_shift, self.local_min_res)
        cached = self.__dict__.get('_desc')
        if cached is not None and cached[0] == key:
            return cached[1]
        d = _lib.GridDesc()
        d.n_levels = self.n_levels
        d.interpolation = _lib.INTERP_SMOOTHSTEP if self.interpolation == 'Smoothstep' else _lib.INTERP_LINEAR
        for l in range(self.n_levels):
            d.scale[l] = float(self.scale[l]); d.res[l] = int(self.res[l]); d.size[l] = int(self.size[l])
            d.offset[l] = int(self.offset[l]); d.hashed[l] = int(self.hashed[l])
            d.local[l] = int(self.local[l]); d.nsx[l] = int(self.nsx[l]); d.nsxy[l] = int(self.nsxy[l])
        d.layout = {'tcnn': _lib.LAYOUT_TCNN, 'line_local': _lib.LAYOUT_LINE_LOCAL, 'line_overlap': _lib.LAYOUT_LINE_OVERLAP}[self.layout]
        for k in range(3):
            d.sb_shift[k] = self.sb_shift[k]
        self.__dict__['_desc'] = (key, d)
        return d


_ACTS = {'None': _lib.ACT_NONE, 'Sigmoid': _lib.ACT_SIGMOID, 'Exponential': _lib.ACT_EXP}


@dataclass
class MlpConfig:
    n_levels: int                 # inputs = 2 * n_levels
    n_hidden_layers: int = 1
    n_output_dims: int = 1
    output_activation: str = 'None'
    exp_shift: float = 0.0
    n_neurons: int = 64

    def __post_init__(self):
        if self.n_neurons != 64:
            raise ValueError('only n_neurons == 64 is supported by the gfx950 MFMA kernels')
        if self.n_hidden_layers not in (1, 2):
            raise ValueError('n_hidden_layers must be 1 or 2')
        if not (1 <= self.n_output_dims <= 16):
            raise ValueError('n_output_dims must be in [1, 16]')
        if self.output_activation not in _ACTS:
            raise ValueError(f'unsupported output activation {self.output_activation!r}')
        if not (1 <= self.n_levels <= _lib.MAX_LEVELS):
            raise ValueError(f'n_levels must be in [1, {_lib.MAX_LEVELS}]')

    @property
    def n_in_pad(self) -> int:
        return 16 * (-(-self.n_levels // 8))         # inputs padded to the MFMA k-step: 16, 32, or 48 (inference only)

    @property
    def shapes(self):
        return [(64, self.n_in_pad)] + [(64, 64)] * (self.n_hidden_layers - 1) + [(16, 64)]

    @property
    def n_params(self) -> int:
        return sum(o * i for o, i in self.shapes)

    def desc(self) -> '_lib.MlpDesc':
        d = _lib.MlpDesc()
        d.n_levels = self.n_levels
        d.n_hidden_layers = self.n_hidden_layers
        d.n_out = self.n_output_dims
        d.out_act = _ACTS[self.output_activation]
        d.exp_shift = self.exp_shift
        return d
