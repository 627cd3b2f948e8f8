// Device-side geometry of the multiresolution hash grid shared by the encode kernels (hashgrid.hip) and the fused
// encode + MLP kernel (mlp.hip): level table, grid position, corner indices, interpolation weights.
#pragma once
#include "common.hpp"

namespace perf {

struct GridParams {
    int32_t n_levels;
    int32_t interpolation;
    float scale[PERF_MAX_LEVELS];
    uint32_t res[PERF_MAX_LEVELS];
    uint32_t size[PERF_MAX_LEVELS];
    uint64_t offset[PERF_MAX_LEVELS];
    uint32_t hashed[PERF_MAX_LEVELS];
};

// the line-local part of a descriptor (PERF_LAYOUT_LINE_LOCAL), handed to the kernels that understand it beside GridParams
struct GridLocal {
    uint32_t any;                       // 1: some level is line-local
    uint32_t ovl;                       // 1: PERF_LAYOUT_LINE_OVERLAP (the x runs of a line-local level overlap by one vertex)
    uint32_t shx, shy, shz;
    uint32_t local[PERF_MAX_LEVELS];
    uint32_t nsx[PERF_MAX_LEVELS];
    uint32_t nsxy[PERF_MAX_LEVELS];
};

// loc == nullptr: the caller only understands tcnn's layout (gradients, second order, the fused encode + MLP kernel, the
// 16-level forward kernel): a line-local descriptor is refused
static int fill_params(const perf_grid_desc* g, GridParams* p, GridLocal* loc = nullptr) {
    PERF_REQUIRE(g != nullptr, "grid desc is NULL");
    PERF_REQUIRE(g->n_levels >= 1 && g->n_levels <= PERF_MAX_LEVELS, "n_levels %d out of range", g->n_levels);
    PERF_REQUIRE(g->layout == PERF_LAYOUT_TCNN || g->layout == PERF_LAYOUT_LINE_LOCAL || g->layout == PERF_LAYOUT_LINE_OVERLAP, "unknown table layout %d", (int)g->layout);
    PERF_REQUIRE(g->layout == PERF_LAYOUT_TCNN || loc != nullptr,
                 "this entry point takes tcnn-layout grids only (PERF_LAYOUT_LINE_LOCAL / _OVERLAP are inference only: perf_hashgrid_fwd, perf_hashgrid_corners, perf_field_infer)");
    if (loc) {
        loc->any = 0; loc->ovl = 0; loc->shx = loc->shy = 2; loc->shz = 1;
        for (int l = 0; l < PERF_MAX_LEVELS; ++l) loc->local[l] = loc->nsx[l] = loc->nsxy[l] = 0;
        if (g->layout != PERF_LAYOUT_TCNN) {
            loc->ovl = g->layout == PERF_LAYOUT_LINE_OVERLAP ? 1u : 0u;
            loc->shx = g->sb_shift[0]; loc->shy = g->sb_shift[1]; loc->shz = g->sb_shift[2];
            PERF_REQUIRE(loc->shx >= 2 && loc->shy >= 2 && loc->shz >= 1 && loc->shx + loc->shy + loc->shz <= 24, "bad super-block shape");
            const uint32_t per_sb = 1u << (loc->shx + loc->shy + loc->shz);
            for (int l = 0; l < g->n_levels; ++l) {
                loc->local[l] = g->local[l] ? 1u : 0u; loc->nsx[l] = g->nsx[l]; loc->nsxy[l] = g->nsxy[l];
                if (!g->local[l]) continue;
                loc->any = 1;
                PERF_REQUIRE(g->size[l] >= per_sb && g->size[l] % per_sb == 0, "line-local level %d: size %u is not a whole number of super-blocks", l, g->size[l]);
                PERF_REQUIRE(g->offset[l] % 32u == 0, "line-local level %d starts inside a 128-byte line (offset %llu entries)", l, (unsigned long long)g->offset[l]);
                if (g->hashed[l]) { const uint32_t ns = g->size[l] / per_sb; PERF_REQUIRE((ns & (ns - 1)) == 0, "line-local hashed level %d: %u super-blocks is not a power of two", l, ns); }
            }
        }
    }
    p->n_levels = g->n_levels;
    p->interpolation = g->interpolation;
    for (int l = 0; l < PERF_MAX_LEVELS; ++l) {
        p->scale[l] = g->scale[l]; p->res[l] = g->res[l]; p->size[l] = g->size[l];
        p->offset[l] = g->offset[l]; p->hashed[l] = g->hashed[l];
        if (l < g->n_levels) {
            PERF_REQUIRE(g->size[l] > 0, "level %d has size 0", l);
            if (g->hashed[l] && !(g->layout != PERF_LAYOUT_TCNN && g->local[l]))
                PERF_REQUIRE((g->size[l] & (g->size[l] - 1)) == 0, "hashed level %d size %u is not a power of two", l, g->size[l]);
        }
    }
    return PERF_OK;
}

constexpr uint32_t kPrimeY = 2654435761u;
constexpr uint32_t kPrimeZ = 805459861u;

// Grid position of a coordinate: ONE rounding, pos = fl(x*scale + 0.5), as tiny-cuda-nn's pos_fract computes it
// (fmaf(scale, input, 0.5f)); oracle/perf_oracle.py:grid_pos restates the same rounding for numpy.
__device__ __forceinline__ float grid_pos(float x, float scale) { return __builtin_fmaf(x, scale, 0.5f); }

// Corner bookkeeping of one (sample, level): 8 table indices + fractional position.
struct Corners {
    uint32_t idx[8];
    float f[3];
    uint32_t cell[3];       // integer cell: two samples with equal cells gather the same 8 entries
};

__device__ __forceinline__ Corners corners_of(float x, float y, float z, float scale, uint32_t res,
                                              uint32_t size, bool hashed) {
    Corners c;
    float px = grid_pos(x, scale);
    float py = grid_pos(y, scale);
    float pz = grid_pos(z, scale);
    float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    c.f[0] = px - flx; c.f[1] = py - fly; c.f[2] = pz - flz;
    uint32_t gx = (uint32_t)(int32_t)flx, gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
    c.cell[0] = gx; c.cell[1] = gy; c.cell[2] = gz;
    if (hashed) {
        uint32_t hy0 = gy * kPrimeY, hy1 = hy0 + kPrimeY;
        uint32_t hz0 = gz * kPrimeZ, hz1 = hz0 + kPrimeZ;
        uint32_t m = size - 1u;
        uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
        for (int k = 0; k < 8; ++k) c.idx[k] = ((gx + (uint32_t)(k & 1)) ^ yz[k >> 1]) & m;
    } else {
        uint32_t r2 = res * res;
        uint32_t base = gx + gy * res + gz * r2;
#pragma unroll
        for (int k = 0; k < 8; ++k) c.idx[k] = base + (uint32_t)(k & 1) + ((k & 2) ? res : 0u) + ((k & 4) ? r2 : 0u);
        // (modulo the level's size: ONE test of the cell's last corner covers the eight -- it only fires at the table's last cells)
        const uint32_t last = base + 1u + res + r2;
        if (last >= size || last < base) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c.idx[k] >= size) c.idx[k] = c.idx[k] % size;
        }
    }
    return c;
}

// ---- PERF_LAYOUT_LINE_LOCAL (include/perf_hip.h): entry index of vertex (vx, vy, vz) of a line-local level ----------------------
__device__ __forceinline__ uint32_t local_vertex_index(const GridLocal& gl, int l, uint32_t size, bool hashed, uint32_t vx, uint32_t vy, uint32_t vz) {
    const uint32_t shx = gl.shx, shy = gl.shy, shz = gl.shz;
    const uint32_t sx = vx >> shx, sy = vy >> shy, sz = vz >> shz;
    const uint32_t sh = shx + shy + shz;
    const uint32_t slot = hashed ? ((sx ^ (sy * kPrimeY) ^ (sz * kPrimeZ)) & ((size >> sh) - 1u)) : (sx + sy * gl.nsx[l] + sz * gl.nsxy[l]);
    const uint32_t blk = ((vx >> 2) & ((1u << (shx - 2)) - 1u)) + (((vy >> 2) & ((1u << (shy - 2)) - 1u)) << (shx - 2))
                       + (((vz >> 1) & ((1u << (shz - 1)) - 1u)) << (shx - 2 + shy - 2));
    return (slot << sh) + (blk << 5) + (vx & 3u) + ((vy & 3u) << 2) + ((vz & 1u) << 4);
}

// ---- PERF_LAYOUT_LINE_OVERLAP: the 16-byte x runs of a line-local level overlap by one vertex.  The x corner pair of CELL gx lives
// in ONE run -- run gx / 3 of its row, positions gx % 3 and gx % 3 + 1 -- i.e. at the storage coordinates X = gx + gx / 3 and X + 1 of
// the line-local rule above; a run's last entry repeats the next run's first (the owner of the table keeps the two equal), except in
// the last run of a super-block row: its last cell takes its second corner from the next super-block's first run (storage X + 2), so
// that no vertex is ever stored in two super-blocks (hashed super-blocks could not keep such copies equal).
__device__ __forceinline__ uint32_t overlap_x(uint32_t gx) { return gx + gx / 3u; }
__device__ __forceinline__ uint32_t overlap_x1(const GridLocal& gl, uint32_t X0) {
    const uint32_t m = (1u << gl.shx) - 1u;
    return (X0 & m) == m - 1u ? X0 + 2u : X0 + 1u;
}

// corners_of for either layout
__device__ __forceinline__ Corners corners_of_any(const GridParams& gp, const GridLocal& gl, int l, float x, float y, float z) {
    if (!gl.local[l]) return corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
    Corners c;
    const float s = gp.scale[l];
    const float px = grid_pos(x, s), py = grid_pos(y, s), pz = grid_pos(z, s);
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    c.f[0] = px - flx; c.f[1] = py - fly; c.f[2] = pz - flz;
    const uint32_t gx = (uint32_t)(int32_t)flx, gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
    c.cell[0] = gx; c.cell[1] = gy; c.cell[2] = gz;
    const uint32_t X0 = gl.ovl ? overlap_x(gx) : gx, X1 = gl.ovl ? overlap_x1(gl, X0) : gx + 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.idx[k] = local_vertex_index(gl, l, gp.size[l], gp.hashed[l] != 0, (k & 1) ? X1 : X0, gy + (uint32_t)((k >> 1) & 1), gz + (uint32_t)(k >> 2));
    return c;
}

__device__ __forceinline__ void corner_weights(const float f[3], bool smooth, float w[8]) {
    float fx = f[0], fy = f[1], fz = f[2];
    if (smooth) {
        fx = fx * fx * (3.0f - 2.0f * fx);
        fy = fy * fy * (3.0f - 2.0f * fy);
        fz = fz * fz * (3.0f - 2.0f * fz);
    }
    float wx[2] = {1.0f - fx, fx}, wy[2] = {1.0f - fy, fy}, wz[2] = {1.0f - fz, fz};
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = (wx[k & 1] * wy[(k >> 1) & 1]) * wz[k >> 2];
}

// the packed 16-bit feature pair of one (sample, level): 8 gathers, fp32 interpolation, one rounding per feature -- THE
// definition both the level-major encode kernels and the fused encode + MLP kernel use (bit-identical by construction)
template <typename T16>
__device__ __forceinline__ uint32_t encode_pair(const GridParams& gp, const uint32_t* __restrict__ table, int l, float x, float y,
                                                float z, bool smooth) {
    const Corners c = corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
    const uint32_t* t = table + gp.offset[l];
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = t[c.idx[k]];
    float w[8];
    corner_weights(c.f, smooth, w);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a0 = fmaf(w[k], T16::lo(v[k]), a0);
        a1 = fmaf(w[k], T16::hi(v[k]), a1);
    }
    return T16::pack(a0, a1);
}

}  // namespace perf
