// Multiresolution hash-grid encoding for gfx950 (tcnn "HashGrid" semantics).
//
// Design (MI355X-first): the encode is a random 4-byte-gather kernel whose working set
// (13 MB of 16-bit tables per encoder) exceeds one XCD's 4 MiB L2.  Work is therefore cut
// by LEVEL GROUP, not by sample alone: block b serves level group (b % 8) -- with the
// dispatcher's round-robin block->XCD placement each XCD's private L2 then only ever sees
// the two levels {g, L-1-g} of its group (<= 2 MiB), so gathers are L2 hits instead of
// Infinity-Cache round trips.  Placement is a speed assumption only: results do not depend
// on it.  Features leave the kernel LEVEL-MAJOR (feat[l][sample] as one packed 2x16-bit
// dword), so every store and the MLP kernel's loads are fully coalesced.
#include "common.hpp"

namespace perf {

struct GridParams {
    int32_t n_levels;
    int32_t interpolation;
    float scale[PERF_MAX_LEVELS];
    uint32_t res[PERF_MAX_LEVELS];
    uint32_t size[PERF_MAX_LEVELS];
    uint32_t offset[PERF_MAX_LEVELS];
    uint32_t hashed[PERF_MAX_LEVELS];
};

static int fill_params(const perf_grid_desc* g, GridParams* p) {
    PERF_REQUIRE(g != nullptr, "grid desc is NULL");
    PERF_REQUIRE(g->n_levels >= 1 && g->n_levels <= PERF_MAX_LEVELS, "n_levels %d out of range", g->n_levels);
    p->n_levels = g->n_levels;
    p->interpolation = g->interpolation;
    for (int l = 0; l < PERF_MAX_LEVELS; ++l) {
        p->scale[l] = g->scale[l]; p->res[l] = g->res[l]; p->size[l] = g->size[l];
        p->offset[l] = g->offset[l]; p->hashed[l] = g->hashed[l];
        if (l < g->n_levels) {
            PERF_REQUIRE(g->size[l] > 0, "level %d has size 0", l);
            if (g->hashed[l]) PERF_REQUIRE((g->size[l] & (g->size[l] - 1)) == 0, "hashed level %d size %u is not a power of two", l, g->size[l]);
        }
    }
    return PERF_OK;
}

constexpr uint32_t kPrimeY = 2654435761u;
constexpr uint32_t kPrimeZ = 805459861u;

// Corner bookkeeping of one (sample, level): 8 table indices + fractional position.
// pos = fl(fl(x*scale)+0.5) (unfused, matches oracle/perf_oracle.py:grid_corner_indices).
struct Corners {
    uint32_t idx[8];
    float f[3];
};

__device__ __forceinline__ Corners corners_of(float x, float y, float z, float scale, uint32_t res,
                                              uint32_t size, bool hashed) {
    Corners c;
    float px = add_rn(mul_rn(x, scale), 0.5f);
    float py = add_rn(mul_rn(y, scale), 0.5f);
    float pz = add_rn(mul_rn(z, scale), 0.5f);
    float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    c.f[0] = px - flx; c.f[1] = py - fly; c.f[2] = pz - flz;
    uint32_t gx = (uint32_t)(int32_t)flx, gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
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
        for (int k = 0; k < 8; ++k) {
            uint32_t i = base + (uint32_t)(k & 1) + ((k & 2) ? res : 0u) + ((k & 4) ? r2 : 0u);
            if (i >= size) i = i % size;
            c.idx[k] = i;
        }
    }
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

// level l handled by (group, pass): pass 0 -> g, pass 1 -> L-1-g (if different)
__device__ __forceinline__ int level_of(int group, int pass, int L) {
    int a = group, b = L - 1 - group;
    if (a > b) return -1;
    if (pass == 0) return a;
    return (b != a) ? b : -1;
}

template <typename T16>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(GridParams gp, const float* __restrict__ x01,
                                                           const uint32_t* __restrict__ table,
                                                           uint32_t* __restrict__ feat, int64_t n) {
    const int group = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int l = level_of(group, pass, gp.n_levels);
        if (l < 0) continue;
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
        feat[(int64_t)l * n + i] = T16::pack(a0, a1);
    }
}

__global__ __launch_bounds__(256) void hashgrid_fwd_f32_kernel(GridParams gp, const float* __restrict__ x01,
                                                               const float2* __restrict__ table,
                                                               float2* __restrict__ feat, int64_t n) {
    const int group = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int l = level_of(group, pass, gp.n_levels);
        if (l < 0) continue;
        const Corners c = corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
        const float2* t = table + gp.offset[l];
        float w[8];
        corner_weights(c.f, smooth, w);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float2 v = t[c.idx[k]];
            a0 = fmaf(w[k], v.x, a0);
            a1 = fmaf(w[k], v.y, a1);
        }
        feat[(int64_t)l * n + i] = make_float2(a0, a1);
    }
}

// Parameter gradient: scatter w_c * dfeat into the fp32 gradient table with hardware fp32
// atomics (global_atomic_add_f32).  Samples whose incoming gradient is exactly zero (masked by
// the selector, or pruned) issue no atomics.
__global__ __launch_bounds__(256) void hashgrid_bwd_kernel(GridParams gp, const float* __restrict__ x01,
                                                           const float2* __restrict__ dfeat,
                                                           float* __restrict__ grad, int64_t n) {
    const int group = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int l = level_of(group, pass, gp.n_levels);
        if (l < 0) continue;
        const float2 g = dfeat[(int64_t)l * n + i];
        if (g.x == 0.f && g.y == 0.f) continue;
        const Corners c = corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
        float w[8];
        corner_weights(c.f, smooth, w);
        float* t = grad + 2 * (int64_t)gp.offset[l];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsafeAtomicAdd(t + 2 * (int64_t)c.idx[k], w[k] * g.x);
            unsafeAtomicAdd(t + 2 * (int64_t)c.idx[k] + 1, w[k] * g.y);
        }
    }
}

// Input gradient dL/dx (fp32 table).  One thread walks all levels of its sample.
__global__ __launch_bounds__(256) void hashgrid_bwd_input_kernel(GridParams gp, const float* __restrict__ x01,
                                                                 const float2* __restrict__ dfeat,
                                                                 const float2* __restrict__ table,
                                                                 float* __restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int l = 0; l < gp.n_levels; ++l) {
        const float2 g = dfeat[(int64_t)l * n + i];
        const Corners c = corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
        const float2* t = table + gp.offset[l];
        float f[3] = {c.f[0], c.f[1], c.f[2]};
        float s[3], ds[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (smooth) { s[d] = f[d] * f[d] * (3.f - 2.f * f[d]); ds[d] = 6.f * f[d] * (1.f - f[d]); }
            else { s[d] = f[d]; ds[d] = 1.f; }
        }
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float2 v = t[c.idx[k]];
            float dot = v.x * g.x + v.y * g.y;
            float wx = (k & 1) ? s[0] : 1.f - s[0], wy = (k & 2) ? s[1] : 1.f - s[1], wz = (k & 4) ? s[2] : 1.f - s[2];
            float sx = (k & 1) ? 1.f : -1.f, sy = (k & 2) ? 1.f : -1.f, sz = (k & 4) ? 1.f : -1.f;
            ax += sx * wy * wz * dot; ay += wx * sy * wz * dot; az += wx * wy * sz * dot;
        }
        gx += ax * ds[0] * gp.scale[l]; gy += ay * ds[1] * gp.scale[l]; gz += az * ds[2] * gp.scale[l];
    }
    dx[3 * i] = gx; dx[3 * i + 1] = gy; dx[3 * i + 2] = gz;
}

static inline unsigned grouped_grid(int64_t n) { return (unsigned)(div_up(n, 256) * 8); }

}  // namespace perf

using namespace perf;

extern "C" int perf_hashgrid_fwd(const perf_grid_desc* grid, const float* x01, const void* table16,
                                 void* feat16, int64_t n, int dtype, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    PERF_REQUIRE(n >= 0 && n < (int64_t(1) << 31) * 16, "n out of range");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && table16 && feat16, "NULL pointer");
    dim3 g(grouped_grid(n)), b(256);
    if (dtype == PERF_DTYPE_BF16)
        hipLaunchKernelGGL(hashgrid_fwd_kernel<BF16>, g, b, 0, as_stream(stream), gp, x01, (const uint32_t*)table16, (uint32_t*)feat16, n);
    else if (dtype == PERF_DTYPE_FP16)
        hipLaunchKernelGGL(hashgrid_fwd_kernel<FP16>, g, b, 0, as_stream(stream), gp, x01, (const uint32_t*)table16, (uint32_t*)feat16, n);
    else { set_error("perf_hashgrid_fwd: bad dtype %d", dtype); return PERF_E_INVALID; }
    PERF_LAUNCH_CHECK("perf_hashgrid_fwd");
    return PERF_OK;
}

extern "C" int perf_hashgrid_fwd_f32(const perf_grid_desc* grid, const float* x01, const float* table,
                                     float* feat, int64_t n, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && table && feat, "NULL pointer");
    hipLaunchKernelGGL(hashgrid_fwd_f32_kernel, dim3(grouped_grid(n)), dim3(256), 0, as_stream(stream), gp, x01,
                       (const float2*)table, (float2*)feat, n);
    PERF_LAUNCH_CHECK("perf_hashgrid_fwd_f32");
    return PERF_OK;
}

extern "C" int perf_hashgrid_bwd(const perf_grid_desc* grid, const float* x01, const float* dfeat,
                                 float* grad_table, int64_t n, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && dfeat && grad_table, "NULL pointer");
    hipLaunchKernelGGL(hashgrid_bwd_kernel, dim3(grouped_grid(n)), dim3(256), 0, as_stream(stream), gp, x01,
                       (const float2*)dfeat, grad_table, n);
    PERF_LAUNCH_CHECK("perf_hashgrid_bwd");
    return PERF_OK;
}

extern "C" int perf_hashgrid_bwd_input(const perf_grid_desc* grid, const float* x01, const float* dfeat,
                                       const float* table, float* dx, int64_t n, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && dfeat && table && dx, "NULL pointer");
    hipLaunchKernelGGL(hashgrid_bwd_input_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, as_stream(stream), gp,
                       x01, (const float2*)dfeat, (const float2*)table, dx, n);
    PERF_LAUNCH_CHECK("perf_hashgrid_bwd_input");
    return PERF_OK;
}
