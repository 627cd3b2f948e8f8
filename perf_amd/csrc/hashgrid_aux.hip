// Hash grid, the rest: job-wide fixed-point units and slots of a data-parallel step (perf_dp_*), conversion of summed integer fields
// (perf_fixed_unfix), the input gradient of tcnn.Encoding and its second-order pieces (perf_hashgrid_bwd_input / _bwd_bwd_*).
#include <stdlib.h>
#include "common.hpp"
#include "grid_device.hpp"
#include "grid_fixed_point.hpp"

namespace perf {

// ---- job-wide fixed-point units for data-parallel training ----------------------------------------------------------------
// Every rank of a data-parallel step scatters ITS samples into integer fields; the ranks' tables can be added up exactly
// (an integer reduce-scatter) -- and equal the single-process table bit for bit -- iff all ranks use the units the single
// process would: derived from the job-wide max |dfeat| per level, the job-wide live sample count and the headroom state
// driven by the largest field of the SUMMED table of the previous step.  Ranks exchange one small block of statistics
// (perf_dp_stats_pack -> all-gather -> perf_dp_units) between the MLP backward and the grid backward.
__global__ void dp_stats_pack_kernel(const float* __restrict__ level_absmax, const int32_t* __restrict__ field_max_prev,
                                     const int64_t* __restrict__ n_dev, int64_t n, int32_t* __restrict__ stats) {
    const int i = threadIdx.x;
    if (i < PERF_MAX_LEVELS) {
        stats[i] = __float_as_int(level_absmax[i]);
        stats[PERF_MAX_LEVELS + i] = field_max_prev ? field_max_prev[i] : -1;
    } else if (i == 2 * PERF_MAX_LEVELS) {
        const int64_t live = live_count(n, n_dev);
        stats[i] = (int32_t)(live & 0xffffffffll);
        stats[i + 1] = (int32_t)(live >> 32);
    } else if (i > 2 * PERF_MAX_LEVELS + 1 && i < PERF_DP_STATS) {
        stats[i] = 0;
    }
}

__global__ void dp_units_kernel(GridParams gp, const int32_t* __restrict__ stats_all, int world, int32_t* __restrict__ hr_state,
                                int32_t* __restrict__ shifts, int64_t* __restrict__ n_total_out, int margin_bits) {
    __shared__ long long total_s;
    if (threadIdx.x == 0) {
        long long tot = 0;
        for (int r = 0; r < world; ++r) {
            const int32_t* st = stats_all + (int64_t)r * PERF_DP_STATS + 2 * PERF_MAX_LEVELS;
            tot += (long long)(uint32_t)st[0] | ((long long)st[1] << 32);
        }
        total_s = tot;
        if (n_total_out) n_total_out[0] = tot;
    }
    __syncthreads();
    const int l = threadIdx.x;
    if (l >= gp.n_levels) return;
    float am = 0.f;
    int fm = -1;
    for (int r = 0; r < world; ++r) {
        const int32_t* st = stats_all + (int64_t)r * PERF_DP_STATS;
        am = fmaxf(am, __int_as_float(st[l]));
        fm = max(fm, st[PERF_MAX_LEVELS + l]);
    }
    if (fm >= 0) hr_state[l] = headroom_feedback(hr_state[l], fm);        // (-1: no previous call, nothing to feed back)
    // margin_bits > 0: the units come from the PREVIOUS step's statistics (lagged mode, see perf_dp_slot_pack).  The closed loop
    // lets the headroom h of a level sink to 4 bits where an entry's contributions cancel (late in a phase the gradient is
    // noise): ONE contribution of a sample whose |dfeat| is 2^(h-2) times last step's maximum then reaches the flag level --
    // |dfeat| is heavy tailed, and a two-episode soak with h as the exact units have it lost 33 of 9,000 steps to such
    // outliers (tools/exp/dp_lag_diag.py: sporadic, in the second half of the geometry phase).  Lagged units therefore keep at
    // least kLaggedMinHeadroom bits (a single contribution needs 2^11 times last step's maximum to flag; the unit stays below
    // 2^-18 of that maximum) and add margin_bits on top.
    int sh = fixed_point_shift(am, total_s, gp.size[l], hr_state, l);
    if (margin_bits > 0) {
        int e = 0;
        if (am > 0.f) (void)frexpf(am, &e);
        if (e < -80) e = -80;
        int h = 31 - e - sh;
        if (h < kLaggedMinHeadroom) h = kLaggedMinHeadroom;
        sh = 31 - e - (h + margin_bits);
        // ... and never get more than kLaggedMaxFinerBits finer than the units of the step before (shifts[] still holds them: a
        // lagged call always follows a call that set it).  max |dfeat| is heavy tailed DOWNWARDS too: the depth loss of a batch the
        // field already fits vanishes (1e-21, 3e-38, 0 observed), units derived from that are 2^50 times too fine for the
        // ordinary batch that follows, and the job-wide gate dropped that step -- 10-12 of the 300 geometry steps of
        // tests/golden/psnr_curve.json's schedule at every margin from 1 to 6 bits (tools/exp/dp_margin_sweep.py).  Units may get
        // coarser at once.
        const int prev = shifts[l];
        if (sh > prev + kLaggedMaxFinerBits) sh = prev + kLaggedMaxFinerBits;
    }
    shifts[l] = sh;
}

// ---- the small all-reduce of a data-parallel step: one slot of PERF_DP_SLOT floats per rank behind the MLP weight gradient ----
// A SUM all-reduce over a buffer in which every rank fills only ITS slot is an all-gather; integers travel as 16-bit pieces
// (exact in fp32).  Slot layout: [0,24) max |dfeat| per level; [24,48) / [48,72) low / high 16 bits of the largest |field| per
// level of the rank's slice of THIS step's summed table; [72,76) the live sample count in 16-bit pieces; [76] overflow flag
// (local grid backward OR the rank's slice of the summed table); [77] batch truncated at the sample capacity.
__global__ void dp_slot_pack_kernel(const float* __restrict__ level_absmax, const int32_t* __restrict__ field_max,
                                    const int64_t* __restrict__ n_dev, int64_t n, const int32_t* __restrict__ overflow_flag,
                                    const int64_t* __restrict__ n_marched_dev, int64_t capacity, int rank, int world,
                                    float* __restrict__ slots) {
    for (int i = threadIdx.x; i < world * PERF_DP_SLOT; i += blockDim.x) {
        float v = 0.f;
        const int r = i / PERF_DP_SLOT, k = i % PERF_DP_SLOT;
        if (r == rank) {
            if (k < PERF_MAX_LEVELS) v = level_absmax ? level_absmax[k] : 0.f;
            else if (k < 2 * PERF_MAX_LEVELS) v = field_max ? (float)(field_max[k - PERF_MAX_LEVELS] & 0xffff) : 0.f;
            else if (k < 3 * PERF_MAX_LEVELS) v = field_max ? (float)((uint32_t)field_max[k - 2 * PERF_MAX_LEVELS] >> 16) : 0.f;
            else if (k < 3 * PERF_MAX_LEVELS + 4) {
                const uint64_t live = (uint64_t)live_count(n, n_dev);
                v = (float)((live >> (16 * (k - 3 * PERF_MAX_LEVELS))) & 0xffffull);
            } else if (k == 3 * PERF_MAX_LEVELS + 4) v = (overflow_flag && overflow_flag[0] != 0) ? 1.f : 0.f;
            else if (k == 3 * PERF_MAX_LEVELS + 5) v = (n_marched_dev && capacity > 0 && n_marched_dev[0] > capacity) ? 1.f : 0.f;
        }
        slots[i] = v;
    }
}

// after the all-reduce: the ranks' slots -> the statistics block perf_dp_units reads (as if all-gathered by
// perf_dp_stats_pack, with the field maxima of THIS step), the job-wide flags {overflow, truncated} perf_step_bookkeeping
// reads as remote_flags, and the job's sample count
__global__ void dp_slot_unpack_kernel(const float* __restrict__ slots, int world, int32_t* __restrict__ stats_all,
                                      float* __restrict__ job_flags, int64_t* __restrict__ n_total_out) {
    if (threadIdx.x == 0) {
        float ovf = 0.f, trunc = 0.f;
        long long tot = 0;
        for (int r = 0; r < world; ++r) {
            const float* s = slots + (int64_t)r * PERF_DP_SLOT + 3 * PERF_MAX_LEVELS;
            ovf += s[4]; trunc += s[5];
            tot += (long long)s[0] + ((long long)s[1] << 16) + ((long long)s[2] << 32) + ((long long)s[3] << 48);
        }
        if (job_flags) { job_flags[0] = ovf; job_flags[1] = trunc; }
        if (n_total_out) n_total_out[0] = tot;
    }
    if (!stats_all) return;
    for (int i = threadIdx.x; i < world * PERF_DP_STATS; i += blockDim.x) {
        const int r = i / PERF_DP_STATS, k = i % PERF_DP_STATS;
        const float* s = slots + (int64_t)r * PERF_DP_SLOT;
        int32_t v = 0;
        if (k < PERF_MAX_LEVELS) v = __float_as_int(s[k]);
        else if (k < 2 * PERF_MAX_LEVELS) v = (int32_t)s[k] | ((int32_t)s[k + PERF_MAX_LEVELS] << 16);
        else if (k == 2 * PERF_MAX_LEVELS) v = (int32_t)s[3 * PERF_MAX_LEVELS] | ((int32_t)s[3 * PERF_MAX_LEVELS + 1] << 16);
        else if (k == 2 * PERF_MAX_LEVELS + 1) v = (int32_t)s[3 * PERF_MAX_LEVELS + 2] | ((int32_t)s[3 * PERF_MAX_LEVELS + 3] << 16);
        stats_all[i] = v;
    }
}

// int32 field pairs of table entries [entry_lo, entry_hi) -> fp32 gradients, in place; per-level largest |field| of the
// slice (atomicMax into field_max, zeroed by the caller) and the overflow flag
__global__ __launch_bounds__(256) void fixed_unfix_kernel(GridParams gp, int32_t* __restrict__ buf, int64_t entry_lo, int64_t entry_hi,
                                                          const int32_t* __restrict__ shifts, int32_t* __restrict__ field_max,
                                                          int32_t* __restrict__ overflow_flag) {
    __shared__ int32_t fm_s[PERF_MAX_LEVELS];
    // level starts and units in LDS: gp.offset[l] with a per-lane l is a vector load from the kernel-argument segment, waited for
    // with vmcnt(0) -- i.e. behind the data loads, once per entry (this kernel took 44 us for 53 MB)
    __shared__ uint64_t start_s[PERF_MAX_LEVELS + 1];
    __shared__ float unit_s[PERF_MAX_LEVELS];
    if (threadIdx.x < PERF_MAX_LEVELS) {
        fm_s[threadIdx.x] = 0;
        start_s[threadIdx.x] = (int)threadIdx.x < gp.n_levels ? gp.offset[threadIdx.x] : ~0ull;
        unit_s[threadIdx.x] = (int)threadIdx.x < gp.n_levels ? ldexpf(1.0f, -shifts[threadIdx.x]) : 0.f;
    }
    if (threadIdx.x == 0) start_s[PERF_MAX_LEVELS] = ~0ull;
    __syncthreads();
    int cur_l = 0, cur_m = 0;            // a thread's entries ascend: it stays in one level for long runs
    float from_fixed = unit_s[0];
    // The walk is "load, convert, store IN PLACE": a load behind a store through the same pointer waits for it, so eight entries
    // are loaded before the first of them is stored (a load per iteration was a round trip per entry).
    // (a workgroup takes 2,048 consecutive entries at a time: its threads change level together, and rarely)
    for (int64_t e0 = entry_lo + (int64_t)blockIdx.x * 2048 + threadIdx.x; e0 < entry_hi; e0 += (int64_t)gridDim.x * 2048) {
        int2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t e = e0 + j * 256;
            v[j] = reinterpret_cast<const int2*>(buf)[(e < entry_hi ? e : entry_hi - 1) - entry_lo];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t e = e0 + j * 256;
            if (e >= entry_hi) break;
            if ((uint64_t)e >= start_s[cur_l + 1]) {
                if (cur_m > 0) atomicMax(&fm_s[cur_l], cur_m);
                while ((uint64_t)e >= start_s[cur_l + 1]) ++cur_l;
                cur_m = 0;
                from_fixed = unit_s[cur_l];
            }
            if (v[j].x == 0 && v[j].y == 0) continue;                  // (integer 0 is 0.0f)
            reinterpret_cast<float2*>(buf)[e - entry_lo] = make_float2((float)v[j].x * from_fixed, (float)v[j].y * from_fixed);
            const int32_t ax = v[j].x < 0 ? -(v[j].x + 1) : v[j].x, ay = v[j].y < 0 ? -(v[j].y + 1) : v[j].y;
            cur_m = max(cur_m, max(ax, ay));
        }
    }
    // (a wave's lanes nearly always end in the same level: one LDS atomic per wave instead of 64 on one address)
    const int l0 = __shfl(cur_l, 0);
    if (__all(cur_l == l0)) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cur_m = max(cur_m, __shfl_xor(cur_m, off));
        if ((threadIdx.x & 63) == 0 && cur_m > 0) atomicMax(&fm_s[cur_l], cur_m);
    } else if (cur_m > 0) {
        atomicMax(&fm_s[cur_l], cur_m);
    }
    __syncthreads();
    if (threadIdx.x < PERF_MAX_LEVELS && fm_s[threadIdx.x] > 0) {
        // same-address read-modify-writes retire one after the other (~11 ns each): a maximum only needs the workgroups that
        // RAISE it -- a handful of thousands -- so look first (an atomic load is served by the L2 like any other)
        if (field_max && fm_s[threadIdx.x] > __hip_atomic_load(&field_max[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&field_max[threadIdx.x], fm_s[threadIdx.x]);
        if (overflow_flag && fm_s[threadIdx.x] >= (1 << 29)) atomicOr(overflow_flag, 1);
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

// ---- second order: the backward of the input gradient (tcnn kernel_grid_backward_input_backward_*) --------------------
// The input gradient  gx_i = sum_l sum_c (d w_c / d x_i) (theta[idx_c] . dy_l)  is linear in dy and in the table and
// non-linear in x.  Given gg = dL/d gx [n,3] its backward has three pieces:
//   d_dy[l]      = sum_c W'_c theta[idx_c]                     with  W'_c = sum_i gg_i d w_c / d x_i
//   d_theta[idx] += W'_c dy_l                                  (hashgrid_bwd_bwd_param_kernel)
//   d_x_j        = sum_l sum_c (sum_i gg_i d^2 w_c / d x_i d x_j) (theta[idx_c] . dy_l)
// with  w_c = prod_d u_d,  u_d = s_d or 1 - s_d,  s_d = f_d (Linear) or f_d^2 (3 - 2 f_d) (Smoothstep),  d s_d / d x_d = s' scale.
// Consumer: SphereDistanceField (modules/geo_predictors/pano_joint_predictor.py:50-69: autograd.grad(distance, directions,
// create_graph=True) followed by a loss on that gradient).
struct Interp { float s[3], ds[3], dds[3]; };        // per dimension: value, d/dx, d^2/dx^2 (scale folded in)

__device__ __forceinline__ Interp interp_of(const float f[3], bool smooth, float scale) {
    Interp t;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (smooth) { t.s[d] = f[d] * f[d] * (3.f - 2.f * f[d]); t.ds[d] = 6.f * f[d] * (1.f - f[d]) * scale; t.dds[d] = (6.f - 12.f * f[d]) * scale * scale; }
        else { t.s[d] = f[d]; t.ds[d] = scale; t.dds[d] = 0.f; }
    }
    return t;
}

// W'_c = sum_i gg_i d w_c / d x_i   for corner c (bit0 = x, bit1 = y, bit2 = z)
__device__ __forceinline__ float corner_dw_dot(const Interp& t, int c, const float gg[3]) {
    const float u[3] = {(c & 1) ? t.s[0] : 1.f - t.s[0], (c & 2) ? t.s[1] : 1.f - t.s[1], (c & 4) ? t.s[2] : 1.f - t.s[2]};
    const float sg[3] = {(c & 1) ? 1.f : -1.f, (c & 2) ? 1.f : -1.f, (c & 4) ? 1.f : -1.f};
    return gg[0] * sg[0] * t.ds[0] * u[1] * u[2] + gg[1] * sg[1] * t.ds[1] * u[0] * u[2] + gg[2] * sg[2] * t.ds[2] * u[0] * u[1];
}

__global__ __launch_bounds__(256) void hashgrid_bwd_bwd_input_kernel(GridParams gp, const float* __restrict__ x01,
                                                                     const float2* __restrict__ dy, const float2* __restrict__ table,
                                                                     const float* __restrict__ ggx, float2* __restrict__ d_dy,
                                                                     float* __restrict__ d_x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
    const float gg[3] = {ggx[3 * i], ggx[3 * i + 1], ggx[3 * i + 2]};
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
    float hx[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < gp.n_levels; ++l) {
        const Corners c = corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
        const Interp t = interp_of(c.f, smooth, gp.scale[l]);
        const float2* tb = table + gp.offset[l];
        const float2 g = dy ? dy[(int64_t)l * n + i] : make_float2(0.f, 0.f);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 v = tb[c.idx[k]];
            a0 += corner_dw_dot(t, k, gg) * v.x;
            a1 += corner_dw_dot(t, k, gg) * v.y;
            if (d_x) {
                const float dot = v.x * g.x + v.y * g.y;
                const float u[3] = {(k & 1) ? t.s[0] : 1.f - t.s[0], (k & 2) ? t.s[1] : 1.f - t.s[1], (k & 4) ? t.s[2] : 1.f - t.s[2]};
                const float sg[3] = {(k & 1) ? 1.f : -1.f, (k & 2) ? 1.f : -1.f, (k & 4) ? 1.f : -1.f};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int p = (j + 1) % 3, q = (j + 2) % 3;
                    // sum_i gg_i d^2 w / dx_i dx_j: the diagonal term and the two mixed terms
                    const float h = gg[j] * sg[j] * t.dds[j] * u[p] * u[q]
                                  + gg[p] * sg[p] * t.ds[p] * sg[j] * t.ds[j] * u[q]
                                  + gg[q] * sg[q] * t.ds[q] * sg[j] * t.ds[j] * u[p];
                    hx[j] += h * dot;
                }
            }
        }
        if (d_dy) d_dy[(int64_t)l * n + i] = make_float2(a0, a1);
    }
    if (d_x) { d_x[3 * i] = hx[0]; d_x[3 * i + 1] = hx[1]; d_x[3 * i + 2] = hx[2]; }
}

// d_theta[idx_c] += W'_c dy_l   (one thread per (sample, level); global fp32 atomics: this consumer's batches are 10^4
// points, see the comment on hashgrid_bwd_atomic_kernel for the rate)
__global__ __launch_bounds__(256) void hashgrid_bwd_bwd_param_kernel(GridParams gp, const float* __restrict__ x01,
                                                                     const float2* __restrict__ dy, const float* __restrict__ ggx,
                                                                     float* __restrict__ grad, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= n || l >= gp.n_levels) return;
    const float2 g = dy[(int64_t)l * n + i];
    if (g.x == 0.f && g.y == 0.f) return;
    const float gg[3] = {ggx[3 * i], ggx[3 * i + 1], ggx[3 * i + 2]};
    const Corners c = corners_of(x01[3 * i], x01[3 * i + 1], x01[3 * i + 2], gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
    const Interp t = interp_of(c.f, gp.interpolation == PERF_INTERP_SMOOTHSTEP, gp.scale[l]);
    float* tb = grad + 2 * gp.offset[l];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = corner_dw_dot(t, k, gg);
        unsafeAtomicAdd(tb + 2 * (uint64_t)c.idx[k], w * g.x);
        unsafeAtomicAdd(tb + 2 * (uint64_t)c.idx[k] + 1, w * g.y);
    }
}

}  // namespace perf

using namespace perf;

extern "C" int perf_dp_stats_pack(const float* level_absmax, const int32_t* field_max_prev, const int64_t* n_dev, int64_t n,
                                  int32_t* stats_out, void* stream) {
    PERF_REQUIRE(level_absmax && stats_out, "NULL pointer");
    dp_stats_pack_kernel<<<dim3(1), dim3(64), 0, as_stream(stream)>>>(level_absmax, field_max_prev, n_dev, n, stats_out);
    PERF_LAUNCH_CHECK("perf_dp_stats_pack");
    return PERF_OK;
}

extern "C" int perf_dp_units(const perf_grid_desc* grid, const int32_t* stats_all, int32_t world, int32_t* headroom_state,
                             int32_t* shifts_out, int64_t* n_total_out, int32_t margin_bits, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    PERF_REQUIRE(stats_all && headroom_state && shifts_out && world >= 1 && margin_bits >= 0 && margin_bits <= 8, "perf_dp_units: bad arguments");
    dp_units_kernel<<<dim3(1), dim3(64), 0, as_stream(stream)>>>(gp, stats_all, world, headroom_state, shifts_out, n_total_out, margin_bits);
    PERF_LAUNCH_CHECK("perf_dp_units");
    return PERF_OK;
}

extern "C" int perf_dp_slot_pack(const float* level_absmax, const int32_t* field_max, const int64_t* n_dev, int64_t n,
                                 const int32_t* overflow_flag, const int64_t* n_marched_dev, int64_t capacity, int32_t rank,
                                 int32_t world, float* slots, void* stream) {
    PERF_REQUIRE(slots && world >= 1 && rank >= 0 && rank < world, "perf_dp_slot_pack: bad arguments");
    dp_slot_pack_kernel<<<dim3(1), dim3(256), 0, as_stream(stream)>>>(level_absmax, field_max, n_dev, n, overflow_flag, n_marched_dev,
                                                                       capacity, rank, world, slots);
    PERF_LAUNCH_CHECK("perf_dp_slot_pack");
    return PERF_OK;
}

extern "C" int perf_dp_slot_unpack(const float* slots, int32_t world, int32_t* stats_all, float* job_flags, int64_t* n_total_out,
                                   void* stream) {
    PERF_REQUIRE(slots && world >= 1, "perf_dp_slot_unpack: bad arguments");
    dp_slot_unpack_kernel<<<dim3(1), dim3(256), 0, as_stream(stream)>>>(slots, world, stats_all, job_flags, n_total_out);
    PERF_LAUNCH_CHECK("perf_dp_slot_unpack");
    return PERF_OK;
}

extern "C" int perf_fixed_unfix(const perf_grid_desc* grid, void* fields, int64_t entry_lo, int64_t entry_hi,
                                const int32_t* shifts_dev, int32_t* field_max, int32_t* overflow_flag, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    PERF_REQUIRE(fields && shifts_dev, "NULL pointer");
    const int64_t total = (int64_t)(gp.offset[gp.n_levels - 1] + gp.size[gp.n_levels - 1]);
    PERF_REQUIRE(entry_lo >= 0 && entry_lo <= entry_hi && entry_hi <= total, "perf_fixed_unfix: bad entry range");
    if (field_max) PERF_REQUIRE(hipMemsetAsync(field_max, 0, PERF_MAX_LEVELS * sizeof(int32_t), as_stream(stream)) == hipSuccess, "memset failed");
    if (entry_hi == entry_lo) return PERF_OK;
    // few workgroups: each ends with atomics on the 24 maxima, which share one cache line and retire one at a time (~11 ns):
    // 4,096 workgroups spent 30 us there (tools/exp/unfix_probe.py: 56 / 40 / 35 / 39 us at 4096 / 1024 / 512 / 256)
    constexpr int64_t kMaxBlocks = 512;       // (more workgroups only queue at the 24 same-line maxima: tools/exp/unfix_probe.py)
    int64_t blocks = div_up(entry_hi - entry_lo, 256 * 8);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    fixed_unfix_kernel<<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(gp, (int32_t*)fields, entry_lo, entry_hi, shifts_dev,
                                                                                       field_max, overflow_flag);
    PERF_LAUNCH_CHECK("perf_fixed_unfix");
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

extern "C" int perf_hashgrid_bwd_bwd_input(const perf_grid_desc* grid, const float* x01, const float* dfeat, const float* table,
                                           const float* ggx, float* d_dfeat, float* d_x, int64_t n, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && table && ggx, "NULL pointer");
    PERF_REQUIRE(d_dfeat || d_x, "perf_hashgrid_bwd_bwd_input: nothing to compute");
    PERF_REQUIRE(!d_x || dfeat, "perf_hashgrid_bwd_bwd_input: d_x needs dfeat");
    hipLaunchKernelGGL(hashgrid_bwd_bwd_input_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, as_stream(stream), gp, x01,
                       (const float2*)dfeat, (const float2*)table, ggx, (float2*)d_dfeat, d_x, n);
    PERF_LAUNCH_CHECK("perf_hashgrid_bwd_bwd_input");
    return PERF_OK;
}

extern "C" int perf_hashgrid_bwd_bwd_param(const perf_grid_desc* grid, const float* x01, const float* dfeat, const float* ggx,
                                           float* grad_table, int64_t n, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    PERF_REQUIRE(grad_table, "NULL pointer");
    const uint64_t total = gp.offset[gp.n_levels - 1] + gp.size[gp.n_levels - 1];
    PERF_REQUIRE(hipMemsetAsync(grad_table, 0, (size_t)total * 2 * sizeof(float), as_stream(stream)) == hipSuccess,
                 "perf_hashgrid_bwd_bwd_param: memset failed");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && dfeat && ggx, "NULL pointer");
    hipLaunchKernelGGL(hashgrid_bwd_bwd_param_kernel, dim3((unsigned)div_up(n, 256), gp.n_levels), dim3(256), 0, as_stream(stream),
                       gp, x01, (const float2*)dfeat, ggx, grad_table, n);
    PERF_LAUNCH_CHECK("perf_hashgrid_bwd_bwd_param");
    return PERF_OK;
}
