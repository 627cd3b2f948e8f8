// The second stage of the MLP backward (dw[i] = sum over the workgroups' partial gradients, fixed order; per-level max |dfeat| from
// the waves' slots) as a device function, so that it can ride in another launch: perf_field_bwd folds it into the tile-code
// pre-pass of the grid backward -- independent work that follows the same producer (mlp_bwd_kernel) -- instead of paying a launch
// of its own (one graph node of the training step).
#pragma once
#include "common.hpp"

namespace perf {

struct MlpReduceJob {
    const float* partials;       // [n_partials][n_params]
    float* dw;                   // [n_params]
    const float* amax_slots;     // [n_partials * 4][2] or NULL
    float* level_absmax;         // [PERF_MAX_LEVELS] or NULL
    int32_t n_params, n_partials, n_levels;
    int32_t n_blocks;            // workgroups of 256 threads the job takes: div_up(n_params, 16) + 1; 0 = no job
};

// workgroup `block` of `job.n_blocks` (256 threads)
__device__ __forceinline__ void mlp_reduce_block(const MlpReduceJob& j, int block) {
    __shared__ float acc[4][64];
    if (block == j.n_blocks - 1) {
        // extra block: level_absmax[l] = max over all (wave, half) slots of the half that owns level l
        if (j.level_absmax == nullptr) return;
        float m0 = 0.f, m1 = 0.f;
        for (int k = threadIdx.x; k < j.n_partials * 4; k += 256) { m0 = fmaxf(m0, j.amax_slots[2 * k]); m1 = fmaxf(m1, j.amax_slots[2 * k + 1]); }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, off)); m1 = fmaxf(m1, __shfl_xor(m1, off)); }
        if ((threadIdx.x & 63) == 0) { acc[0][threadIdx.x >> 6] = m0; acc[1][threadIdx.x >> 6] = m1; }
        __syncthreads();
        if (threadIdx.x < PERF_MAX_LEVELS) {
            const int h = (threadIdx.x >> 1) & 1;          // levels {0,1,4,5,..} live in half 0, {2,3,6,7,..} in half 1
            const float v = fmaxf(fmaxf(acc[h][0], acc[h][1]), fmaxf(acc[h][2], acc[h][3]));
            j.level_absmax[threadIdx.x] = (int)threadIdx.x < j.n_levels ? v : 0.f;
        }
        return;
    }
    // 16 parameters x 16 partial-segments per block (one 64-byte sector per row), 4 independent loads in flight
    const int pl = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int pi = block * 16 + pl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (pi < j.n_params) {
        int k = seg;
        for (; k + 48 < j.n_partials; k += 64) {
            s0 += j.partials[(int64_t)k * j.n_params + pi];
            s1 += j.partials[(int64_t)(k + 16) * j.n_params + pi];
            s2 += j.partials[(int64_t)(k + 32) * j.n_params + pi];
            s3 += j.partials[(int64_t)(k + 48) * j.n_params + pi];
        }
        for (; k < j.n_partials; k += 16) s0 += j.partials[(int64_t)k * j.n_params + pi];
    }
    float* a = &acc[0][0];                          // 256 floats: [seg][param]
    a[seg * 16 + pl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (seg == 0 && pi < j.n_params) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += a[q * 16 + pl];
        j.dw[pi] = t;
    }
}

}  // namespace perf

// library-internal entry points (hidden: not part of the C ABI) that perf_field_bwd chains
#define PERF_INTERNAL __attribute__((visibility("hidden")))
PERF_INTERNAL int perf_internal_mlp_bwd(const perf_mlp_desc* mlp, const void* w16, const void* feat16, const int32_t* feat_index, int64_t feat_stride,
                                        const uint8_t* sel, const float* dout, float* dfeat, float* dw, float* level_absmax, void* workspace,
                                        int64_t workspace_bytes, int64_t n, const int64_t* n_dev, int dtype, void* stream, perf::MlpReduceJob* defer);
PERF_INTERNAL int perf_internal_hashgrid_bwd(const perf_grid_desc* grid, const float* x01, const float* dfeat, float* grad_table, int64_t n,
                                             const int64_t* n_dev, int accumulate, const float* level_absmax, int32_t* overflow_flag,
                                             int32_t* headroom_state, const int32_t* shifts_dev, int raw_fields, const int32_t* redo_flag,
                                             void* workspace, int64_t workspace_bytes, void* stream, const perf::MlpReduceJob* job,
                                             const perf_step_book* book = nullptr);
PERF_INTERNAL void perf_internal_launch_mlp_reduce(const perf::MlpReduceJob& job, void* stream);
