// Shared device/host helpers for libperf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/perf_hip.h"

namespace perf {

void set_error(const char* fmt, ...);

#define PERF_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            perf::set_error(__VA_ARGS__);       \
            return PERF_E_INVALID;              \
        }                                       \
    } while (0)

#define PERF_LAUNCH_CHECK(name)                                                   \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            perf::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return PERF_E_LAUNCH;                                                 \
        }                                                                         \
    } while (0)

constexpr int kWave = 64;
constexpr int kNumCU = 256;
constexpr int kNumXCD = 8;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// ---- 16-bit storage types -----------------------------------------------------------------
struct BF16 {
    using scalar = __bf16;
    using vec8 = bf16x8;
    static __device__ __forceinline__ float lo(uint32_t pair) { return __uint_as_float(pair << 16); }
    static __device__ __forceinline__ float hi(uint32_t pair) { return __uint_as_float(pair & 0xffff0000u); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) __bf16 v2;
        v2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;
        return __builtin_bit_cast(uint32_t, p);
    }
    static __device__ __forceinline__ uint16_t one(float a) { return __builtin_bit_cast(uint16_t, (__bf16)a); }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {      // 16 x 16 x 32
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

struct FP16 {
    using scalar = _Float16;
    using vec8 = f16x8;
    static __device__ __forceinline__ float lo(uint32_t pair) {
        return (float)__builtin_bit_cast(_Float16, (uint16_t)(pair & 0xffffu));
    }
    static __device__ __forceinline__ float hi(uint32_t pair) {
        return (float)__builtin_bit_cast(_Float16, (uint16_t)(pair >> 16));
    }
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) _Float16 v2;
        v2 p; p[0] = (_Float16)a; p[1] = (_Float16)b;
        return __builtin_bit_cast(uint32_t, p);
    }
    static __device__ __forceinline__ uint16_t one(float a) { return __builtin_bit_cast(uint16_t, (_Float16)a); }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {      // 16 x 16 x 32
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// unfused fp32 arithmetic for bookkeeping that must match numpy bit for bit
// (HIP's __fmul_rn/__fadd_rn are plain operators that the compiler may contract into an FMA, so the
// wrappers switch contraction off locally; the library is also built with -ffp-contract=off.)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Device-side sample counts: per-sample kernels are launched for the CAPACITY n of their arrays (which is also the
// stride of level-major buffers) and process only the first min(n, *n_dev) samples when n_dev != NULL -- the count a
// preceding scan left in device memory.  No host read-back, so variable-count batches are hipGraph-capturable.
__device__ __forceinline__ int64_t live_count(int64_t n, const int64_t* __restrict__ n_dev) {
    if (!n_dev) return n;
    const int64_t m = *n_dev;
    return m < n ? (m > 0 ? m : 0) : n;
}

struct Aabb { float lo[3], hi[3]; };

__device__ __forceinline__ void normalize_store(float px, float py, float pz, const Aabb& bb, float* x01, uint8_t* sel,
                                                int64_t i) {
    // (x - aabb_min) / (aabb_max - aabb_min), IEEE division like torch
    float ux = __fdiv_rn(sub_rn(px, bb.lo[0]), sub_rn(bb.hi[0], bb.lo[0]));
    float uy = __fdiv_rn(sub_rn(py, bb.lo[1]), sub_rn(bb.hi[1], bb.lo[1]));
    float uz = __fdiv_rn(sub_rn(pz, bb.lo[2]), sub_rn(bb.hi[2], bb.lo[2]));
    x01[3 * i] = ux; x01[3 * i + 1] = uy; x01[3 * i + 2] = uz;
    if (sel) sel[i] = (ux > 0.f && ux < 1.f && uy > 0.f && uy < 1.f && uz > 0.f && uz < 1.f) ? 1 : 0;
}

// sample position of one lattice interval: t_origins + t_dirs * (t0+t1) / 2.0 (nerf_renderer.py:127) -- multiply,
// divide, add, unfused -- then the aabb normalisation above
__device__ __forceinline__ void sample_point_store(const float* o, const float* d, float t0, float t1, const Aabb& bb,
                                                   float* x01, uint8_t* sel, int64_t i) {
    const float tsum = add_rn(t0, t1);
    const float px = add_rn(o[0], __fdiv_rn(mul_rn(d[0], tsum), 2.0f));
    const float py = add_rn(o[1], __fdiv_rn(mul_rn(d[1], tsum), 2.0f));
    const float pz = add_rn(o[2], __fdiv_rn(mul_rn(d[2], tsum), 2.0f));
    normalize_store(px, py, pz, bb, x01, sel, i);
}

}  // namespace perf
