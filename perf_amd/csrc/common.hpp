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

}  // namespace perf
