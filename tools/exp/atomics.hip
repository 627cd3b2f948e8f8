// Experiment: throughput of fp32 / packed-16 atomics by scope and by address distribution.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) short s2;

__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(float* t, uint32_t mask, int64_t n, int coherent) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t key = coherent ? (uint32_t)(i >> 4) : (uint32_t)i;   // coherent: 16 consecutive lanes share addresses
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        uint32_t idx = rnd(key * 16 + c) & mask;
        float v = 1.0f;
        if (MODE == 0) { unsafeAtomicAdd(t + 2 * idx, v); unsafeAtomicAdd(t + 2 * idx + 1, v); }
        if (MODE == 1) { __hip_atomic_fetch_add(t + 2 * idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                         __hip_atomic_fetch_add(t + 2 * idx + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (MODE == 2) { h2 hv = {(_Float16)1.0f, (_Float16)1.0f}; __builtin_amdgcn_global_atomic_fadd_v2f16((h2*)(t) + idx, hv); }
        if (MODE == 3) { t[2 * idx] = v; t[2 * idx + 1] = v; }
        if (MODE == 4) { __hip_atomic_fetch_add(t + 2 * idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                         __hip_atomic_fetch_add(t + 2 * idx + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (MODE == 5) { __hip_atomic_fetch_add(t + 2 * idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                         __hip_atomic_fetch_add(t + 2 * idx + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
        if (MODE == 6) { atomicAdd((unsigned long long*)(t) + idx, 1ull); }
    }
}

template <int MODE>
float run(float* t, uint32_t mask, int64_t n, int coherent) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    dim3 g((n + 255) / 256);
    k<MODE><<<g, 256>>>(t, mask, n, coherent); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<MODE><<<g, 256>>>(t, mask, n, coherent);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    const int64_t n = 1 << 20;
    float* t; hipMalloc(&t, (size_t)8 << 20 << 2); hipMemset(t, 0, (size_t)8 << 20 << 2);
    const char* names[] = {"unsafeAtomicAdd(agent)", "fetch_add(workgroup)", "pk_add_f16(agent)", "plain store", "fetch_add(agent)", "fetch_add(wavefront)", "u64 add(agent)"};
    for (int coh = 0; coh < 2; ++coh)
        for (uint32_t bits : {12u, 18u, 21u}) {
            uint32_t mask = (1u << bits) - 1;
            float ms[7];
            ms[0] = run<0>(t, mask, n, coh); ms[1] = run<1>(t, mask, n, coh); ms[2] = run<2>(t, mask, n, coh);
            ms[3] = run<3>(t, mask, n, coh); ms[4] = run<4>(t, mask, n, coh); ms[5] = run<5>(t, mask, n, coh); ms[6] = run<6>(t, mask, n, coh);
            for (int m = 0; m < 7; ++m)
                printf("coherent=%d entries=2^%u %-26s %8.3f ms  %8.2f G addr-ops/s\n", coh, bits, names[m], ms[m], n * 16.0 / ms[m] / 1e6);
        }
    return 0;
}
