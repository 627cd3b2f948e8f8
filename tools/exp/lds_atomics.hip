// Experiment: cost of ds_add_f32 as a function of same-address conflict degree and active lanes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int run, int active_every, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32768; i += 1024) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool act = (lane % active_every) == 0;
    uint32_t a = ((lane / run) * 97u + (threadIdx.x >> 6) * 1031u) & 16383u;
    float v = 1.0f;
    for (int it = 0; it < iters; ++it) {
        if (act) {
            if (MODE == 0) { unsafeAtomicAdd(&lds[2 * a], v); unsafeAtomicAdd(&lds[2 * a + 1], v); }
            if (MODE == 1) { lds[2 * a] = v; lds[2 * a + 1] = v; }
            if (MODE == 2) { atomicAdd((unsigned int*)&lds[2 * a], 1u); atomicAdd((unsigned int*)&lds[2 * a + 1], 1u); }
        }
        a = (a * 5u + 1u) & 16383u;
        if (run > 1) a = (a / (uint32_t)run) * (uint32_t)run, a = __shfl(a, (lane / run) * run);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0];
}

template <int MODE>
float run(float* out, int r, int ae, int iters) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 1024, 131072>>>(out, r, ae, iters); hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<256, 1024, 131072>>>(out, r, ae, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    const int iters = 2000;
    printf("per wave-instruction-pair cost (ns) ; 16 waves/CU, 2 ds ops per iteration\n");
    for (int ae : {1, 4, 16})
        for (int r : {1, 2, 4, 8, 16, 64}) {
            float f = run<0>(out, r, ae, iters), s = run<1>(out, r, ae, iters), u = run<2>(out, r, ae, iters);
            // per CU: 16 waves * iters pairs
            printf("active 1/%-2d run=%-2d  f32-atomic %.1f ns  store %.1f ns  u32-atomic %.1f ns   (cycles@2.4GHz per pair per CU: %.0f / %.0f / %.0f)\n", ae, r,
                   f * 1e6 / (16.0 * iters), s * 1e6 / (16.0 * iters), u * 1e6 / (16.0 * iters),
                   f * 1e6 / (16.0 * iters) * 2.4, s * 1e6 / (16.0 * iters) * 2.4, u * 1e6 / (16.0 * iters) * 2.4);
        }
    return 0;
}
