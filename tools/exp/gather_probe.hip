// Dev tool: how many random 64-byte sectors per second does one MI355X deliver from HBM, by load flavour?
//   hipcc --offload-arch=gfx950 -O3 tools/exp/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe [log2_bytes]
// Every lane issues K independent 4/8/16-byte loads at hashed (uniformly random, aligned) addresses of a table no cache holds
// (default 16 GiB), sums them and writes one word per lane.  Prints requests/s and requests x 64 B in TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

enum { PLAIN = 0, NT = 1, SC = 2, X2 = 3, X4 = 4 };

template <int MODE>
__device__ __forceinline__ uint32_t load_at(const uint32_t* p) {
    if (MODE == PLAIN) return *p;
    if (MODE == NT) return __builtin_nontemporal_load(p);
    if (MODE == SC) { uint32_t v; asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory"); return v; }
    if (MODE == X2) { const uint2 v = *reinterpret_cast<const uint2*>(p); return v.x ^ v.y; }
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    return v.x ^ v.y ^ v.z ^ v.w;
}

template <int MODE, int K>
__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ tab, uint64_t mask_words, uint32_t* __restrict__ out,
                                             uint32_t seed, int rounds) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    const uint64_t align = MODE == X2 ? ~1ull : (MODE == X4 ? ~3ull : ~0ull);
    for (int r = 0; r < rounds; ++r) {
        uint32_t v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = load_at<MODE>(tab + (mix(tid * 1315423911ull + (uint64_t)(r * K + k) * 2654435761ull + seed) & mask_words & align));
        if (MODE == SC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= v[k];
    }
    out[tid] = acc;
}

template <int MODE, int K>
static void run(const char* name, const uint32_t* tab, uint64_t words, uint32_t* out, int blocks, int rounds) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    probe<MODE, K><<<blocks, 256>>>(tab, words - 1, out, 1u, rounds);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    const int reps = 3;
    for (int i = 0; i < reps; ++i) probe<MODE, K><<<blocks, 256>>>(tab, words - 1, out, 7u + i, rounds);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double req = (double)blocks * 256 * K * rounds;
    printf("{\"variant\": \"%s\", \"loads_in_flight_per_lane\": %d, \"blocks\": %d, \"ms\": %.3f, \"Grequests_per_s\": %.1f, \"TBps_at_64B_per_request\": %.2f}\n",
           name, K, blocks, ms, req / ms * 1e-6, req * 64 / ms * 1e-9);
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 34;
    const uint64_t bytes = 1ull << lg, words = bytes / 4;
    uint32_t* tab; uint32_t* out;
    CHECK(hipMalloc(&tab, bytes));
    CHECK(hipMemset(tab, 1, bytes));
    const int blocks = 1 << 16;                    // 16.7 M lanes
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    printf("{\"table_GiB\": %.1f}\n", bytes / 1073741824.0);
    run<PLAIN, 8>("dword", tab, words, out, blocks, 2);
    run<PLAIN, 16>("dword", tab, words, out, blocks, 1);
    run<PLAIN, 32>("dword", tab, words, out, blocks, 1);
    run<NT, 16>("dword nt", tab, words, out, blocks, 1);
    run<SC, 16>("dword sc0 sc1", tab, words, out, blocks, 1);
    run<X2, 16>("dwordx2", tab, words, out, blocks, 1);
    run<X4, 16>("dwordx4", tab, words, out, blocks, 1);
    run<X4, 8>("dwordx4", tab, words, out, blocks, 2);
    return 0;
}
