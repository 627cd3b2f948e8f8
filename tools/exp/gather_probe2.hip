// Dev tool (round 6, second session): what does a random gather from an L2-RESIDENT table cost the L1, by request width?
//   hipcc --offload-arch=gfx950 -O3 tools/exp/gather_probe2.hip -o /tmp/gather_probe2 && /tmp/gather_probe2 [log2_bytes ...]
// gather_probe.hip forms its addresses with two 64-bit multiplies -- at cache-resident table sizes that arithmetic, not the memory path,
// is its limit (270 G requests/s at every width).  Here an address costs one 32-bit multiply: K independent 4 / 8 / 16-byte loads per lane
// at pseudo-random aligned addresses, and -- "pair" -- the encode's case: HALF the lanes issue one 8-byte load, the other half two 4-byte
// loads to unrelated addresses (a cell whose first x vertex is even / odd in tcnn's hashed layout).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; return x; }

enum { W4 = 0, W8 = 1, W16 = 2, PAIR = 3, TWO4 = 4 };

template <int MODE, int K>
__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ tab, uint32_t mask_words, uint32_t* __restrict__ out, uint32_t seed, int rounds) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint32_t v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t a = rnd(tid * 2654435761u + (uint32_t)(r * K + k) * 0x85EBCA6Bu + seed) & mask_words;
            if (MODE == W4) v[k] = tab[a];
            else if (MODE == W8) { const uint2 q = *reinterpret_cast<const uint2*>(tab + (a & ~1u)); v[k] = q.x ^ q.y; }
            else if (MODE == W16) { const uint4 q = *reinterpret_cast<const uint4*>(tab + (a & ~3u)); v[k] = q.x ^ q.y ^ q.z ^ q.w; }
            else if (MODE == TWO4) { v[k] = tab[a] ^ tab[rnd(a + 0x9E3779B9u) & mask_words]; }
            else {      // PAIR: even lanes one 8-byte request, odd lanes two 4-byte requests
                if ((tid & 1u) == 0u) { const uint2 q = *reinterpret_cast<const uint2*>(tab + (a & ~1u)); v[k] = q.x ^ q.y; }
                else v[k] = tab[a] ^ tab[rnd(a + 0x9E3779B9u) & mask_words];
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= v[k];
    }
    out[tid] = acc;
}

template <int MODE, int K>
static void run(const char* name, double pairs_per_load, const uint32_t* tab, uint32_t words, uint32_t* out, int blocks, int rounds) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    probe<MODE, K><<<blocks, 256>>>(tab, words - 1, out, 1u, rounds);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) probe<MODE, K><<<blocks, 256>>>(tab, words - 1, out, 7u + i, rounds);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const double items = (double)blocks * 256 * K * rounds;
    printf("{\"variant\": \"%s\", \"K\": %d, \"ms\": %.3f, \"G_items_per_s\": %.1f, \"G_x_pairs_per_s\": %.1f}\n", name, K, ms, items / ms * 1e-6, items * pairs_per_load / ms * 1e-6);
}

int main(int argc, char** argv) {
    const int blocks = 1 << 15;
    uint32_t* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    for (int i = 1; i < (argc > 1 ? argc : 2); ++i) {
        const int lg = argc > 1 ? atoi(argv[i]) : 20;
        const uint32_t bytes = 1u << lg, words = bytes / 4;
        uint32_t* tab;
        CHECK(hipMalloc(&tab, bytes));
        CHECK(hipMemset(tab, 1, bytes));
        printf("{\"table_MiB\": %.2f}\n", bytes / 1048576.0);
        // "x pairs": how many (x, x+1) corner pairs of the encode one item stands for
        run<W4, 16>("4-byte gather (half a pair)", 0.5, tab, words, out, blocks, 4);
        run<TWO4, 8>("two 4-byte gathers (a pair, odd x)", 1.0, tab, words, out, blocks, 4);
        run<W8, 16>("8-byte gather (a pair, even x)", 1.0, tab, words, out, blocks, 4);
        run<W16, 16>("16-byte gather", 1.0, tab, words, out, blocks, 4);
        run<PAIR, 8>("half the lanes 8 bytes, half two 4-byte gathers (a pair each)", 1.0, tab, words, out, blocks, 4);
        CHECK(hipFree(tab));
    }
    return 0;
}
