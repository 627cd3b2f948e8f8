// Dev tool (round 6): BASELINE config 5's encode (L = 20 levels up to resolution 8192, 16-bit F = 2 tables of 2^T entries per
// hashed level) on real panorama sample positions, by TABLE LAYOUT and by thread mapping.  Stand-alone: no torch, no library.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exp/c5_encode_probe.hip -o /tmp/c5_probe && /tmp/c5_probe 28 [30]
// Layouts:  0 = tcnn (x + y*res + z*res^2 dense; prime-XOR hash of the vertex otherwise)
//           1 = line blocks: a 4x4x2 block of vertices = 32 entries = one 128-byte line; blocks hashed one by one
//           2 = line blocks inside 64 KiB super-blocks (8x8x8 blocks = 32x32x16 vertices contiguous); super-blocks hashed
//           3 = line blocks inside 2 MiB super-blocks (32x32x16 blocks = 128x128x32 vertices)
// Mappings: 0 = shipped generic kernel (block -> level group b % 8, levels {g, 15-g, 16+g} one after the other)
//           1 = one level per workgroup, levels cycling with the block index (every XCD serves every level)
//           2 = shipped groups, all of a thread's gathers issued before the first is consumed
// Prints one JSON line per (T, rows, layout, mapping): ms per 4.2 M-sample launch, algorithmic fraction of 8 TB/s (640 B/sample).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int L = 20;
constexpr uint32_t kPY = 2654435761u, kPZ = 805459861u;

struct GP {
    float scale[L];
    uint32_t res[L];
    uint32_t size[L];        // entries of the level
    uint64_t offset[L];      // entry offset of the level
    uint32_t hashed[L];
    uint32_t local[L];       // 1: line-local layout at this level
    uint32_t nsx[L], nsy[L]; // dense line-local levels: super-blocks per row / per slice
    uint32_t sb_shift[3];    // log2 vertices per super-block along x, y, z (layouts 2, 3); layout 1: {2, 2, 1}
};

__device__ __forceinline__ uint32_t pack_half2(float a, float b) {
    return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)a) | ((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)b) << 16);
}
__device__ __forceinline__ float lo16(uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu)); }
__device__ __forceinline__ float hi16(uint32_t v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); }

// entry index of vertex (vx, vy, vz) at level l
__device__ __forceinline__ uint32_t vertex_index(const GP& gp, int l, uint32_t vx, uint32_t vy, uint32_t vz) {
    const uint32_t size = gp.size[l];
    if (!gp.local[l]) {
        if (gp.hashed[l]) return (vx ^ (vy * kPY) ^ (vz * kPZ)) & (size - 1u);
        const uint32_t res = gp.res[l];
        uint32_t i = vx + vy * res + vz * res * res;
        if (i >= size) i = i % size;
        return i;
    }
    const uint32_t shx = gp.sb_shift[0], shy = gp.sb_shift[1], shz = gp.sb_shift[2];
    const uint32_t sx = vx >> shx, sy = vy >> shy, sz = vz >> shz;
    const uint32_t per_sb = 1u << (shx + shy + shz);                 // entries of a super-block
    uint32_t slot;
    if (gp.hashed[l]) slot = (sx ^ (sy * kPY) ^ (sz * kPZ)) & (size / per_sb - 1u);
    else slot = sx + sy * gp.nsx[l] + sz * gp.nsy[l];
    // blocks of a super-block x-major, vertices of a block x-major
    const uint32_t bx = (vx >> 2) & ((1u << (shx - 2)) - 1u), by = (vy >> 2) & ((1u << (shy - 2)) - 1u), bz = (vz >> 1) & ((1u << (shz - 1)) - 1u);
    const uint32_t blk = bx + (by << (shx - 2)) + (bz << (shx - 2 + shy - 2));
    const uint32_t within = (vx & 3u) + ((vy & 3u) << 2) + ((vz & 1u) << 4);
    return slot * per_sb + (blk << 5) + within;
}

struct Cor { uint32_t idx[8]; float f[3]; };

__device__ __forceinline__ Cor corners(const GP& gp, int l, float x, float y, float z) {
    Cor c;
    const float s = gp.scale[l];
    const float px = __builtin_fmaf(x, s, 0.5f), py = __builtin_fmaf(y, s, 0.5f), pz = __builtin_fmaf(z, s, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    c.f[0] = px - fx; c.f[1] = py - fy; c.f[2] = pz - fz;
    const uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.idx[k] = vertex_index(gp, l, gx + (k & 1), gy + ((k >> 1) & 1), gz + (k >> 2));
    return c;
}

__device__ __forceinline__ uint32_t interp(const Cor& c, const uint32_t v[8]) {
    const float wx[2] = {1.f - c.f[0], c.f[0]}, wy[2] = {1.f - c.f[1], c.f[1]}, wz[2] = {1.f - c.f[2], c.f[2]};
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float w = (wx[k & 1] * wy[(k >> 1) & 1]) * wz[k >> 2];
        a0 = fmaf(w, lo16(v[k]), a0); a1 = fmaf(w, hi16(v[k]), a1);
    }
    return pack_half2(a0, a1);
}

__device__ __forceinline__ int level_of(int group, int pass) {
    if (pass == 2) return (16 + group < L) ? 16 + group : -1;
    const int a = group, b = 15 - group;
    if (a > b) return -1;
    if (pass == 0) return a;
    return (b != a) ? b : -1;
}

// mapping 0: the shipped generic kernel
__global__ __launch_bounds__(256) void enc_groups(GP gp, const float* __restrict__ x01, const uint32_t* __restrict__ table, uint32_t* __restrict__ feat, int64_t n) {
    const int nchunks = (int)(gridDim.x >> 3);
    const int group = (int)(blockIdx.x & 7);
    for (int64_t chunk = (int64_t)(blockIdx.x >> 3); chunk * 256 < n; chunk += nchunks) {
        const int64_t i = chunk * 256 + threadIdx.x;
        if (i >= n) break;
        const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            const int l = level_of(group, pass);
            if (l < 0) continue;
            const Cor c = corners(gp, l, x, y, z);
            const uint32_t* t = table + gp.offset[l];
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = t[c.idx[k]];
            feat[(int64_t)l * n + i] = interp(c, v);
        }
    }
}

// mapping 2: shipped groups, all gathers of a thread in flight together
__global__ __launch_bounds__(256) void enc_groups_deep(GP gp, const float* __restrict__ x01, const uint32_t* __restrict__ table, uint32_t* __restrict__ feat, int64_t n) {
    const int nchunks = (int)(gridDim.x >> 3);
    const int group = (int)(blockIdx.x & 7);
    for (int64_t chunk = (int64_t)(blockIdx.x >> 3); chunk * 256 < n; chunk += nchunks) {
        const int64_t i = chunk * 256 + threadIdx.x;
        if (i >= n) break;
        const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
        Cor c[3]; uint32_t v[3][8]; int lv[3];
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            lv[pass] = level_of(group, pass);
            const int l = lv[pass] < 0 ? 0 : lv[pass];
            c[pass] = corners(gp, l, x, y, z);
            const uint32_t* t = table + gp.offset[l];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[pass][k] = t[c[pass].idx[k]];
        }
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
            if (lv[pass] >= 0) feat[(int64_t)lv[pass] * n + i] = interp(c[pass], v[pass]);
    }
}

// mapping 1: one level per workgroup; block b -> level order[b % L], chunk b / L (looping)
__global__ __launch_bounds__(256) void enc_levels(GP gp, const float* __restrict__ x01, const uint32_t* __restrict__ table, uint32_t* __restrict__ feat, int64_t n) {
    const int nchunks = (int)(gridDim.x / L);
    const int l = (int)(blockIdx.x % L);
    for (int64_t chunk = (int64_t)(blockIdx.x / L); chunk * 256 < n; chunk += nchunks) {
        const int64_t i = chunk * 256 + threadIdx.x;
        if (i >= n) break;
        const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
        const Cor c = corners(gp, l, x, y, z);
        const uint32_t* t = table + gp.offset[l];
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t[c.idx[k]];
        feat[(int64_t)l * n + i] = interp(c, v);
    }
}

// panorama sample positions: rows [row0, row0 + nrows) of a 2048 x 4096 panorama, 256 lattice midpoints of 0.99 / 256 per ray
__global__ void positions(float* __restrict__ x01, int row0, int nrows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)nrows * 4096 * 256;
    if (i >= n) return;
    const int k = (int)(i & 255), col = (int)((i >> 8) & 4095), row = row0 + (int)(i >> 20);
    const float yy = (row + .5f) / 2048.f, xx = (col + .5f) / 4096.f;
    const float beta = -(yy - .5f) * 3.14159265358979f, alpha = -(xx - .5f) * 6.28318530717959f;
    const float t = (k + .5f) * (0.99f / 256.f);
    x01[3 * i] = cosf(alpha) * cosf(beta) * t * .5f + .5f;
    x01[3 * i + 1] = sinf(alpha) * cosf(beta) * t * .5f + .5f;
    x01[3 * i + 2] = sinf(beta) * t * .5f + .5f;
}

__global__ void fill(uint32_t* t, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) t[i] = 0x2e662e66u ^ (uint32_t)(i * 2654435761u & 0x03ff03ffu);
}

static GP make_grid(int log2_t, int layout, uint64_t* total) {
    GP g{};
    const double b = std::exp(std::log(8192.0 / 16.0) / (L - 1));
    const uint64_t T = 1ull << log2_t;
    const int sh[4][3] = {{0, 0, 0}, {2, 2, 1}, {5, 5, 4}, {7, 7, 5}};
    for (int d = 0; d < 3; ++d) g.sb_shift[d] = sh[layout][d];
    uint64_t off = 0;
    for (int l = 0; l < L; ++l) {
        const double s = 16.0 * std::pow(b, l) - 1.0;
        g.scale[l] = (float)s;
        const uint32_t res = (uint32_t)std::ceil(s) + 1;
        g.res[l] = res;
        const bool local = layout > 0 && res >= 64;
        g.local[l] = local;
        uint64_t want;
        if (!local) {
            want = ((uint64_t)res * res * res + 7) / 8 * 8;
        } else {
            const uint64_t nx = (res + (1u << sh[layout][0])) >> sh[layout][0], ny = (res + (1u << sh[layout][1])) >> sh[layout][1],      // (vertices 0..res)
                           nz = (res + (1u << sh[layout][2])) >> sh[layout][2];
            g.nsx[l] = (uint32_t)nx; g.nsy[l] = (uint32_t)(nx * ny);
            want = nx * ny * nz << (sh[layout][0] + sh[layout][1] + sh[layout][2]);
        }
        g.hashed[l] = want > T;
        g.size[l] = (uint32_t)(want > T ? T : want);
        g.offset[l] = off;
        off += g.size[l];
    }
    *total = off;
    return g;
}

typedef void (*kern_t)(GP, const float*, const uint32_t*, uint32_t*, int64_t);

int main(int argc, char** argv) {
    const int64_t n = 4ll * 4096 * 256;
    float* x01; uint32_t* feat;
    CHECK(hipMalloc(&x01, n * 12)); CHECK(hipMalloc(&feat, n * 4 * L));
    const int rows[] = {1022, 512, 0};
    const char* mnames[] = {"groups (shipped)", "one level per workgroup", "groups, 24 gathers in flight"};
    kern_t kerns[] = {enc_groups, enc_levels, enc_groups_deep};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int a = 1; a < argc; ++a) {
        const int log2_t = atoi(argv[a]);
        uint64_t maxtot = 0;
        for (int layout = 0; layout < 4; ++layout) { uint64_t t; make_grid(log2_t, layout, &t); if (t > maxtot) maxtot = t; }
        uint32_t* table;
        CHECK(hipMalloc(&table, maxtot * 4));
        fill<<<65536, 256>>>(table, maxtot);
        CHECK(hipDeviceSynchronize());
        for (int layout = 0; layout < 4; ++layout) {
            uint64_t tot;
            const GP gp = make_grid(log2_t, layout, &tot);
            int nh = 0, nl = 0; for (int l = 0; l < L; ++l) { nh += gp.hashed[l]; nl += gp.local[l]; }
            for (int m = 0; m < 3; ++m) {
                double sum = 0; double per_row[3];
                for (int r = 0; r < 3; ++r) {
                    positions<<<(unsigned)((n + 255) / 256), 256>>>(x01, rows[r], 4);
                    const unsigned grid = m == 1 ? 4096u * L : 4096u * 8;
                    kerns[m]<<<grid, 256>>>(gp, x01, table, feat, n);             // warm-up
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipEventRecord(e0));
                    for (int rep = 0; rep < 3; ++rep) kerns[m]<<<grid, 256>>>(gp, x01, table, feat, n);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    per_row[r] = ms / 3; sum += ms / 3;
                }
                const double ms = sum / 3;
                printf("{\"log2_T\": %d, \"layout\": %d, \"mapping\": \"%s\", \"table_GiB\": %.2f, \"hashed_levels\": %d, \"local_levels\": %d, "
                       "\"ms_equator\": %.3f, \"ms_mid\": %.3f, \"ms_pole\": %.3f, \"ms_mean\": %.3f, \"algorithmic_frac_of_8TBps\": %.3f}\n",
                       log2_t, layout, mnames[m], tot * 4 / 1073741824.0, nh, nl, per_row[0], per_row[1], per_row[2], ms,
                       640.0 * n / (ms * 1e-3) / 8e12);
                fflush(stdout);
            }
        }
        CHECK(hipFree(table));
    }
    return 0;
}
