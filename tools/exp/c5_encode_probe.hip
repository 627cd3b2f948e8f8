// Dev tool (round 6): BASELINE config 5's encode (L = 20 levels up to resolution 8192, 16-bit F = 2 tables of 2^T entries per
// hashed level) on real panorama sample positions, by TABLE LAYOUT and by thread mapping.  Stand-alone: no torch, no library.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exp/c5_encode_probe.hip -o /tmp/c5_probe && /tmp/c5_probe 28 [30]
// Layouts:  0 = tcnn (x + y*res + z*res^2 dense; prime-XOR hash of the vertex otherwise)
//           1 = line blocks: a 4x4x2 block of vertices = 32 entries = one 128-byte line; blocks hashed one by one
//           2 = line blocks inside 64 KiB super-blocks (8x8x8 blocks = 32x32x16 vertices contiguous); super-blocks hashed
//           3 = line blocks inside 2 MiB super-blocks (32x32x16 blocks = 128x128x32 vertices); 4: 64x64x128; 5: 64x64x64 (1 MiB); 6: 128x64x64
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


// work item of block b under the XCD-stable balanced mapping: XCD x = b % 8 serves level l for EIGHT CONSECUTIVE rays (256-sample
// chunks) back to back, then the next level of the same 64-ray stripe; which eighth of a stripe an XCD serves rotates with the
// level, so every XCD serves every level for an eighth of the rays (balance) and a level's neighbouring rays meet in one L2.
__device__ __forceinline__ bool stable_item(int64_t b, int64_t nchunks, int* l, int64_t* chunk) {
    const int x = (int)(b & 7);
    const int64_t j = b >> 3;
    const int cr = (int)(j & 7);
    const int64_t t = j >> 3;
    *l = (int)(t % L);
    const int64_t Q = t / L;
    *chunk = (((Q << 3) + ((x - *l) & 7)) << 3) + cr;
    return *chunk < nchunks;
}

// mapping 3: XCD-stable balanced, eight 4-byte gathers
__global__ __launch_bounds__(256) void enc_stable(GP gp, const float* __restrict__ x01, const uint32_t* __restrict__ table, uint32_t* __restrict__ feat, int64_t n) {
    int l; int64_t chunk;
    if (!stable_item(blockIdx.x, (n + 255) >> 8, &l, &chunk)) return;
    const int64_t i = chunk * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
    const Cor c = corners(gp, l, x, y, z);
    const uint32_t* t = table + gp.offset[l];
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = t[c.idx[k]];
    feat[(int64_t)l * n + i] = interp(c, v);
}

// mapping 4: XCD-stable balanced; line-local levels fetch the aligned 16-byte x-run of every (y, z) corner pair (both x corners
// unless the cell starts at the last vertex of a block: a fifth..eighth 4-byte gather for those lanes only)
template <int NT>
__global__ __launch_bounds__(256) void enc_stable_quad_t(GP gp, const float* __restrict__ x01, const uint32_t* __restrict__ table, uint32_t* __restrict__ feat, int64_t n) {
    int l; int64_t chunk;
    if (!stable_item(blockIdx.x, (n + 255) >> 8, &l, &chunk)) return;
    const int64_t i = chunk * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
    const uint32_t* t = table + gp.offset[l];
    uint32_t v[8];
    Cor c;
    if (!gp.local[l]) {
        c = corners(gp, l, x, y, z);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t[c.idx[k]];
    } else {
        const float s = gp.scale[l];
        const float px = __builtin_fmaf(x, s, 0.5f), py = __builtin_fmaf(y, s, 0.5f), pz = __builtin_fmaf(z, s, 0.5f);
        const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
        c.f[0] = px - fx; c.f[1] = py - fy; c.f[2] = pz - fz;
        const uint32_t gx = (uint32_t)(int)fx, gy = (uint32_t)(int)fy, gz = (uint32_t)(int)fz;
        const uint32_t lx = gx & 3u;
        uint4 q[4]; uint32_t e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t base = vertex_index(gp, l, gx & ~3u, gy + (k & 1), gz + (k >> 1));        // 16-byte aligned by construction
            if (NT) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                const u4 r = __builtin_nontemporal_load(reinterpret_cast<const u4*>(t + base));
                q[k] = make_uint4(r[0], r[1], r[2], r[3]);
            } else q[k] = *reinterpret_cast<const uint4*>(t + base);
        }
        if (lx == 3u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = t[vertex_index(gp, l, gx + 1u, gy + (k & 1), gz + (k >> 1))];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t a0 = lx == 0u ? q[k].x : (lx == 1u ? q[k].y : (lx == 2u ? q[k].z : q[k].w));
            const uint32_t a1 = lx == 0u ? q[k].y : (lx == 1u ? q[k].z : (lx == 2u ? q[k].w : e[k]));
            v[2 * k] = a0; v[2 * k + 1] = a1;
        }
    }
    feat[(int64_t)l * n + i] = interp(c, v);
}


// mapping 6: XCD-stable balanced; line-local levels give every sample FOUR LANES, one per (y, z) corner pair: the four 16-byte
// x-runs of a sample -- 2.3 lines on average -- are requested by ONE instruction (the texture addresser merges lanes that name the
// same line) instead of four consecutive ones that find the line pending; a wave serves 16 samples per instruction, four groups
// in flight; the rows' partial sums are added across the quad with DPP.
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    return v;
}

template <int G>
__global__ __launch_bounds__(256) void enc_stable_rows_t(GP gp, const float* __restrict__ x01, const uint32_t* __restrict__ table, uint32_t* __restrict__ feat, int64_t n) {
    int l; int64_t chunk;
    constexpr int CH = 64 * G;          // samples per workgroup (4 waves x 16 G)
    if (!stable_item(blockIdx.x, (n + CH - 1) / CH, &l, &chunk)) return;
    const uint32_t* t = table + gp.offset[l];
    if (!gp.local[l]) {
        for (int rep = 0; rep < CH / 256; ++rep) {
            const int64_t i = chunk * CH + rep * 256 + threadIdx.x;
            if (i >= n) return;
            const Cor c = corners(gp, l, x01[3 * i], x01[3 * i + 1], x01[3 * i + 2]);
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = t[c.idx[k]];
            feat[(int64_t)l * n + i] = interp(c, v);
        }
        return;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t s = lane >> 2, r = lane & 3u;
    const int64_t base = chunk * CH + wave * (16 * G);
    const float sc = gp.scale[l];
    uint4 q[G]; uint32_t e[G]; float fxs[G], wrow[G]; uint32_t lxs[G];
#pragma unroll
    for (int it = 0; it < G; ++it) e[it] = 0u;
#pragma unroll
    for (int it = 0; it < G; ++it) {
        int64_t i = base + 16 * it + s;
        if (i >= n) i = n - 1;
        const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
        const float px = __builtin_fmaf(x, sc, 0.5f), py = __builtin_fmaf(y, sc, 0.5f), pz = __builtin_fmaf(z, sc, 0.5f);
        const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
        const float fx = px - flx, fy = py - fly, fz = pz - flz;
        const uint32_t gx = (uint32_t)(int)flx, gy = (uint32_t)(int)fly, gz = (uint32_t)(int)flz;
        const uint32_t vy = gy + (r & 1u), vz = gz + (r >> 1);
        lxs[it] = gx & 3u; fxs[it] = fx;
        wrow[it] = ((r & 1u) ? fy : 1.f - fy);                      // (weights in tcnn's order: (wx * wy) * wz)
        const float wz = (r >> 1) ? fz : 1.f - fz;
        q[it] = *reinterpret_cast<const uint4*>(t + vertex_index(gp, l, gx & ~3u, vy, vz));
        if (lxs[it] == 3u) e[it] = t[vertex_index(gp, l, gx + 1u, vy, vz)];
        (void)wz;                                                   // (recomputed below from the position: cheaper than holding it)
    }
    uint32_t mine[G / 4];
#pragma unroll
    for (int k = 0; k < G / 4; ++k) mine[k] = 0u;
#pragma unroll
    for (int it = 0; it < G; ++it) {
        int64_t i = base + 16 * it + s;
        if (i >= n) i = n - 1;
        const float z = x01[3 * i + 2];
        const float pz = __builtin_fmaf(z, sc, 0.5f);
        const float fz = pz - floorf(pz);
        const float wz = (r >> 1) ? fz : 1.f - fz;
        const uint32_t lx = lxs[it];
        const uint32_t a0 = lx == 0u ? q[it].x : (lx == 1u ? q[it].y : (lx == 2u ? q[it].z : q[it].w));
        const uint32_t a1 = lx == 0u ? q[it].y : (lx == 1u ? q[it].z : (lx == 2u ? q[it].w : e[it]));
        const float w0 = ((1.f - fxs[it]) * wrow[it]) * wz, w1 = (fxs[it] * wrow[it]) * wz;
        float p0 = fmaf(w1, lo16(a1), w0 * lo16(a0)), p1 = fmaf(w1, hi16(a1), w0 * hi16(a0));
        p0 = quad_sum(p0); p1 = quad_sum(p1);
        if ((int)r == (it & 3)) mine[it >> 2] = pack_half2(p0, p1);
    }
#pragma unroll
    for (int k = 0; k < G / 4; ++k) {
        const int64_t io = base + 64 * k + 16 * r + s;
        if (io < n) feat[(int64_t)l * n + io] = mine[k];
    }
}


// panorama sample positions: rows [row0, row0 + nrows) of a 2048 x 4096 panorama, 256 lattice midpoints of 0.99 / 256 per ray
__global__ void positions(float* __restrict__ x01, int row0, int tile) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = 4ll * 4096 * 256;
    if (i >= n) return;
    const int k = (int)(i & 255);
    const int ray = (int)(i >> 8);                       // 16,384 rays: 4 rows x 4096 columns, or a 128 x 128 pixel tile
    const int row = tile ? row0 + (ray >> 7) : row0 + (ray >> 12), col = tile ? 1984 + (ray & 127) : (ray & 4095);
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

static uint32_t g_local_min_res = 64;
static GP make_grid(int log2_t, int layout, uint64_t* total) {
    GP g{};
    const double b = std::exp(std::log(8192.0 / 16.0) / (L - 1));
    const uint64_t T = 1ull << log2_t;
    const int sh[10][3] = {{0, 0, 0}, {2, 2, 1}, {5, 5, 4}, {7, 7, 5}, {6, 6, 7}, {6, 6, 6}, {7, 6, 6}, {5, 5, 9}, {6, 6, 8}, {5, 6, 8}};
    for (int d = 0; d < 3; ++d) g.sb_shift[d] = sh[layout][d];
    uint64_t off = 0;
    for (int l = 0; l < L; ++l) {
        const double s = 16.0 * std::pow(b, l) - 1.0;
        g.scale[l] = (float)s;
        const uint32_t res = (uint32_t)std::ceil(s) + 1;
        g.res[l] = res;
        const bool local = layout > 0 && res >= g_local_min_res;
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
    const char* mnames[] = {"groups (shipped)", "one level per workgroup", "groups, 24 gathers in flight", "XCD-stable balanced", "XCD-stable balanced, 16-byte x-runs", "XCD-stable balanced, 16-byte x-runs, nontemporal", "XCD-stable balanced, four lanes per sample", "XCD-stable balanced, four lanes per sample, 8 groups in flight"};
    kern_t kerns[] = {enc_groups, enc_levels, enc_groups_deep, enc_stable, enc_stable_quad_t<0>, enc_stable_quad_t<1>, enc_stable_rows_t<4>, enc_stable_rows_t<8>};
    const int nchunks = (int)(n >> 8);
    const unsigned grids[] = {4096u * 8, 4096u * L, 4096u * 8, (unsigned)((nchunks + 63) / 64 * L * 64), (unsigned)((nchunks + 63) / 64 * L * 64), (unsigned)((nchunks + 63) / 64 * L * 64), (unsigned)((nchunks + 63) / 64 * L * 64), (unsigned)((nchunks / 2 + 63) / 64 * L * 64)};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // (layout, mapping, tile) triples of this run
    // (layout, mapping, tile, local_min_res)
    const int variants[][4] = {{9, 6, 0, 64}, {9, 7, 0, 64}, {9, 6, 1, 64}, {9, 7, 1, 64}};
    for (int a = 1; a < argc; ++a) {
        const int log2_t = atoi(argv[a]);
        uint64_t maxtot = 0;
        for (int layout = 0; layout < 10; ++layout) { uint64_t t; make_grid(log2_t, layout, &t); if (t > maxtot) maxtot = t; }
        uint32_t* table;
        CHECK(hipMalloc(&table, maxtot * 4));
        fill<<<65536, 256>>>(table, maxtot);
        CHECK(hipDeviceSynchronize());
        {   // four-lanes-per-sample kernel against the 16-byte x-run kernel (another summation order: not bit-identical)
            uint32_t* feat2; CHECK(hipMalloc(&feat2, n * 4 * L));
            g_local_min_res = 16;
            uint64_t tot; const GP gp = make_grid(log2_t, 9, &tot);
            positions<<<(unsigned)((n + 255) / 256), 256>>>(x01, 1022, 0);
            enc_stable_quad_t<0><<<grids[4], 256>>>(gp, x01, table, feat, n);
            enc_stable_rows_t<8><<<grids[7], 256>>>(gp, x01, table, feat2, n);
            CHECK(hipDeviceSynchronize());
            std::vector<uint32_t> ha((size_t)n * L), hb((size_t)n * L);
            CHECK(hipMemcpy(ha.data(), feat, n * 4 * L, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hb.data(), feat2, n * 4 * L, hipMemcpyDeviceToHost));
            size_t bad = 0; double worst = 0;
            for (size_t k = 0; k < ha.size(); ++k) if (ha[k] != hb[k]) {
                ++bad;
                for (int h = 0; h < 2; ++h) {
                    const float a = (float)__builtin_bit_cast(_Float16, (unsigned short)(ha[k] >> (16 * h))), b = (float)__builtin_bit_cast(_Float16, (unsigned short)(hb[k] >> (16 * h)));
                    const double d = std::fabs((double)a - b) / (std::fabs((double)a) + 1e-6); if (d > worst) worst = d;
                }
            }
            printf("{\"check\": \"four lanes per sample vs 16-byte x-runs\", \"log2_T\": %d, \"differing_features\": %zu, \"of\": %zu, \"worst_relative\": %.3g}\n", log2_t, bad, ha.size(), worst);
            CHECK(hipFree(feat2));
        }
        for (const auto& var : variants) {
            const int layout = var[0], m = var[1], tile = var[2];
            g_local_min_res = (uint32_t)var[3];
            uint64_t tot;
            const GP gp = make_grid(log2_t, layout, &tot);
            int nh = 0, nl = 0; for (int l = 0; l < L; ++l) { nh += gp.hashed[l]; nl += gp.local[l]; }
            double sum = 0; double per_row[3];
            for (int r = 0; r < 3; ++r) {
                positions<<<(unsigned)((n + 255) / 256), 256>>>(x01, tile ? (rows[r] < 64 ? 0 : rows[r] - 62) : rows[r], tile);
                kerns[m]<<<grids[m], 256>>>(gp, x01, table, feat, n);             // warm-up
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                for (int rep = 0; rep < 3; ++rep) kerns[m]<<<grids[m], 256>>>(gp, x01, table, feat, n);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                per_row[r] = ms / 3; sum += ms / 3;
            }
            const double ms = sum / 3;
            printf("{\"log2_T\": %d, \"layout\": %d, \"mapping\": \"%s\", \"rays\": \"%s\", \"table_GiB\": %.2f, \"hashed_levels\": %d, \"local_levels\": %d, \"sb\": [%d, %d, %d], "
                   "\"ms_equator\": %.3f, \"ms_mid\": %.3f, \"ms_pole\": %.3f, \"ms_mean\": %.3f, \"algorithmic_frac_of_8TBps\": %.3f}\n",
                   log2_t, layout, mnames[m], tile ? "128x128 tile" : "4x4096 strip", tot * 4 / 1073741824.0, nh, nl, 1 << gp.sb_shift[0], 1 << gp.sb_shift[1], 1 << gp.sb_shift[2], per_row[0], per_row[1], per_row[2], ms,
                   640.0 * n / (ms * 1e-3) / 8e12);
            fflush(stdout);
        }
        CHECK(hipFree(table));
    }
    return 0;
}
