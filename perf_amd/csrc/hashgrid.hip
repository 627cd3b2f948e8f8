// Multiresolution hash-grid encoding for gfx950 (tcnn "HashGrid" semantics).
//
// Design (MI355X-first): the encode is a random 4-byte-gather kernel whose working set
// (13 MB of 16-bit tables per encoder) exceeds one XCD's 4 MiB L2.  Work is therefore cut
// by LEVEL GROUP, not by sample alone: block b serves level group (b % 8) -- with the
// dispatcher's round-robin block->XCD placement each XCD's private L2 then only ever sees
// the two levels {g, L-1-g} of its group (<= 2 MiB), so gathers are L2 hits instead of
// Infinity-Cache round trips.  Placement is a speed assumption only: results do not depend
// on it.  Features leave the kernel LEVEL-MAJOR (feat[l][sample] as one packed 2x16-bit
// dword), so every store and the MLP kernel's loads are fully coalesced.
#include <stdlib.h>
#include <mutex>
#include "common.hpp"

#include "grid_device.hpp"

namespace perf {

constexpr int kHeadroomStartBias = 3;

// level l handled by (group, pass).  L <= 16: pass 0 -> g, pass 1 -> L-1-g (if different) -- a coarse (small) and a
// fine (large) table per group.  Deeper grids (L <= 24) add pass 2 -> 16+g; their tables exceed the L2 anyway.
constexpr int kFwdPasses = 3;
// at most this many 256-sample chunks per level group in one launch (the workgroups loop beyond): launches of up to 1 M
// samples keep one workgroup per chunk (measured equal either way), while a capacity-sized launch of an eval frame -- 33 M
// rows for a tail pass that holds a few thousand live samples -- no longer dispatches 10^6 workgroups that only read the
// device-side count and leave (0.25 ms per frame)
constexpr int64_t kFwdMaxChunks = 4096;
__device__ __forceinline__ int level_of(int group, int pass, int L) {
    if (pass == 2) return (16 + group < L) ? 16 + group : -1;
    const int Lc = L < 16 ? L : 16;
    int a = group, b = Lc - 1 - group;
    if (a > b) return -1;
    if (pass == 0) return a;
    return (b != a) ? b : -1;
}

template <typename T16>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(GridParams gp, const float* __restrict__ x01,
                                                           const uint32_t* __restrict__ table,
                                                           uint32_t* __restrict__ feat, int64_t n,
                                                           const int64_t* __restrict__ n_dev, int xcd_affinity) {
    // xcd_affinity == 0 (experiment only): consecutive blocks of one XCD walk through all level groups, so every L2
    // sees the whole table -- used to measure what the level-group <-> XCD pinning is worth.
    const int nchunks = (int)(gridDim.x >> 3);
    const int group = xcd_affinity ? (int)(blockIdx.x & 7) : (int)((blockIdx.x >> 3) & 7);
    const int64_t chunk0 = xcd_affinity ? (int64_t)(blockIdx.x >> 3)
                                        : (int64_t)(blockIdx.x & 7) * ((nchunks + 7) >> 3) + (int64_t)(blockIdx.x >> 6);
    if (!xcd_affinity && chunk0 >= nchunks) return;
    const int64_t n_live = live_count(n, n_dev);                 // (n stays the level stride)
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
    // chunk-stride loop: a capacity-sized launch (n >> n_live) is capped at `nchunks` chunks per level group, so that
    // it does not pay for tens of thousands of workgroups that only find out that they have nothing to do
    for (int64_t chunk = chunk0; chunk * 256 < n_live; chunk += nchunks) {
    const int64_t i = chunk * 256 + threadIdx.x;
    if (i >= n_live) break;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
#pragma unroll
    for (int pass = 0; pass < kFwdPasses; ++pass) {
        const int l = level_of(group, pass, gp.n_levels);
        if (l < 0) continue;
        const Corners c = corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
        const uint32_t* t = table + gp.offset[l];
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t[c.idx[k]];
        float w[8];
        corner_weights(c.f, smooth, w);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a0 = fmaf(w[k], T16::lo(v[k]), a0);
            a1 = fmaf(w[k], T16::hi(v[k]), a1);
        }
        feat[(int64_t)l * n + i] = T16::pack(a0, a1);
    }
    }
}


// ---- forward encode, second generation: run de-duplication + rotating level groups -----------------------------------------
// The gathers of this kernel are bound by the L1's request rate (one cache-line look-up per active lane and cycle; the
// tables sit in L2 / Infinity Cache), not by bandwidth.  Two things reduce what the slowest XCD has to issue:
//  (1) run de-duplication: consecutive samples of a ray (training batches) and equal-rank samples of neighbouring pixels
//      (eval frames) fall into the SAME cell at the coarse levels, i.e. neighbouring lanes gather the same eight entries.
//      One DPP compare per level finds the runs; only run heads issue the gathers, the others fetch the head's packed
//      dwords with ds_bpermute (LDS crossbar, ~2 cycles per 64 lanes instead of 64 L1 look-ups).  Interpolation stays
//      per lane: features are bit-identical.  A wave whose lanes share little (> kShareMaxHeads heads) gathers as before.
//  (2) de-duplication makes the level groups unequal (a coarse level costs a fraction of a fine one), and a fixed
//      group <-> XCD pinning would leave the kernel as long as its most expensive group.  The pinning therefore ROTATES:
//      the chunks of a launch are cut into eight phases, and in phase p XCD x serves group (x + p) % 8.  Every XCD serves
//      every group for an eighth of the samples -- equal work whatever the levels cost -- while its L2 still holds two
//      tables at a time (refilled from the Infinity Cache at each of the seven phase changes).  (A first attempt handed
//      out (group, chunk) tickets through one device counter per group: same-address atomics retire at ~105 ns each on
//      gfx950, 4096 tickets per counter made the kernel three times SLOWER -- tools/exp/fwd_v2.py, profiles/README.md.)
constexpr int kShareMaxHeads = 56;

template <typename T16>
__global__ __launch_bounds__(256) void hashgrid_fwd_v2_kernel(GridParams gp, const float* __restrict__ x01,
                                                              const uint32_t* __restrict__ table,
                                                              uint32_t* __restrict__ feat, int64_t n,
                                                              const int64_t* __restrict__ n_dev) {
    const int64_t n_live = live_count(n, n_dev);                 // (n stays the level stride)
    const int64_t nchunks_live = (n_live + 255) >> 8;
    const int64_t nchunks_grid = (int64_t)(gridDim.x >> 3);
    const int xcd = (int)(blockIdx.x & 7);
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (lane == 63u) ? ~0ull : ((2ull << lane) - 1ull);      // lanes <= mine
    // chunk-stride loop: a capacity-sized launch (n >> n_live) is capped at nchunks_grid chunks per XCD
    for (int64_t chunk = (int64_t)(blockIdx.x >> 3); chunk < nchunks_live; chunk += nchunks_grid) {
        // phase of a chunk: its position within the pass of the grid over the chunks (any function of the chunk alone keeps
        // the eight workgroups of a chunk on eight different groups)
        const int64_t in_pass = chunk % nchunks_grid, pass_len = nchunks_live < nchunks_grid ? nchunks_live : nchunks_grid;
        const int phase = (int)((in_pass * 8) / pass_len) & 7;
        const int g = (xcd + phase) & 7;
        const int64_t i = chunk * 256 + threadIdx.x;
        const bool live = i < n_live;
        const int64_t ii = live ? i : n_live - 1;                // (idle lanes of the last chunk repeat its last sample)
        const float x = x01[3 * ii], y = x01[3 * ii + 1], z = x01[3 * ii + 2];
        // Both levels of the group are set up first, then all their gathers are issued, then shared and interpolated: the
        // (L1-hit) coarse and the (L2-served) fine level stay in flight together.
        int lv[2];
        Corners c[2];
        bool head[2], share[2];
        uint32_t src[2];
        uint32_t v[2][8];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            lv[pass] = level_of(g, pass, gp.n_levels);
            head[pass] = lv[pass] >= 0; share[pass] = false; src[pass] = lane;
            if (lv[pass] < 0) continue;
            const int l = lv[pass];
            c[pass] = corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
            {
                // lane - 1's cell through DPP (wave_shr:1; lane 0 keeps the `old` operand)
                const uint32_t px = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)c[pass].cell[0], 0x138, 0xf, 0xf, false);
                const uint32_t py = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)c[pass].cell[1], 0x138, 0xf, 0xf, false);
                const uint32_t pz = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)c[pass].cell[2], 0x138, 0xf, 0xf, false);
                const bool same = lane != 0u && px == c[pass].cell[0] && py == c[pass].cell[1] && pz == c[pass].cell[2];
                const unsigned long long heads = __ballot(!same);
                share[pass] = __popcll(heads) <= kShareMaxHeads;      // wave-uniform
                if (share[pass]) {
                    head[pass] = !same;
                    src[pass] = 63u - (uint32_t)__clzll((long long)(heads & below));     // the head of my run
                }
            }
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[pass][k] = 0u;
            if (head[pass]) {
                const uint32_t* t = table + gp.offset[lv[pass]];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[pass][k] = t[c[pass].idx[k]];
            }
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (lv[pass] < 0) continue;
            if (share[pass]) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[pass][k] = (uint32_t)__shfl((int)v[pass][k], (int)src[pass]);
            }
            float w[8];
            corner_weights(c[pass].f, smooth, w);
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0 = fmaf(w[k], T16::lo(v[pass][k]), a0);
                a1 = fmaf(w[k], T16::hi(v[pass][k]), a1);
            }
            if (live) feat[(int64_t)lv[pass] * n + i] = T16::pack(a0, a1);
        }
    }
}

// Two tables with the SAME grid geometry (PeRF's density and colour fields, ngp_nerf.py:96-134) evaluated at the
// same points: corner indices and weights are computed once, 16 gathers are in flight per (sample, level).
template <typename T16>
__global__ __launch_bounds__(256) void hashgrid_fwd2_kernel(GridParams gp, const float* __restrict__ x01,
                                                            const uint32_t* __restrict__ table_a,
                                                            const uint32_t* __restrict__ table_b,
                                                            uint32_t* __restrict__ feat_a, uint32_t* __restrict__ feat_b,
                                                            int64_t n) {
    const int group = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
#pragma unroll
    for (int pass = 0; pass < kFwdPasses; ++pass) {
        const int l = level_of(group, pass, gp.n_levels);
        if (l < 0) continue;
        const Corners c = corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
        const uint32_t* ta = table_a + gp.offset[l];
        const uint32_t* tb = table_b + gp.offset[l];
        uint32_t va[8], vb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { va[k] = ta[c.idx[k]]; vb[k] = tb[c.idx[k]]; }
        float w[8];
        corner_weights(c.f, smooth, w);
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a0 = fmaf(w[k], T16::lo(va[k]), a0); a1 = fmaf(w[k], T16::hi(va[k]), a1);
            b0 = fmaf(w[k], T16::lo(vb[k]), b0); b1 = fmaf(w[k], T16::hi(vb[k]), b1);
        }
        feat_a[(int64_t)l * n + i] = T16::pack(a0, a1);
        feat_b[(int64_t)l * n + i] = T16::pack(b0, b1);
    }
}

// corner table indices (absolute entry index, level offset included) of every (level, sample): the integer half of
// the encoding, exported so that arbitrarily-often differentiable compositions can be built on top of it
__global__ __launch_bounds__(256) void hashgrid_corners_kernel(GridParams gp, const float* __restrict__ x01,
                                                               int32_t* __restrict__ idx_out, int64_t n) {
    const int l = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || l >= gp.n_levels) return;
    const Corners c = corners_of(x01[3 * i], x01[3 * i + 1], x01[3 * i + 2], gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
    int32_t* o = idx_out + ((int64_t)l * n + i) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (int32_t)(gp.offset[l] + c.idx[k]);
}

__global__ __launch_bounds__(256) void hashgrid_fwd_f32_kernel(GridParams gp, const float* __restrict__ x01,
                                                               const float2* __restrict__ table,
                                                               float2* __restrict__ feat, int64_t n) {
    const int group = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
#pragma unroll
    for (int pass = 0; pass < kFwdPasses; ++pass) {
        const int l = level_of(group, pass, gp.n_levels);
        if (l < 0) continue;
        const Corners c = corners_of(x, y, z, gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
        const float2* t = table + gp.offset[l];
        float w[8];
        corner_weights(c.f, smooth, w);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float2 v = t[c.idx[k]];
            a0 = fmaf(w[k], v.x, a0);
            a1 = fmaf(w[k], v.y, a1);
        }
        feat[(int64_t)l * n + i] = make_float2(a0, a1);
    }
}

// Parameter gradient.  gfx950 global fp32 atomics retire at a flat ~2e10/s whatever their scope or
// address distribution (tools/exp/atomics.hip), i.e. ~13 ms for the 2.7e8 corner updates of a
// 1 M-sample batch -- so the scatter is turned inside out: the fp32 gradient table (26.6 MB for
// L16/T18) is cut into 128 KiB tiles of 16384 entries and each tile is OWNED by one workgroup that
// keeps it in LDS (the multi-tile levels' 192 + 12 tiles plus the replicated coarse ones <= 256 CUs: the whole
// table is resident in the chip's aggregate LDS).  Every owner walks all samples of its level and applies only the
// corner updates that fall in its tile with LDS atomics -- multi-tile levels through a 4-byte tile code per sample
// that a pre-pass computes once for all owners (bwd_stream_codes), single-tile levels from positions (bwd_stream);
// at the end the tile is written back with plain coalesced stores.  No global atomics, no zero-fill pass, every
// table entry written exactly once.
constexpr int kTileEntries = 16384;
constexpr int kBwdThreads = 1024;
constexpr int kMaxReplicas = 16;
constexpr int kMaxWork = 512;              // workgroups the XCD-aware placement table can hold
constexpr int kXcds = 8;

// Load balance: a hashed level has 16 tiles, each receiving 1/16 of the level's 8 corner updates per
// sample; a coarse dense level has only 1..8 tiles receiving the same total.  Coarse tiles are therefore
// REPLICATED (R_l copies, each streaming 1/R_l of the samples) so that every workgroup takes about as long as
// a hashed-tile owner; replicas are summed by a small second kernel.  L16/T18, fixed-point mode:
// 8 + 8 + 4x3 + 8x2 + 12x16 = 236 workgroups.
struct TileParams {
    int32_t tiles_of[PERF_MAX_LEVELS];     // tiles per level
    int32_t replicas_of[PERF_MAX_LEVELS];  // replicas per tile
    int64_t ws_off[PERF_MAX_LEVELS];       // float2 offset of the level's replica slabs in the workspace
    int32_t accumulate;
    int32_t code_slot[PERF_MAX_LEVELS];    // >=0: the level's tile codes are codes[slot][n_pad] (see tile_codes_kernel)
    int64_t n_pad;
    // XCD-aware placement: workgroup b runs work[b] = level << 16 | tile << 8 | replica (0xffffffff: idle).  The
    // dispatcher deals workgroups round-robin over the 8 XCDs, so b % 8 is the XCD: the owners of one level are put on
    // the same XCD -- they stream the same codes and gather the same position / gradient lines at about the same
    // time, which then hit that XCD's L2 instead of crossing the fabric 16 times.  use_work == 0: plain level order.
    int32_t use_work;
    int32_t raw_out;                       // fixed point: the gradient table receives the int32 field pairs themselves (see perf_hashgrid_bwd)
    uint32_t atomic_levels;                // bit l: level l is too large for LDS owners (see hashgrid_bwd_atomic_kernel)
    uint32_t bitmap_levels;                // bit l: hashed level of 256..kBitmapMaxTiles tiles whose owners read per-tile bitmaps (tile_bitmap_kernel)
    int32_t bm_row[PERF_MAX_LEVELS];       // first bitmap row (= tile 0) of such a level, its index among them in bm_idx
    int32_t bm_idx[PERF_MAX_LEVELS];
    int64_t bm_row_words;                  // 32-bit words per bitmap row (bm_samples / 32 per pre-pass block)
    int64_t bm_blocks;                     // pre-pass blocks = escape words per bitmap level
    int32_t bm_samples;                    // samples per pre-pass block (256..1024: tiles x bm_samples / 8 bytes of LDS staging <= 64 KiB)
    int32_t run_merge;                     // single-tile dense levels: a thread sums runs of samples in one cell in registers
    uint32_t work[kMaxWork];
};

constexpr int kQueueCap = 448;             // per-wave match queue (entries): <128 left over + 4 x 64 new + 64 re-queued
constexpr int64_t kMaxCodedSamples = (int64_t)1 << 28;

constexpr int kBitmapMinDenseTiles = 32;
constexpr int kBitmapMaxTiles = 2048;      // (2048 x 32 bytes of LDS staging per pre-pass block; log2_hashmap_size <= 25)

// bitmap_tiles > 0: hashed levels of 256..bitmap_tiles tiles get LDS owners fed by per-tile bitmaps instead of the global
// atomics (the caller checks that the workspace holds the bitmaps and plans again with 0 otherwise)
static void plan_tiles(const GridParams& gp, bool fixed, TileParams* tp, int* n_blocks, int64_t* ws_entries, int bitmap_tiles = 0,
                       bool no_replicas = false) {
    int nb = 0;
    int64_t ws = 0;
    tp->atomic_levels = 0u;
    tp->bitmap_levels = 0u;
    const int rs[3] = {8, 3, 2};       // replicas of dense levels of 1 / <= 4 / <= 16 tiles
    bool large_grid = false;        // some level takes bitmap owners
    for (int l = 0; l < gp.n_levels && bitmap_tiles > 0; ++l) {
        int nt = (int)((gp.size[l] + kTileEntries - 1) / kTileEntries);
        if (!gp.hashed[l]) { int p2 = 1; while (p2 < nt) p2 <<= 1; nt = p2; }
        if (nt <= bitmap_tiles && (gp.hashed[l] ? (nt > 255 && gp.res[l] + 2u < (uint32_t)kTileEntries) : nt >= kBitmapMinDenseTiles)) large_grid = true;
    }
    for (int l = 0; l < PERF_MAX_LEVELS; ++l) {
        tp->tiles_of[l] = 0; tp->replicas_of[l] = 1; tp->ws_off[l] = 0;
        if (l >= gp.n_levels) continue;
        int nt = (int)((gp.size[l] + kTileEntries - 1) / kTileEntries);
        if (!gp.hashed[l]) { int p = 1; while (p < nt) p <<= 1; nt = p; }     // dense ownership is a bit field of the index
        // A level of more than 255 (hashed) / 64 (dense) tiles = 4 M / 1 M entries would need that many owners, each
        // walking every sample: beyond that the plain global-atomics scatter is cheaper (log2_hashmap_size >= 22).
        // (dense levels already from 32 tiles: a sample touches 4-5 of them, and code-streaming owners that each test every
        //  sample are the long pole of a 20-level grid -- 1.6 ms per 1 M random points at 64 tiles)
        if (nt > (gp.hashed[l] ? 255 : 64) || (!gp.hashed[l] && nt >= kBitmapMinDenseTiles && nt <= bitmap_tiles)) {
            if (nt <= bitmap_tiles && (!gp.hashed[l] || gp.res[l] + 2u < (uint32_t)kTileEntries)) {
                tp->bitmap_levels |= 1u << l;
                tp->tiles_of[l] = nt;
                nb += nt;
            } else {
                tp->atomic_levels |= 1u << l;
            }
            continue;
        }
        // replication factors from measured per-workgroup times (tools/exp/bwd_block_times.py, 1 M samples, fixed):
        // hashed tile (coded) 0.36-0.39 ms; dense tile streaming ALL samples: 1 tile 2.4 ms, 4 tiles 0.88 ms, 8 tiles 0.68 ms
        // (fp32 mode is bound by ds_add_f32 lane-serialisation instead: equal corner-update counts, r = 16 / nt)
        int r = 1;
        if (!gp.hashed[l]) r = fixed ? ((nt == 1) ? rs[0] : (nt <= 4 ? rs[1] : (nt <= 16 ? rs[2] : 1))) : kMaxReplicas / nt;
        // A grid with bitmap levels launches thousands of short owners anyway: its few-tile dense levels -- where every sample
        // is an entry of most owners -- get the replicas that keep them from being the kernel's long pole (measured on a
        // 20-level grid: 2 tiles x 3 replicas 1.1 ms per workgroup, 8 x 2 0.66 ms, against 0.07-0.3 ms everywhere else)
        if (!gp.hashed[l] && fixed && bitmap_tiles > 0 && large_grid) r = nt == 1 ? 8 : (nt == 2 ? 8 : (nt == 4 ? 6 : (nt == 8 ? 4 : 2)));
        if (r < 1 || no_replicas) r = 1;
        if (r > kMaxReplicas) r = kMaxReplicas;
        tp->tiles_of[l] = nt; tp->replicas_of[l] = r;
        if (r > 1) { tp->ws_off[l] = ws; ws += (int64_t)r * gp.size[l]; }
        nb += nt * r;
    }
    *n_blocks = nb; *ws_entries = ws;
    // ---- XCD-aware placement (a speed assumption only: results do not depend on it)
    tp->use_work = 0;
    if (nb > kMaxWork || tp->bitmap_levels) return;       // (the table packs the tile in 8 bits)
    uint32_t lists[kXcds][kMaxWork];                        // 16 KiB of stack: the planner is re-entrant
    int len[kXcds] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto least = [&]() { int x = 0; for (int i = 1; i < kXcds; ++i) if (len[i] < len[x]) x = i; return x; };
    const int per_xcd = (nb + kXcds - 1) / kXcds;
    for (int pass = 0; pass < 2; ++pass)                    // multi-tile levels first (they are the ones that share), then the rest
        for (int l = gp.n_levels - 1; l >= 0; --l) {
            const int nt = tp->tiles_of[l], R = tp->replicas_of[l];
            if ((nt > 1) != (pass == 0)) continue;
            int x = least();
            for (int t = 0; t < nt; ++t)
                for (int r = 0; r < R; ++r) {
                    if (len[x] >= per_xcd) x = least();     // the level spills over to the emptiest XCD
                    lists[x][len[x]++] = ((uint32_t)l << 16) | ((uint32_t)t << 8) | (uint32_t)r;
                }
        }
    int longest = 0;
    for (int x = 0; x < kXcds; ++x) longest = len[x] > longest ? len[x] : longest;
    if (longest * kXcds > kMaxWork) return;
    for (int slot = 0; slot < longest; ++slot)
        for (int x = 0; x < kXcds; ++x) tp->work[slot * kXcds + x] = slot < len[x] ? lists[x][slot] : 0xffffffffu;
    tp->use_work = 1;
    *n_blocks = longest * kXcds;
}

// Tile ownership.  Hashed levels: tile = idx / 16384 (the hash already spreads cells uniformly).  Dense levels:
// ownership is INTERLEAVED in chunks of 32 entries over a power-of-two number of tiles (chunk c = idx/32 belongs
// to tile c % n_tiles, local slot (c / n_tiles)*32 + idx%32) -- contiguous slabs would be spatial slabs, and samples concentrate near the
// camera / the surfaces, which overloads a few owners.
constexpr uint32_t kChunk = 32;

struct BwdCtx {
    float scale, to_fixed;
    uint32_t res, r2, size, mask, n_tiles, t;
    uint32_t tile_shift;        // dense levels: log2(n_tiles)
    bool smooth;
};

// (y,z)-combination updates of one sample in a hashed tile: both x-corners of every combination in `cm`
// (bit c: by = c & 1, bz = c >> 1).  Requires gx + 1 < kTileEntries (the tile then depends on (y,z) only).
template <bool FIXED>
__device__ __forceinline__ void apply_pairs(const BwdCtx& cx, float* lds_tile, const float2 g, const uint32_t gx, float fx,
                                            float fy, float fz, const uint32_t ay0, const uint32_t az0, uint32_t cm) {
    unsigned long long* lds64 = reinterpret_cast<unsigned long long*>(lds_tile);
    if (cx.smooth) { fx = fx * fx * (3.0f - 2.0f * fx); fy = fy * fy * (3.0f - 2.0f * fy); fz = fz * fz * (3.0f - 2.0f * fz); }
    const uint32_t tlo = cx.t * (uint32_t)kTileEntries;
    const float sx = g.x * cx.to_fixed, sy = g.y * cx.to_fixed;
    while (cm) {
        const int c = __ffs(cm) - 1;
        cm &= cm - 1u;
        const int by = c & 1, bz = c >> 1;
        const uint32_t h = (by ? ay0 + kPrimeY : ay0) ^ (bz ? az0 + kPrimeZ : az0);
        const float wyz = (by ? fy : 1.0f - fy) * (bz ? fz : 1.0f - fz);
        const uint32_t a0 = ((gx ^ h) & cx.mask) - tlo, a1 = (((gx + 1u) ^ h) & cx.mask) - tlo;
        const float w0 = (1.0f - fx) * wyz, w1 = fx * wyz;
        if (FIXED) {
            const long long v0 = ((long long)__float2int_rn(w0 * sy) << 32) + (long long)__float2int_rn(w0 * sx);
            const long long v1 = ((long long)__float2int_rn(w1 * sy) << 32) + (long long)__float2int_rn(w1 * sx);
            atomicAdd(&lds64[a0], (unsigned long long)v0);
            atomicAdd(&lds64[a1], (unsigned long long)v1);
        } else {
            unsafeAtomicAdd(&lds_tile[2 * a0], w0 * g.x); unsafeAtomicAdd(&lds_tile[2 * a0 + 1], w0 * g.y);
            unsafeAtomicAdd(&lds_tile[2 * a1], w1 * g.x); unsafeAtomicAdd(&lds_tile[2 * a1 + 1], w1 * g.y);
        }
    }
}

// Dense-level counterpart (interleaved 32-entry chunks): the two x-corners of a (y,z) combination are tested
// separately -- they part at a chunk boundary -- and a combination that would wrap past the end of the level
// (position far outside the unit cube; such a level takes the generic owners, see tile_codes_kernel) is skipped.
template <bool FIXED>
__device__ __forceinline__ void apply_pairs_dense(const BwdCtx& cx, float* lds_tile, const float2 g, const uint32_t gx, float fx,
                                                  float fy, float fz, const uint32_t ay0, const uint32_t az0, uint32_t cm) {
    unsigned long long* lds64 = reinterpret_cast<unsigned long long*>(lds_tile);
    if (cx.smooth) { fx = fx * fx * (3.0f - 2.0f * fx); fy = fy * fy * (3.0f - 2.0f * fy); fz = fz * fz * (3.0f - 2.0f * fz); }
    const float sx = g.x * cx.to_fixed, sy = g.y * cx.to_fixed;
    const uint32_t tmask = cx.n_tiles - 1u;
    while (cm) {
        const int c = __ffs(cm) - 1;
        cm &= cm - 1u;
        const int by = c & 1, bz = c >> 1;
        const uint32_t i0 = gx + (by ? ay0 + cx.res : ay0) + (bz ? az0 + cx.r2 : az0), i1 = i0 + 1u;
        if (i1 >= cx.size || i1 == 0u) continue;
        const float wy = by ? fy : 1.0f - fy, wz = bz ? fz : 1.0f - fz;
        const float w0 = ((1.0f - fx) * wy) * wz, w1 = (fx * wy) * wz;      // association of the generic owners
        const uint32_t c0 = i0 / kChunk, c1 = i1 / kChunk;
        const uint32_t a0 = (c0 >> cx.tile_shift) * kChunk + (i0 % kChunk), a1 = (c1 >> cx.tile_shift) * kChunk + (i1 % kChunk);
        if ((c0 & tmask) == cx.t) {
            if (FIXED) atomicAdd(&lds64[a0], (unsigned long long)(((long long)__float2int_rn(w0 * sy) << 32) + (long long)__float2int_rn(w0 * sx)));
            else { unsafeAtomicAdd(&lds_tile[2 * a0], w0 * g.x); unsafeAtomicAdd(&lds_tile[2 * a0 + 1], w0 * g.y); }
        }
        if ((c1 & tmask) == cx.t) {
            if (FIXED) atomicAdd(&lds64[a1], (unsigned long long)(((long long)__float2int_rn(w1 * sy) << 32) + (long long)__float2int_rn(w1 * sx)));
            else { unsafeAtomicAdd(&lds_tile[2 * a1], w1 * g.x); unsafeAtomicAdd(&lds_tile[2 * a1 + 1], w1 * g.y); }
        }
    }
}

// Scale of the fixed-point gradient fields of level l: one unit = 2^-sh.
__device__ __forceinline__ int fixed_point_shift(const float am, const int64_t n_live, const uint32_t size,
                                                 const int32_t* __restrict__ hr_state, const int l) {
    int e = 0;
    if (am > 0.f) (void)frexpf(am, &e);                         // am < 2^e
    if (e < -80) e = -80;                                        // (vanishing gradients: keep 2^sh finite)
    // Headroom of the fixed-point fields: an entry of level l sums 8 n / size_l contributions on average (1,700 at
    // the coarsest level of a 1 M-sample batch, 32 at a hashed one); 64x that average before the overflow flag is
    // raised, never less than 2^12, never more than 2^24 (which still leaves 2^-7 of the largest contribution as the
    // unit).  Derived from the LIVE sample count, so a capacity-sized launch keeps the resolution of an exact one.
    const unsigned long long fan = (8ull * (unsigned long long)n_live + size - 1ull) / size;     // ceil(8 n / size)
    int h = (fan <= 1ull ? 0 : 64 - __clzll((long long)(fan - 1ull))) + 6;                      // ceil(log2(fan)) + 6
    if (hr_state) {
        // Closed loop (caller-owned state, see perf_hashgrid_bwd): the static guess is corrected by what the fields of
        // the PREVIOUS calls really reached -- entries near a panorama's common ray origin sum 30x the average number
        // of contributions, hashed levels far fewer than the guess allows.  Starts 3 bits on the safe side.
        h += hr_state[l] + kHeadroomStartBias;
        h = h < 4 ? 4 : (h > 28 ? 28 : h);
    } else {
        h = h < 12 ? 12 : (h > 24 ? 24 : h);
    }
    return 31 - h - e;
}

// Headroom feedback: keep the largest field of a level between 2^21 and 2^25 units.  Above: add the excess bits at once
// (+1); below: give one bit back per call.  fm = the largest |field| the level's FINAL sums reached in the previous call
// (all replicas -- and, under data parallelism, all ranks -- added up), so that every partition of a batch follows the
// same sequence of units.  Deterministic: the state is a function of the call history only.
// The top of the band sits 16x below the level at which the overflow flag is raised (2^29) and the step gate drops the
// step: a soak of 25 episodes with the band at [2^23, 2^27] (4x) lost 8 of 112,500 steps to flags that were not
// overflows -- the colour table's largest sum quadrupling from one batch to the next (tools/soak_episodes.py).
constexpr int kHeadroomTopBit = 25, kHeadroomLowBit = 21;
constexpr int kLaggedMinHeadroom = 12;     // (dp_units_kernel, lagged units)
constexpr int kLaggedMaxFinerBits = 2;
__device__ __forceinline__ int headroom_feedback(int adj, int fm) {
    if (fm >= (1 << kHeadroomTopBit)) adj += (32 - __clz(fm)) - kHeadroomTopBit + 1;
    else if (fm < (1 << kHeadroomLowBit) && adj > -24) adj -= 1;
    return adj;
}

// one sample's contribution to the tile this workgroup owns
template <bool FIXED, bool HASHED>
__device__ __forceinline__ void bwd_apply(const BwdCtx& cx, float* lds_tile, const float2 g, const float x, const float y,
                                          const float z) {
    unsigned long long* lds64 = reinterpret_cast<unsigned long long*>(lds_tile);
    const float px = grid_pos(x, cx.scale), py = grid_pos(y, cx.scale), pz = grid_pos(z, cx.scale);
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    float fx = px - flx, fy = py - fly, fz = pz - flz;
    const uint32_t gx = (uint32_t)(int32_t)flx, gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
    uint32_t ay[2], az[2];
    uint32_t match = 0;     // bit k: corner k (bit0 = x, bit1 = y, bit2 = z) is owned by this tile
    if (HASHED) {
        ay[0] = gy * kPrimeY; ay[1] = ay[0] + kPrimeY; az[0] = gz * kPrimeZ; az[1] = az[0] + kPrimeZ;
        if (gx < (uint32_t)(kTileEntries - 1)) {
            // The tile of a hashed corner depends on (y,z) only while gx+1 < 2^14, so the two x-corners of a (y,z)
            // combination are tested and applied together: 4 tests and at most 4 (usually 0-1) pair updates.
            uint32_t cm = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                cm |= (((((ay[c & 1] ^ az[c >> 1]) & cx.mask) / (uint32_t)kTileEntries) == cx.t) ? 1u : 0u) << c;
            if (cm == 0) return;
            apply_pairs<FIXED>(cx, lds_tile, g, gx, fx, fy, fz, ay[0], az[0], cm);
            return;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t idx = ((gx + (uint32_t)(k & 1)) ^ ay[(k >> 1) & 1] ^ az[k >> 2]) & cx.mask;
            match |= ((idx / (uint32_t)kTileEntries) == cx.t ? 1u : 0u) << k;
        }
    } else {
        ay[0] = gy * cx.res; ay[1] = ay[0] + cx.res; az[0] = gz * cx.r2; az[1] = az[0] + cx.r2;
        if (cx.n_tiles == 1) match = 0xffu;
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t idx = (gx + (uint32_t)(k & 1)) + ay[(k >> 1) & 1] + az[k >> 2];
                if (idx >= cx.size) idx = idx % cx.size;
                match |= (((idx / kChunk) & (cx.n_tiles - 1u)) == cx.t ? 1u : 0u) << k;
            }
        }
    }
    if (match == 0) return;
    if (cx.smooth) { fx = fx * fx * (3.0f - 2.0f * fx); fy = fy * fy * (3.0f - 2.0f * fy); fz = fz * fz * (3.0f - 2.0f * fz); }
    while (match) {
        const int k = __ffs(match) - 1;
        match &= match - 1u;
        const int bx = k & 1, by = (k >> 1) & 1, bz = k >> 2;
        const uint32_t cy = by ? ay[1] : ay[0], cz = bz ? az[1] : az[0];
        uint32_t a;
        if (HASHED) a = (((gx + (uint32_t)bx) ^ cy ^ cz) & cx.mask) - cx.t * (uint32_t)kTileEntries;
        else {
            uint32_t idx = (gx + (uint32_t)bx) + cy + cz;
            if (idx >= cx.size) idx = idx % cx.size;
            a = ((idx / kChunk) >> cx.tile_shift) * kChunk + (idx % kChunk);
        }
        const float w = ((bx ? fx : 1.0f - fx) * (by ? fy : 1.0f - fy)) * (bz ? fz : 1.0f - fz);
        if (FIXED) {
            const long long lo = (long long)__float2int_rn(w * g.x * cx.to_fixed);
            const long long hi = (long long)__float2int_rn(w * g.y * cx.to_fixed);
            atomicAdd(&lds64[a], (unsigned long long)((hi << 32) + lo));
        } else {
            unsafeAtomicAdd(&lds_tile[2 * a], w * g.x);
            unsafeAtomicAdd(&lds_tile[2 * a + 1], w * g.y);
        }
    }
}


// ---- hashed owners, coded variant ------------------------------------------------------------------------------
// The owner test of a hashed level depends on (y,z) only, and it is the same for the 16 owners of the level, so a
// pre-pass (tile_codes_kernel) evaluates it ONCE per (sample, level): one byte per (y,z) combination = the tile the
// combination's two x-corners fall in.  An owner then streams 4-byte codes instead of positions + gradients, keeps
// the samples that name its tile in a wave-private LDS queue (one ballot per sample) and applies them 64 at a time
// at full lane occupancy; positions and gradients are gathered for queued samples only (about 22 % of them), and the
// gather of one batch is issued one drain ahead of its use.
constexpr int kCodeSamplesPerBlock = 256;
// (Round 3 measured two ways of sparing the owners their tests, both bit-identical, neither kept -- tools/exp/bwd_sort.py,
//  profiles/r03_bwd_sorted_variants.json, r03_bwd_bitmap_variants.json, the code is in the history:
//  * counting-sorted per-tile record lists, so that an owner walks only its own records: the owners became GATHER bound
//    -- 0.37-0.43 ms per workgroup against 0.32; 0.164 ms with the gathers stubbed out -- because a list that is not in
//    sample order loses the 2-3 queued samples per 128-byte line that neighbouring lanes share, and the sort pass took
//    0.65 ms;
//  * one bit per (tile, sample) instead of the codes, expanded to queue entries with a wave scan: owners 0.325 ->
//    0.277 ms per workgroup, but the kernel ends with its burstiest level (0.31 ms) and the pre-pass grew by 0.085 ms.
//  What an owner costs is the drain: gather 20 B per queued sample, ~95 instructions and two 64-bit LDS atomics per
//  combination; tests, gathers and LDS are within 2x of each other, so removing one of them moves little.)

// The tile code of sample (x, y, z) at level l (see tile_codes_kernel); bad: the premise of the code does not hold.
__device__ __forceinline__ uint32_t tile_code_of(const GridParams& gp, const TileParams& tp, int l, float x, float y, float z, bool& bad) {
    const float py = grid_pos(y, gp.scale[l]), pz = grid_pos(z, gp.scale[l]);
    const uint32_t gy = (uint32_t)(int32_t)floorf(py), gz = (uint32_t)(int32_t)floorf(pz);
    const uint32_t gx = (uint32_t)(int32_t)floorf(grid_pos(x, gp.scale[l]));
    uint32_t code = 0u;
    if (gp.hashed[l]) {
        const uint32_t ay0 = gy * kPrimeY, ay1 = ay0 + kPrimeY, az0 = gz * kPrimeZ, az1 = az0 + kPrimeZ;
        const uint32_t m = gp.size[l] - 1u;
        code = (((ay0 ^ az0) & m) / (uint32_t)kTileEntries) | ((((ay1 ^ az0) & m) / (uint32_t)kTileEntries) << 8) |
               ((((ay0 ^ az1) & m) / (uint32_t)kTileEntries) << 16) | ((((ay1 ^ az1) & m) / (uint32_t)kTileEntries) << 24);
        // a position so far outside the unit cube that its x-corners leave the first 16384 columns breaks
        // "(y,z) decides the tile"
        bad = gx >= (uint32_t)(kTileEntries - 1);
    } else {
        // dense level: byte = tile of the x0 corner, bit 7 set when the x1 corner sits in the next chunk (= next
        // tile); 0x7f (no tile) when the pair would wrap past the end of the level
        const uint32_t res = gp.res[l], r2 = res * res, tmask = (uint32_t)tp.tiles_of[l] - 1u;
        bad = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t i0 = gx + (gy + (uint32_t)(c & 1)) * res + (gz + (uint32_t)(c >> 1)) * r2, i1 = i0 + 1u;
            uint32_t b = 0x7fu;
            if (i1 >= gp.size[l] || i1 == 0u) bad = true;
            else b = ((i0 / kChunk) & tmask) | ((((i1 / kChunk) & tmask) != ((i0 / kChunk) & tmask)) ? 0x80u : 0u);
            code |= b << (8 * c);
        }
    }
    return code;
}

__global__ __launch_bounds__(256) void tile_codes_kernel(GridParams gp, TileParams tp, const float* __restrict__ x01,
                                                         const float2* __restrict__ dfeat, uint32_t* __restrict__ codes,
                                                         uint32_t* __restrict__ escape, int64_t n,
                                                         const int64_t* __restrict__ n_dev) {
    const int64_t n_live = live_count(n, n_dev);            // n: capacity = stride of dfeat / codes; n_live: samples present
    __shared__ uint32_t esc_block;
    if (threadIdx.x == 0) esc_block = 0u;
    __syncthreads();
    uint32_t esc = 0u;                  // bit l: level l must take the generic owners (see below)
    const int64_t i0 = (int64_t)blockIdx.x * kCodeSamplesPerBlock;
    for (int64_t i = i0 + threadIdx.x; i < n_live && i < i0 + kCodeSamplesPerBlock; i += 256) {
        const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
        for (int l = 0; l < gp.n_levels; ++l) {
            const int slot = tp.code_slot[l];
            if (slot < 0) continue;
            bool bad;       // the premise of the code does not hold for this sample
            const uint32_t code = tile_code_of(gp, tp, l, x, y, z, bad);
            codes[(int64_t)slot * tp.n_pad + i] = code;
            if (bad && gp.hashed[l]) {      // harmless without gradient; with gradient the level's owners take the generic path
                                            // (dense: the owners apply such a sample corner by corner, see bwd_stream_codes)
                const float2 g = dfeat[(int64_t)l * n + i];
                if (!(g.x == 0.f && g.y == 0.f)) esc |= 1u << l;
            }
        }
    }
    if (esc) atomicOr(&esc_block, esc);
    __syncthreads();
    if (threadIdx.x == 0) escape[blockIdx.x] = esc_block;       // every word is written: no zero-fill needed
}

// The same codes, four consecutive samples per thread: 48 bytes of positions in three 16-byte loads and one 16-byte store per
// level instead of four 4-byte ones (the byte-code kernel spent most of a wave's life queueing stores: 16 per sample).  A wave
// covers the 256 samples of one escape word.  (The kernel above serves workspaces whose code rows are not 16-byte aligned.)
__global__ __launch_bounds__(256) void tile_codes4_kernel(GridParams gp, TileParams tp, const float* __restrict__ x01,
                                                          const float2* __restrict__ dfeat, uint32_t* __restrict__ codes,
                                                          uint32_t* __restrict__ escape, int64_t n, int64_t n_words,
                                                          const int64_t* __restrict__ n_dev) {
    const int64_t n_live = live_count(n, n_dev);
    __shared__ uint32_t esc_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) esc_w[wave] = 0u;
    __builtin_amdgcn_wave_barrier();
    const int64_t word = (int64_t)blockIdx.x * 4 + wave;
    if (word >= n_words) return;                            // (whole waves leave)
    const int64_t i0 = word * kCodeSamplesPerBlock + 4 * lane;
    uint32_t esc = 0u;
    if (i0 < n_live) {
        const int cnt = (n_live - i0) < 4 ? (int)(n_live - i0) : 4;
        float p[12];
        if (cnt == 4 && (reinterpret_cast<uintptr_t>(x01) & 15) == 0) {
            const float4 a = reinterpret_cast<const float4*>(x01 + 3 * i0)[0], b = reinterpret_cast<const float4*>(x01 + 3 * i0)[1],
                         c = reinterpret_cast<const float4*>(x01 + 3 * i0)[2];
            p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; p[4] = b.x; p[5] = b.y; p[6] = b.z; p[7] = b.w;
            p[8] = c.x; p[9] = c.y; p[10] = c.z; p[11] = c.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t i = i0 + (k < cnt ? k : cnt - 1);
                p[3 * k] = x01[3 * i]; p[3 * k + 1] = x01[3 * i + 1]; p[3 * k + 2] = x01[3 * i + 2];
            }
        }
        for (int l = 0; l < gp.n_levels; ++l) {
            const int slot = tp.code_slot[l];
            if (slot < 0) continue;
            uint32_t code[4];
            bool bad[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) code[k] = tile_code_of(gp, tp, l, p[3 * k], p[3 * k + 1], p[3 * k + 2], bad[k]);
            uint32_t* row = codes + (int64_t)slot * tp.n_pad + i0;
            if (cnt == 4) {
                *reinterpret_cast<uint4*>(row) = make_uint4(code[0], code[1], code[2], code[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cnt) row[k] = code[k];
            }
            if (gp.hashed[l]) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (bad[k] && k < cnt) {   // harmless without gradient; with gradient the level's owners take the generic path
                        const float2 g = dfeat[(int64_t)l * n + i0 + k];
                        if (!(g.x == 0.f && g.y == 0.f)) esc |= 1u << l;
                    }
            }
        }
    }
    if (esc) atomicOr(&esc_w[wave], esc);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) escape[word] = esc_w[wave];              // every word is written: no zero-fill needed
}

// ---- levels of 256..2048 tiles (log2_hashmap_size 22..25): per-tile bitmaps ---------------------------------------
// Beyond 255 tiles a level used to fall back to global atomics (2.1e10/s: 6.6 ms per 1 M samples at L = 20, T = 2^22): that many
// owners cannot each TEST every sample.  They do not have to: the pre-pass leaves one BIT per (tile, sample) -- row (level,
// tile) has bit i set when a (y,z) combination of sample i falls in the tile -- and an owner reads only its own row,
// 128 KiB per million samples, of which 4/tiles of the bits are set.  Rows are staged in LDS per block of 256 samples
// (tiles x 32 bytes) and written out whole, so the bitmaps need no zero fill; an owner gathers position and gradient of
// its few thousand samples lane by lane and works out the combinations from the (y,z) it gathered.  The cost of a level is
// its bitmap traffic (tiles x n / 8 bytes written and read once) plus 128 KiB of LDS zeroing and write-back per owner.
__global__ __launch_bounds__(256) void tile_bitmap_kernel(GridParams gp, TileParams tp, const float* __restrict__ x01,
                                                          const float2* __restrict__ dfeat, uint32_t* __restrict__ bitmaps,
                                                          uint32_t* __restrict__ esc_bm, int64_t n, const int64_t* __restrict__ n_dev) {
    const int64_t n_live = live_count(n, n_dev);
    int l = 0;
    while (!((tp.bitmap_levels >> l) & 1u) || tp.bm_idx[l] != (int)blockIdx.y) ++l;
    const int nt = tp.tiles_of[l];
    const int seg = tp.bm_samples / 32;         // words of a row this block writes (a block takes bm_samples samples: the larger,
                                                // the longer the contiguous pieces of the rows -- 128 bytes at 1024)
    extern __shared__ uint32_t stage[];         // [tile][seg]: the block's bits of every row
    __shared__ uint32_t esc_block;
    for (int r = threadIdx.x; r < nt * seg; r += 256) stage[r] = 0u;
    if (threadIdx.x == 0) esc_block = 0u;
    __syncthreads();
    for (int local = threadIdx.x; local < tp.bm_samples; local += 256) {
        const int64_t i = (int64_t)blockIdx.x * tp.bm_samples + local;
        if (i >= n_live) break;
        const float x = x01[3 * i], y = x01[3 * i + 1], z = x01[3 * i + 2];
        const uint32_t gy = (uint32_t)(int32_t)floorf(grid_pos(y, gp.scale[l])), gz = (uint32_t)(int32_t)floorf(grid_pos(z, gp.scale[l]));
        const uint32_t gx = (uint32_t)(int32_t)floorf(grid_pos(x, gp.scale[l]));
        const uint32_t word = (uint32_t)local >> 5, bit = 1u << ((uint32_t)local & 31u);
        const uint32_t useg = (uint32_t)seg;
        bool bad;
        if (gp.hashed[l]) {
            const uint32_t ay0 = gy * kPrimeY, ay1 = ay0 + kPrimeY, az0 = gz * kPrimeZ, az1 = az0 + kPrimeZ;
            const uint32_t m = gp.size[l] - 1u;
            atomicOr(&stage[(((ay0 ^ az0) & m) / (uint32_t)kTileEntries) * useg + word], bit);
            atomicOr(&stage[(((ay1 ^ az0) & m) / (uint32_t)kTileEntries) * useg + word], bit);
            atomicOr(&stage[(((ay0 ^ az1) & m) / (uint32_t)kTileEntries) * useg + word], bit);
            atomicOr(&stage[(((ay1 ^ az1) & m) / (uint32_t)kTileEntries) * useg + word], bit);
            bad = gx >= (uint32_t)(kTileEntries - 1);       // "(y,z) decides the tile" does not hold (see tile_codes_kernel)
        } else {            // dense: chunk-interleaved ownership; every corner on its own, indices past the end wrap (bwd_apply's rule),
                            // so these levels never escape
            const uint32_t res = gp.res[l], r2 = res * res, tmask = (uint32_t)nt - 1u;
            bad = false;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t idx = (gx + (uint32_t)(k & 1)) + (gy + (uint32_t)((k >> 1) & 1)) * res + (gz + (uint32_t)(k >> 2)) * r2;
                if (idx >= gp.size[l]) idx = idx % gp.size[l];
                atomicOr(&stage[((idx / kChunk) & tmask) * useg + word], bit);
            }
        }
        if (bad) {          // with gradient the level's owners take the generic path
            const float2 g = dfeat[(int64_t)l * n + i];
            if (!(g.x == 0.f && g.y == 0.f)) atomicOr(&esc_block, 1u);
        }
    }
    __syncthreads();
    uint32_t* rows = bitmaps + (int64_t)tp.bm_row[l] * tp.bm_row_words + (int64_t)blockIdx.x * seg;
    for (int r = threadIdx.x; r < nt * seg; r += 256) rows[(int64_t)(r / seg) * tp.bm_row_words + (r % seg)] = stage[r];
    if (threadIdx.x == 0) esc_bm[(int64_t)blockIdx.y * tp.bm_blocks + blockIdx.x] = esc_block;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// The loads of this loop are issued through inline assembly with hand-placed s_waitcnt: the compiler's own
// placement waits for a gather right where it is issued (it packs the loaded y,z into a register pair for a packed
// multiply) and drains vmcnt to 0 around the conditional drain.  VMEM loads return in issue order, so
// "vmcnt(k)" = "everything but the k youngest loads has landed"; the number of loads issued per step is static
// (2 code loads, then 3 gather loads per drain, idle lanes gather sample 0).
// (Round 4 also built 16-bit nibble codes -- two samples tested with four bit-parallel operations --: bit-identical, owners 378.9
//  vs 361.7 us per 1 M samples, profiles/r04_bwd_code16_ab.json; the variant is tools/exp/r05_retired_variants.diff.)
template <bool FIXED, bool DENSE>
__device__ __forceinline__ void bwd_stream_codes(const BwdCtx& cx, float* lds_tile, uint32_t* queue,
                                                 const uint32_t* __restrict__ codes_l, const float* __restrict__ x01,
                                                 const float2* __restrict__ g_l, int64_t n, int rep, int R) {
    // (tells the compiler's own wait-count bookkeeping that nothing it knows of is in flight when the loop starts;
    //  otherwise it drains vmcnt to 0 at the head of every iteration on behalf of the other streaming variants)
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
    constexpr int kPer = 4;                             // samples of a lane per iteration (= per pair of code loads)
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t qn = 0;                                    // wave-uniform queue fill
    const int64_t n_full = n / kPer;
    const int64_t g_lo = n_full * rep / R, g_hi = n_full * (rep + 1) / R;       // this replica's groups of kPer samples
    // registers written by loads in flight: only ever read through the wait_* copies below
    float ld_x = 0.f; f32x2 ld_yz = {0.f, 0.f}, ld_g = {0.f, 0.f}; u32x2 ld_c0 = {0u, 0u}, ld_c1 = {0u, 0u};
    uint32_t bcm = 0, bi = 0;                           // (y,z) combinations and sample index of the batch in flight
    bool blive = false;                                 // this lane holds an entry of the batch in flight
    const uint32_t t_split = 0x80u | cx.t, t_next = 0x80u | ((cx.t - 1u) & (cx.n_tiles - 1u));     // dense codes
    // -> combinations that name this tile; dense: 0x10 when a combination of the sample wraps past the end of the level
    //    (byte 0x7f) -- such a sample is queued by EVERY owner of the level without combinations and applied corner by
    //    corner with bwd_apply's wrapping rule (a few per cent of random points, none of a scene that keeps clear of the
    //    upper faces of its box: no reason to hand the whole level to the generic owners)
    auto test = [&](uint32_t code) {
        uint32_t cm = 0;
        bool wraps = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t b = (code >> (8 * c)) & 0xffu;
            const bool hit = DENSE ? (b == cx.t || b == t_split || b == t_next) : (b == cx.t);
            cm |= (hit ? 1u : 0u) << c;
            if (DENSE) wraps = wraps || b == 0x7fu;
        }
        return wraps ? 0x10u : cm;
    };
    auto enqueue = [&](uint32_t cm, uint32_t i) {
        const unsigned long long b = __ballot(cm != 0u);
        if (b) {
            const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (cm) queue[pos] = (i << 4) | (cm & 15u);
            qn += (uint32_t)__popcll(b);
        }
    };
    auto load_codes = [&](int64_t grp) {                // 2 loads
        const uint32_t off = (uint32_t)(grp < g_hi ? grp : g_hi - 1) * 16u;
        asm volatile("global_load_dwordx2 %0, %2, %3\n\tglobal_load_dwordx2 %1, %2, %3 offset:8"
                     : "=&v"(ld_c0), "=&v"(ld_c1) : "v"(off), "s"(codes_l) : "memory");
    };
    auto pop_and_gather = [&]() {       // up to 64 queued samples: issue the 3 loads of their position and gradient
        const uint32_t take = qn < 64u ? qn : 64u;
        __builtin_amdgcn_wave_barrier();
        uint32_t e = 0;
        if (lane < take) e = queue[qn - take + lane];
        __builtin_amdgcn_wave_barrier();
        qn -= take;
        blive = lane < take;
        bcm = e & 15u;
        bi = e >> 4;
        const uint32_t ox = bi * 12u, og = bi * 8u;
        asm volatile("global_load_dword %0, %3, %4\n\tglobal_load_dwordx2 %1, %3, %4 offset:4\n\tglobal_load_dwordx2 %2, %5, %6"
                     : "=&v"(ld_x), "=&v"(ld_yz), "=&v"(ld_g) : "v"(ox), "s"(x01), "v"(og), "s"(g_l) : "memory");
    };
    // wait until at most `younger` loads are in flight, then copy the landed registers (the copy is part of the
    // asm statement: the compiler must not move a read of those registers above the wait)
#define PERF_WAIT_BATCH(younger)                                                                                        \
    float bx; f32x2 byz, bg;                                                                                            \
    asm volatile("s_waitcnt vmcnt(" #younger ")\n\tv_mov_b32 %0, %3\n\tv_mov_b64 %1, %4\n\tv_mov_b64 %2, %5"           \
                 : "=&v"(bx), "=&v"(byz), "=&v"(bg) : "v"(ld_x), "v"(ld_yz), "v"(ld_g) : "memory")
    // The batch gathered one drain ago.  Every lane applies ONE combination (loop-free at full occupancy); the 7 % of
    // samples that name this tile with two or more combinations go back into the queue with the remaining ones.
    auto apply_batch = [&](float bx, f32x2 byz, f32x2 bg) {
        if (DENSE && blive && bcm == 0u) {      // a sample with a wrapping corner: every corner on its own
            if (!(bg.x == 0.f && bg.y == 0.f)) bwd_apply<FIXED, false>(cx, lds_tile, make_float2(bg.x, bg.y), bx, byz.x, byz.y);
            return;
        }
        const uint32_t rest = DENSE ? 0u : bcm & (bcm - 1u);        // (dense tiles are named by several combinations as a rule)
        if (!DENSE) bcm &= 0u - bcm;
        if (bcm) {
            const float px = grid_pos(bx, cx.scale), py = grid_pos(byz.x, cx.scale), pz = grid_pos(byz.y, cx.scale);
            const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
            const uint32_t gx = (uint32_t)(int32_t)flx;
            if (DENSE)
                apply_pairs_dense<FIXED>(cx, lds_tile, make_float2(bg.x, bg.y), gx, px - flx, py - fly, pz - flz,
                                         (uint32_t)(int32_t)fly * cx.res, (uint32_t)(int32_t)flz * cx.r2, bcm);
            else if (gx < (uint32_t)(kTileEntries - 1))      // (else: zero gradient, see tile_codes_kernel)
                apply_pairs<FIXED>(cx, lds_tile, make_float2(bg.x, bg.y), gx, px - flx, py - fly, pz - flz,
                                   (uint32_t)(int32_t)fly * kPrimeY, (uint32_t)(int32_t)flz * kPrimeZ, bcm);
        }
        if (!DENSE) enqueue(rest, bi);
    };
    if (g_hi > g_lo) {
        int64_t grp = g_lo + threadIdx.x;
        load_codes(grp);
        pop_and_gather();               // empty queue: dummy gather, keeps the in-flight count of the loop static
        for (int64_t base = g_lo + (int64_t)(threadIdx.x & ~63u); base < g_hi; base += kBwdThreads) {   // wave-uniform trip count
            u32x2 c0, c1;               // in flight, oldest first: 2 code loads, 3 gather loads
            asm volatile("s_waitcnt vmcnt(3)\n\tv_mov_b64 %0, %2\n\tv_mov_b64 %1, %3" : "=&v"(c0), "=&v"(c1) : "v"(ld_c0), "v"(ld_c1) : "memory");
            const int64_t g0 = grp;
            const bool valid = g0 < g_hi;
            grp += kBwdThreads;
            load_codes(grp);
            const uint32_t cs[4] = {c0.x, c0.y, c1.x, c1.y};
#pragma unroll
            for (int s = 0; s < 4; ++s) enqueue(valid ? test(cs[s]) : 0u, (uint32_t)(4 * g0 + s));
            {
                PERF_WAIT_BATCH(2);         // all but the 2 code loads
                apply_batch(bx, byz, bg);
            }
            pop_and_gather();
            while (qn >= 128u) {            // bursts (ray-coherent samples at coarse hashed levels)
                PERF_WAIT_BATCH(0);
                apply_batch(bx, byz, bg);
                pop_and_gather();
            }
        }
    }
    if (threadIdx.x < 64 && rep == 0) {                 // ragged tail (n % kPer samples)
        const int64_t i = n_full * kPer + lane;
        uint32_t cm = 0u;
        if (i < n) cm = test(codes_l[i]);
        enqueue(cm, (uint32_t)i);
    }
    for (;;) {
        PERF_WAIT_BATCH(0);
        apply_batch(bx, byz, bg);
        bcm = 0; blive = false;
        if (qn == 0u) break;
        pop_and_gather();
    }
    asm volatile("" : : "v"(ld_c0), "v"(ld_c1));       // (the last code loads landed with the vmcnt(0) above)
#undef PERF_WAIT_BATCH
}

// inclusive prefix sum over the 64 lanes of a wave (DPP: shifts inside rows of 16, then row broadcasts)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

// Owner of one tile of a bitmap level.  A wave reads 64 words of the row per step (2048 samples), turns the set bits into
// sample indices in its LDS queue with ONE wave scan (a row holds 4-5 / tiles of the samples), and applies the queue 64
// entries at a time at full lane occupancy: position and gradient gathered, the combinations that fall in this tile
// worked out from the gathered (y,z).  (Lanes that walk their own bits one after the other instead -- the first version --
// pay one memory latency per round and as many rounds as the fullest lane has bits: 0.10 ms per owner at 256 hashed
// tiles, 1.66 ms at 64 dense tiles.)  A step that would overflow the queue -- dense rows -- is fed nibble by nibble.
template <bool FIXED, bool DENSE>
__device__ __forceinline__ void bwd_stream_bitmap(const BwdCtx& cx, float* lds_tile, uint32_t* queue, const uint32_t* __restrict__ row,
                                                  const float* __restrict__ x01, const float2* __restrict__ g_l, int64_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t qn = 0;                                    // wave-uniform queue fill
    auto drain = [&]() {                                // the youngest min(qn, 64) entries
        const uint32_t take = qn < 64u ? qn : 64u;
        __builtin_amdgcn_wave_barrier();
        const bool live = lane < take;
        const uint32_t i = live ? queue[qn - take + lane] : 0u;
        __builtin_amdgcn_wave_barrier();
        qn -= take;
        const float2 g = g_l[i];
        const float x = x01[3 * (size_t)i], y = x01[3 * (size_t)i + 1], z = x01[3 * (size_t)i + 2];
        if (!live || (g.x == 0.f && g.y == 0.f)) return;
        if (DENSE) { bwd_apply<FIXED, false>(cx, lds_tile, g, x, y, z); return; }      // (tests the tile of each corner, wraps indices)
        const float px = grid_pos(x, cx.scale), py = grid_pos(y, cx.scale), pz = grid_pos(z, cx.scale);
        const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
        const uint32_t gx = (uint32_t)(int32_t)flx;
        if (gx >= (uint32_t)(kTileEntries - 1)) return;          // (zero gradient or the level escaped, see tile_bitmap_kernel)
        const uint32_t ay0 = (uint32_t)(int32_t)fly * kPrimeY, az0 = (uint32_t)(int32_t)flz * kPrimeZ;
        const uint32_t ay1 = ay0 + kPrimeY, az1 = az0 + kPrimeZ;
        const uint32_t cm = ((((ay0 ^ az0) & cx.mask) / (uint32_t)kTileEntries) == cx.t ? 1u : 0u) |
                            ((((ay1 ^ az0) & cx.mask) / (uint32_t)kTileEntries) == cx.t ? 2u : 0u) |
                            ((((ay0 ^ az1) & cx.mask) / (uint32_t)kTileEntries) == cx.t ? 4u : 0u) |
                            ((((ay1 ^ az1) & cx.mask) / (uint32_t)kTileEntries) == cx.t ? 8u : 0u);
        if (cm) apply_pairs<FIXED>(cx, lds_tile, g, gx, px - flx, py - fly, pz - flz, ay0, az0, cm);
    };
    auto enqueue = [&](uint32_t bits, uint32_t s0, uint32_t incl) {    // set bit b of `bits` = sample s0 + b; incl = inclusive wave
                                                                        // prefix of the popcounts; the caller made room
        const uint32_t cnt = (uint32_t)__popc(bits);
        uint32_t at = qn + incl - cnt;
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            queue[at++] = s0 + (uint32_t)b;
        }
        qn += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    };
    const int64_t n_words = (n + 31) / 32;
    const int64_t w_first = (int64_t)(threadIdx.x >> 6) * 64 + lane, w_step = (int64_t)(kBwdThreads / 64) * 64;
    uint32_t next = w_first < n_words ? row[w_first] : 0u;
    for (int64_t w = w_first; w - lane < n_words; w += w_step) {     // wave-uniform trip count
        const uint32_t bits = next;
        next = (w + w_step < n_words) ? row[w + w_step] : 0u;
        const uint32_t incl = wave_inclusive_sum((uint32_t)__popc(bits));
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (qn + total <= (uint32_t)kQueueCap) {
            enqueue(bits, (uint32_t)w * 32u, incl);
            while (qn >= 64u) drain();
        } else {            // (at most 256 new entries per nibble, fewer than 64 left over from the one before)
#pragma unroll 1
            for (int s = 0; s < 8; ++s) {
                const uint32_t nib = (bits >> (4 * s)) & 15u;
                enqueue(nib, (uint32_t)w * 32u + 4u * (uint32_t)s, wave_inclusive_sum((uint32_t)__popc(nib)));
                while (qn >= 64u) drain();
            }
        }
    }
    while (qn) drain();
}

// Single-tile dense levels (the coarsest ones: a cell is several sample spacings wide): a thread walks consecutive
// samples of a ray, so it sums a RUN of samples that share a cell in registers -- 8 packed corner sums -- and touches LDS
// only when the cell changes.  At res 16 / 23 that is a fifth / a quarter of the LDS atomics, which is what these
// owners were bound by (the lanes of a wave pile up on the few cells a scene populates).  Fixed point: same integers,
// same sums; fp32: a different (shorter) summation order.
template <bool FIXED>
struct RunAcc {
    uint32_t base;              // index of the run's (0,0,0) corner; 0xffffffff: empty
    long long v[8];             // FIXED: packed field pairs; else two floats bit-cast
};

template <bool FIXED>
__device__ __forceinline__ void run_flush(const BwdCtx& cx, float* lds_tile, RunAcc<FIXED>& r) {
    if (r.base == 0xffffffffu) return;
    unsigned long long* lds64 = reinterpret_cast<unsigned long long*>(lds_tile);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t idx = r.base + (uint32_t)(k & 1) + ((k >> 1) & 1 ? cx.res : 0u) + ((k >> 2) ? cx.r2 : 0u);
        if (idx >= cx.size) idx = idx % cx.size;
        if (FIXED) {
            if (r.v[k] != 0ll) atomicAdd(&lds64[idx], (unsigned long long)r.v[k]);
        } else {
            const float a = __int_as_float((int)(uint32_t)(unsigned long long)r.v[k]), b = __int_as_float((int)(uint32_t)((unsigned long long)r.v[k] >> 32));
            unsafeAtomicAdd(&lds_tile[2 * idx], a); unsafeAtomicAdd(&lds_tile[2 * idx + 1], b);
        }
    }
}

template <bool FIXED>
__device__ __forceinline__ void run_apply(const BwdCtx& cx, float* lds_tile, RunAcc<FIXED>& r, const float2 g, const float x,
                                          const float y, const float z) {
    const float px = grid_pos(x, cx.scale), py = grid_pos(y, cx.scale), pz = grid_pos(z, cx.scale);
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    float fx = px - flx, fy = py - fly, fz = pz - flz;
    const uint32_t base = (uint32_t)(int32_t)flx + (uint32_t)(int32_t)fly * cx.res + (uint32_t)(int32_t)flz * cx.r2;
    if (cx.smooth) { fx = fx * fx * (3.0f - 2.0f * fx); fy = fy * fy * (3.0f - 2.0f * fy); fz = fz * fz * (3.0f - 2.0f * fz); }
    const bool fresh = base != r.base;
    if (fresh) { run_flush<FIXED>(cx, lds_tile, r); r.base = base; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int bx = k & 1, by = (k >> 1) & 1, bz = k >> 2;
        const float w = ((bx ? fx : 1.0f - fx) * (by ? fy : 1.0f - fy)) * (bz ? fz : 1.0f - fz);        // association of bwd_apply
        if (FIXED) {
            const long long lo = (long long)__float2int_rn(w * g.x * cx.to_fixed);
            const long long hi = (long long)__float2int_rn(w * g.y * cx.to_fixed);
            const long long v = (hi << 32) + lo;
            r.v[k] = fresh ? v : r.v[k] + v;
        } else {
            const float a = w * g.x, b = w * g.y;
            const float pa = fresh ? 0.f : __int_as_float((int)(uint32_t)(unsigned long long)r.v[k]);
            const float pb = fresh ? 0.f : __int_as_float((int)(uint32_t)((unsigned long long)r.v[k] >> 32));
            r.v[k] = (long long)(((unsigned long long)(uint32_t)__float_as_int(pb + b) << 32) | (unsigned long long)(uint32_t)__float_as_int(pa + a));
        }
    }
}

// Streaming loop: a thread owns 4 consecutive samples per iteration -- 3 x 16 B of positions + 2 x 16 B of
// gradients, all 16-byte loads -- and the next group is in flight while the current one is applied.
template <bool FIXED, bool HASHED>
__device__ __forceinline__ void bwd_stream(const BwdCtx& cx, float* lds_tile, const float* __restrict__ x01,
                                           const float2* __restrict__ g_l, int64_t n, int rep, int R, const bool run_merge = false) {
    constexpr int kGroup = 4;
    const int64_t n_full = n / kGroup;
    const float4* x4 = reinterpret_cast<const float4*>(x01);
    const float4* g4 = reinterpret_cast<const float4*>(g_l);
    const bool aligned = ((reinterpret_cast<uintptr_t>(x01) | reinterpret_cast<uintptr_t>(g_l)) & 15) == 0;
    if (aligned && !HASHED) {
        // Dense levels: samples that follow each other along a ray fall into the same cell, i.e. the lanes of a wave
        // would all add to the same 8 LDS addresses (measured: ~70 cycles per ds_add instruction).  Each thread
        // therefore walks its OWN contiguous run of groups, starting at a lane-dependent rotation, so that the lanes
        // of a wave sit on different rays at different depths.
        const int64_t g_lo = n_full * rep / R, g_hi = n_full * (rep + 1) / R;
        const int64_t len = (g_hi - g_lo + kBwdThreads - 1) / kBwdThreads;
        const int64_t t_lo = g_lo + (int64_t)threadIdx.x * len;
        const int64_t mine = (t_lo >= g_hi) ? 0 : ((g_hi - t_lo < len) ? g_hi - t_lo : len);
        int64_t j = (len * (int64_t)(threadIdx.x & 63u)) / 64;        // rotation (wraps inside [0, len))
        float4 xa = {}, xb = {}, xc = {}, ga = {}, gb = {};
        auto fetch = [&](int64_t jj) {
            if (jj < mine) { const int64_t grp = t_lo + jj; xa = x4[3 * grp]; xb = x4[3 * grp + 1]; xc = x4[3 * grp + 2]; ga = g4[2 * grp]; gb = g4[2 * grp + 1]; }
        };
        fetch(j);
        if (run_merge && cx.n_tiles == 1u) {
            RunAcc<FIXED> run;
            run.base = 0xffffffffu;
            for (int64_t it = 0; it < len; ++it) {
                const float4 cxa = xa, cxb = xb, cxc = xc, cga = ga, cgb = gb;
                const bool live = j < mine;
                j = (j + 1 == len) ? 0 : j + 1;
                if (it + 1 < len) fetch(j);
                if (live) {
                    if (!(cga.x == 0.f && cga.y == 0.f)) run_apply<FIXED>(cx, lds_tile, run, make_float2(cga.x, cga.y), cxa.x, cxa.y, cxa.z);
                    if (!(cga.z == 0.f && cga.w == 0.f)) run_apply<FIXED>(cx, lds_tile, run, make_float2(cga.z, cga.w), cxa.w, cxb.x, cxb.y);
                    if (!(cgb.x == 0.f && cgb.y == 0.f)) run_apply<FIXED>(cx, lds_tile, run, make_float2(cgb.x, cgb.y), cxb.z, cxb.w, cxc.x);
                    if (!(cgb.z == 0.f && cgb.w == 0.f)) run_apply<FIXED>(cx, lds_tile, run, make_float2(cgb.z, cgb.w), cxc.y, cxc.z, cxc.w);
                }
            }
            run_flush<FIXED>(cx, lds_tile, run);
        } else
        for (int64_t it = 0; it < len; ++it) {
            const float4 cxa = xa, cxb = xb, cxc = xc, cga = ga, cgb = gb;
            const bool live = j < mine;
            j = (j + 1 == len) ? 0 : j + 1;
            if (it + 1 < len) fetch(j);
            if (live) {
                if (!(cga.x == 0.f && cga.y == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, make_float2(cga.x, cga.y), cxa.x, cxa.y, cxa.z);
                if (!(cga.z == 0.f && cga.w == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, make_float2(cga.z, cga.w), cxa.w, cxb.x, cxb.y);
                if (!(cgb.x == 0.f && cgb.y == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, make_float2(cgb.x, cgb.y), cxb.z, cxb.w, cxc.x);
                if (!(cgb.z == 0.f && cgb.w == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, make_float2(cgb.z, cgb.w), cxc.y, cxc.z, cxc.w);
            }
        }
    } else if (aligned) {
        int64_t grp = (int64_t)rep * kBwdThreads + threadIdx.x;
        const int64_t gstride = (int64_t)R * kBwdThreads;
        float4 xa = {}, xb = {}, xc = {}, ga = {}, gb = {};
        if (grp < n_full) { xa = x4[3 * grp]; xb = x4[3 * grp + 1]; xc = x4[3 * grp + 2]; ga = g4[2 * grp]; gb = g4[2 * grp + 1]; }
        while (grp < n_full) {
            const float4 cxa = xa, cxb = xb, cxc = xc, cga = ga, cgb = gb;
            grp += gstride;
            if (grp < n_full) { xa = x4[3 * grp]; xb = x4[3 * grp + 1]; xc = x4[3 * grp + 2]; ga = g4[2 * grp]; gb = g4[2 * grp + 1]; }
            if (!(cga.x == 0.f && cga.y == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, make_float2(cga.x, cga.y), cxa.x, cxa.y, cxa.z);
            if (!(cga.z == 0.f && cga.w == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, make_float2(cga.z, cga.w), cxa.w, cxb.x, cxb.y);
            if (!(cgb.x == 0.f && cgb.y == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, make_float2(cgb.x, cgb.y), cxb.z, cxb.w, cxc.x);
            if (!(cgb.z == 0.f && cgb.w == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, make_float2(cgb.z, cgb.w), cxc.y, cxc.z, cxc.w);
        }
    }
    if (aligned) {
        if (rep == 0) {     // ragged tail (n % 4 samples)
            const int64_t i = n_full * kGroup + threadIdx.x;
            if (i < n) { const float2 g = g_l[i]; if (!(g.x == 0.f && g.y == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, g, x01[3 * i], x01[3 * i + 1], x01[3 * i + 2]); }
        }
    } else {
        for (int64_t i = (int64_t)rep * kBwdThreads + threadIdx.x; i < n; i += (int64_t)R * kBwdThreads) {
            const float2 g = g_l[i];
            if (!(g.x == 0.f && g.y == 0.f)) bwd_apply<FIXED, HASHED>(cx, lds_tile, g, x01[3 * i], x01[3 * i + 1], x01[3 * i + 2]);
        }
    }
}

// FIXED = true: the two features of an entry are accumulated as two signed 32-bit fixed-point fields packed in one
// 64-bit LDS word (sum of h*2^32 + l is exact integer arithmetic; fields are recovered at write-back) with ONE
// full-rate integer ds_add_u64 per corner -- gfx950 serialises ds_add_f32 at ~3 cycles per active lane
// (tools/exp/lds_atomics.hip).  The unit is a power of two derived on the device from the level's max |dfeat|:
// unit = 2^ceil(log2(absmax)) * 2^(headroom - 31).  A tile whose largest field comes within 2x of the int32 range
// raises *overflow_flag (the caller then falls back to the fp32 mode).
template <bool FIXED>
__global__ __launch_bounds__(kBwdThreads) void hashgrid_bwd_kernel(GridParams gp, TileParams tp,
                                                                   const float* __restrict__ x01,
                                                                   const float2* __restrict__ dfeat,
                                                                   float2* __restrict__ grad, float2* __restrict__ ws,
                                                                   const float* __restrict__ level_absmax,
                                                                   int32_t* __restrict__ overflow_flag,
                                                                   int32_t* __restrict__ hr_state,
                                                                   const int32_t* __restrict__ shifts_in,
                                                                   int32_t* __restrict__ shifts_ws,
                                                                   const uint32_t* __restrict__ codes,
                                                                   const uint32_t* __restrict__ escape,
                                                                   const uint32_t* __restrict__ bitmaps,
                                                                   const uint32_t* __restrict__ esc_bm, int64_t n,
                                                                   const int64_t* __restrict__ n_dev,
                                                                   const int32_t* __restrict__ redo_flag) {
    // a predicated REDO launch (perf_hashgrid_bwd, redo_flag): nothing happens unless the fixed-point call before it raised
    // the flag -- the graph node costs a dispatch, the gradient table is left as that call wrote it
    if (redo_flag && redo_flag[0] == 0) return;
    if (redo_flag && hr_state && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&hr_state[2 * PERF_MAX_LEVELS + 1], 1);   // (statistics)
    const int64_t n_live = live_count(n, n_dev);            // samples present; n stays the stride of dfeat / codes
    extern __shared__ __attribute__((aligned(16))) float lds_tile[];   // 2 * kTileEntries floats (+ the wave queues)
    unsigned long long* lds64 = reinterpret_cast<unsigned long long*>(lds_tile);
    int b = blockIdx.x, l = 0;
    uint32_t t;
    int rep;
    if (tp.use_work) {
        const uint32_t wk = tp.work[blockIdx.x];
        if (wk == 0xffffffffu) return;
        l = (int)(wk >> 16); t = (wk >> 8) & 0xffu; rep = (int)(wk & 0xffu);
    } else {
        while (b >= tp.tiles_of[l] * tp.replicas_of[l]) { b -= tp.tiles_of[l] * tp.replicas_of[l]; ++l; }
        t = (uint32_t)(b / tp.replicas_of[l]);
        rep = b % tp.replicas_of[l];
    }
    const int R = tp.replicas_of[l];
    const uint32_t n_tiles = (uint32_t)tp.tiles_of[l];
    const uint32_t size = gp.size[l];
    const bool hashed = gp.hashed[l] != 0;
    for (int i = threadIdx.x; i < 2 * kTileEntries / 4; i += kBwdThreads)
        reinterpret_cast<float4*>(lds_tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    float from_fixed = 1.0f;
    BwdCtx cx;
    cx.scale = gp.scale[l]; cx.res = gp.res[l]; cx.r2 = cx.res * cx.res; cx.size = size; cx.mask = size - 1u;
    cx.n_tiles = n_tiles; cx.t = t; cx.tile_shift = (uint32_t)(__ffs((int)n_tiles) - 1);
    cx.smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
    cx.to_fixed = 1.0f;
    if (FIXED) {
        // units per 1.0 = 2^sh: given by the caller (job-wide units of a data-parallel step) or derived here; the level's
        // first workgroup leaves it for the replica reduction
        const int sh = shifts_in ? shifts_in[l] : fixed_point_shift(level_absmax[l], n_live, size, hr_state, l);
        cx.to_fixed = ldexpf(1.0f, sh);
        from_fixed = ldexpf(1.0f, -sh);
        if (shifts_ws && t == 0u && rep == 0 && threadIdx.x == 0) shifts_ws[l] = sh;
    }
    const float2* g_l = dfeat + (int64_t)l * n;
    bool coded = codes && tp.code_slot[l] >= 0;
    if (coded) {                        // any escape bit for this level in the pre-pass blocks' words?
        __shared__ uint32_t esc_any;
        if (threadIdx.x == 0) esc_any = 0u;
        __syncthreads();
        uint32_t e = 0u;
        const int64_t n_words = (n_live + kCodeSamplesPerBlock - 1) / kCodeSamplesPerBlock;      // (blocks without live samples wrote 0)
        for (int64_t w = threadIdx.x; w < n_words; w += kBwdThreads) e |= escape[w];
        if ((e >> l) & 1u) esc_any = 1u;
        __syncthreads();
        coded = esc_any == 0u;
    }
    bool by_bitmap = bitmaps && ((tp.bitmap_levels >> l) & 1u);
    if (by_bitmap) {                    // (same escape rule as the coded levels: one word per pre-pass block and level)
        __shared__ uint32_t esc_bm_any;
        if (threadIdx.x == 0) esc_bm_any = 0u;
        __syncthreads();
        uint32_t e = 0u;
        const int64_t n_words = (n_live + tp.bm_samples - 1) / tp.bm_samples;
        const uint32_t* ew = esc_bm + (int64_t)tp.bm_idx[l] * tp.bm_blocks;
        for (int64_t w = threadIdx.x; w < n_words; w += kBwdThreads) e |= ew[w];
        if (e) esc_bm_any = 1u;
        __syncthreads();
        by_bitmap = esc_bm_any == 0u;
    }
    uint32_t* queue = reinterpret_cast<uint32_t*>(lds_tile + 2 * kTileEntries) + (threadIdx.x >> 6) * kQueueCap;
    if (by_bitmap && hashed) bwd_stream_bitmap<FIXED, false>(cx, lds_tile, queue, bitmaps + (int64_t)(tp.bm_row[l] + (int)t) * tp.bm_row_words, x01, g_l, n_live);
    else if (by_bitmap) bwd_stream_bitmap<FIXED, true>(cx, lds_tile, queue, bitmaps + (int64_t)(tp.bm_row[l] + (int)t) * tp.bm_row_words, x01, g_l, n_live);
    else if (coded && hashed) bwd_stream_codes<FIXED, false>(cx, lds_tile, queue, codes + (int64_t)tp.code_slot[l] * tp.n_pad, x01, g_l, n_live, rep, R);
    else if (coded) bwd_stream_codes<FIXED, true>(cx, lds_tile, queue, codes + (int64_t)tp.code_slot[l] * tp.n_pad, x01, g_l, n_live, rep, R);
    else if (hashed) bwd_stream<FIXED, true>(cx, lds_tile, x01, g_l, n_live, rep, R);
    else bwd_stream<FIXED, false>(cx, lds_tile, x01, g_l, n_live, rep, R, tp.run_merge != 0);
    __syncthreads();
    int32_t field_max = 0;
    // ---- write back: local slot j of tile t is entry e(j)
    const float2* src = reinterpret_cast<const float2*>(lds_tile);
    float2* out = (R > 1) ? ws + tp.ws_off[l] + (int64_t)rep * size : grad + gp.offset[l];
    const bool acc = (R == 1) && tp.accumulate;
    // fixed point: replica slabs -- and, in raw mode, the table itself -- receive the integer fields, so that replicas
    // (and the ranks of a data-parallel step) are added up exactly, in any order
    const bool int_out = FIXED && (R > 1 || tp.raw_out);
    for (uint32_t j = threadIdx.x; j < (uint32_t)kTileEntries; j += kBwdThreads) {
        uint32_t e;
        if (hashed) e = t * (uint32_t)kTileEntries + j;
        else e = ((j / kChunk) * n_tiles + t) * kChunk + (j % kChunk);      // n_tiles is a power of two
        if (e >= size) continue;
        float2 v;
        if (FIXED) {
            const long long tot = (long long)lds64[j];
            const int32_t lo = (int32_t)(tot & 0xffffffffll);
            const int32_t hi = (int32_t)((tot - (long long)lo) >> 32);
            const int32_t alo = lo < 0 ? -(lo + 1) : lo, ahi = hi < 0 ? -(hi + 1) : hi;
            field_max = max(field_max, max(alo, ahi));
            if (int_out) { reinterpret_cast<int2*>(out)[e] = make_int2(lo, hi); continue; }
            v = make_float2((float)lo * from_fixed, (float)hi * from_fixed);
        } else {
            v = src[j];
        }
        if (acc) { const float2 o = out[e]; v.x += o.x; v.y += o.y; }
        out[e] = v;
    }
    // (a field that wrapped past +-2^31 reads back with an arbitrary value; the flag is global, so such a sum escapes only
    //  if NO field of ANY tile ends in the band [2^29, 2^32 - 2^29) -- i.e. if the largest sum of the whole table exceeds
    //  7x the level at which smaller sums already raise the flag while none of them lands there)
    if (FIXED && overflow_flag && field_max >= (1 << 29)) atomicOr(overflow_flag, 1);
    if (FIXED && hr_state && R == 1) {  // largest |field| of the level, for the feedback (one atomic per wave; replicated
                                        // levels report the max of their SUMMED fields from the reduction kernel)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) field_max = max(field_max, __shfl_xor(field_max, off));
        // (the 24 maxima share a cache line, and read-modify-writes on one line retire one at a time, ~11 ns each: 192 hashed
        //  owners x 16 waves ending together would queue for tens of microseconds -- only a wave that RAISES the maximum needs one)
        if ((threadIdx.x & 63) == 0 && field_max > 0 &&
            field_max > __hip_atomic_load(&hr_state[PERF_MAX_LEVELS + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&hr_state[PERF_MAX_LEVELS + l], field_max);
    }
}

// Levels too large for LDS owners: plain scatter with global fp32 atomics (flat ~2e10/s on gfx950: a fallback for
// log2_hashmap_size >= 22, where 256+ owners per level would each have to walk every sample).  The level's slice of
// the gradient table is zeroed by the caller first unless it accumulates.
__global__ __launch_bounds__(256) void hashgrid_bwd_atomic_kernel(GridParams gp, uint32_t levels, const float* __restrict__ x01,
                                                                  const float2* __restrict__ dfeat, float* __restrict__ grad,
                                                                  int64_t n, const int64_t* __restrict__ n_dev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= live_count(n, n_dev) || !((levels >> l) & 1u)) return;
    const float2 g = dfeat[(int64_t)l * n + i];
    if (g.x == 0.f && g.y == 0.f) return;
    const Corners c = corners_of(x01[3 * i], x01[3 * i + 1], x01[3 * i + 2], gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
    float w[8];
    corner_weights(c.f, gp.interpolation == PERF_INTERP_SMOOTHSTEP, w);
    float* t = grad + 2 * gp.offset[l];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsafeAtomicAdd(t + 2 * (uint64_t)c.idx[k], w[k] * g.x);
        unsafeAtomicAdd(t + 2 * (uint64_t)c.idx[k] + 1, w[k] * g.y);
    }
}

// Fixed-point flavour of the fallback (the caller provides level_absmax and does not accumulate): the level's slice of the
// gradient table is used as an array of 64-bit words holding the same two signed 32-bit fixed-point fields as the LDS
// tiles -- ONE global atomic per corner instead of two, and integer sums: the result does not depend on the order the
// atomics retire in.  hashgrid_bwd_unfix_kernel turns the words into float2 in place.
__global__ __launch_bounds__(256) void hashgrid_bwd_atomic_fixed_kernel(GridParams gp, uint32_t levels, const float* __restrict__ x01,
                                                                        const float2* __restrict__ dfeat, float* __restrict__ grad,
                                                                        const float* __restrict__ level_absmax,
                                                                        const int32_t* __restrict__ hr_state, int64_t n,
                                                                        const int64_t* __restrict__ n_dev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    const int64_t n_live = live_count(n, n_dev);
    if (i >= n_live || !((levels >> l) & 1u)) return;
    const float2 g = dfeat[(int64_t)l * n + i];
    if (g.x == 0.f && g.y == 0.f) return;
    const float to_fixed = ldexpf(1.0f, fixed_point_shift(level_absmax[l], n_live, gp.size[l], hr_state, l));
    const Corners c = corners_of(x01[3 * i], x01[3 * i + 1], x01[3 * i + 2], gp.scale[l], gp.res[l], gp.size[l], gp.hashed[l] != 0);
    float w[8];
    corner_weights(c.f, gp.interpolation == PERF_INTERP_SMOOTHSTEP, w);
    if (gp.hashed[l] && c.cell[0] < (uint32_t)(kTileEntries - 1)) {
        // the association of the tile owners (apply_pairs: x-weight times the (y,z) product), so that a level adds up the same
        // integers whether its owners read bitmaps or -- workspace too small for them -- this scatter runs
        float fx = c.f[0], fy = c.f[1], fz = c.f[2];
        if (gp.interpolation == PERF_INTERP_SMOOTHSTEP) { fx = fx * fx * (3.0f - 2.0f * fx); fy = fy * fy * (3.0f - 2.0f * fy); fz = fz * fz * (3.0f - 2.0f * fz); }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float wyz = (((k >> 1) & 1) ? fy : 1.0f - fy) * ((k >> 2) ? fz : 1.0f - fz);
            w[k] = ((k & 1) ? fx : 1.0f - fx) * wyz;
        }
    }
    unsigned long long* t = reinterpret_cast<unsigned long long*>(grad) + gp.offset[l];
    const float sx = g.x * to_fixed, sy = g.y * to_fixed;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long v = ((long long)__float2int_rn(w[k] * sy) << 32) + (long long)__float2int_rn(w[k] * sx);
        atomicAdd(t + c.idx[k], (unsigned long long)v);
    }
}

__global__ __launch_bounds__(256) void hashgrid_bwd_unfix_kernel(GridParams gp, uint32_t levels, float* __restrict__ grad,
                                                                 const float* __restrict__ level_absmax, int32_t* __restrict__ hr_state,
                                                                 int32_t* __restrict__ overflow_flag, int64_t n,
                                                                 const int64_t* __restrict__ n_dev) {
    const int l = blockIdx.y;
    if (!((levels >> l) & 1u)) return;
    const uint32_t size = gp.size[l];
    const float from_fixed = ldexpf(1.0f, -fixed_point_shift(level_absmax[l], live_count(n, n_dev), size, hr_state, l));
    unsigned long long* t = reinterpret_cast<unsigned long long*>(grad) + gp.offset[l];
    int32_t field_max = 0;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < size; e += gridDim.x * 256) {
        const long long tot = (long long)t[e];
        if (tot == 0) continue;                          // (0 is 0.0f, 0.0f: untouched entries need no store)
        const int32_t lo = (int32_t)(tot & 0xffffffffll);
        const int32_t hi = (int32_t)((tot - (long long)lo) >> 32);
        reinterpret_cast<float2*>(t)[e] = make_float2((float)lo * from_fixed, (float)hi * from_fixed);
        const int32_t alo = lo < 0 ? -(lo + 1) : lo, ahi = hi < 0 ? -(hi + 1) : hi;
        field_max = max(field_max, max(alo, ahi));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) field_max = max(field_max, __shfl_xor(field_max, off));
    if ((threadIdx.x & 63) == 0 && field_max > 0) {
        if (overflow_flag && field_max >= (1 << 29)) atomicOr(overflow_flag, 1);
        if (hr_state) atomicMax(&hr_state[PERF_MAX_LEVELS + l], field_max);
    }
}

__host__ __device__ inline int reduce_blocks_of(uint32_t size, int width) {
    const uint32_t want = (size + 1023u) / 1024u;
    return (int)(want < (uint32_t)width ? (want > 0u ? want : 1u) : (uint32_t)width);
}

// sum the replica slabs (ws[level][replica][entry]) of the replicated (coarse) levels into the gradient table; the
// workgroup that finishes last applies the headroom feedback (hr_state: [adjustments][largest fields][done counter])
__global__ __launch_bounds__(256) void hashgrid_bwd_reduce_kernel(GridParams gp, TileParams tp, const float2* __restrict__ ws,
                                                                  float2* __restrict__ grad, int32_t* __restrict__ hr_state,
                                                                  const int32_t* __restrict__ shifts, int fixed,
                                                                  int32_t* __restrict__ overflow_flag, int n_ticket_blocks) {
    const int l = blockIdx.y;
    const int R = tp.replicas_of[l];
    // (rows of levels without replicas have nothing to add up: they leave at once, without a ticket -- same-address atomics
    //  retire at ~11 ns each; with no replicated level at all, workgroup (0, 0) applies the feedback)
    if (R <= 1 && !(n_ticket_blocks == 0 && blockIdx.x == 0 && blockIdx.y == 0)) return;
    // a replicated level takes one workgroup per 1,024 entries (at most the grid's width): the smallest ones must not pay -- in
    // tickets -- for the width the largest one needs (reduce_blocks_of() is what the host counted)
    if (R > 1 && (int)blockIdx.x >= reduce_blocks_of(gp.size[l], (int)gridDim.x)) return;
    int sink = 0;
    if (R > 1) {
        const uint32_t size = gp.size[l];
        const float from_fixed = fixed ? ldexpf(1.0f, -shifts[l]) : 1.0f;
        int32_t field_max = 0;
        const uint32_t stride = (uint32_t)reduce_blocks_of(size, (int)gridDim.x) * 256;
        if (fixed) {
            // Four entries x four slabs per round trip: a thread of the largest replicated level owns a dozen entries, and
            // written as "for entry: for slab: load, add" every one of its 4 R loads was a round trip of its own (14 us for a
            // few megabytes).  Indices are clamped so that all sixteen loads are unconditional; integer sums, any order.
            const int2* p = reinterpret_cast<const int2*>(ws + tp.ws_off[l]);
            for (uint32_t e0 = blockIdx.x * 256 + threadIdx.x; e0 < size; e0 += 4 * stride) {
                int32_t sx[4] = {0, 0, 0, 0}, sy[4] = {0, 0, 0, 0};
                uint32_t ej[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) ej[j] = min(e0 + (uint32_t)j * stride, size - 1u);
                for (int r = 0; r < R; r += 4) {
                    int2 v[4][4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int64_t slab = (int64_t)min(r + q, R - 1) * size;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[q][j] = p[slab + ej[j]];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (r + q < R) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) { sx[j] += v[q][j].x; sy[j] += v[q][j].y; }
                        }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t e = e0 + (uint32_t)j * stride;
                    if (e >= size) break;
                    float2* o = grad + gp.offset[l] + e;
                    const int32_t ax = sx[j] < 0 ? -(sx[j] + 1) : sx[j], ay = sy[j] < 0 ? -(sy[j] + 1) : sy[j];
                    field_max = max(field_max, max(ax, ay));
                    if (tp.raw_out) { *reinterpret_cast<int2*>(o) = make_int2(sx[j], sy[j]); continue; }
                    float fx = (float)sx[j] * from_fixed, fy = (float)sy[j] * from_fixed;
                    if (tp.accumulate) { const float2 c = *o; fx += c.x; fy += c.y; }
                    *o = make_float2(fx, fy);
                }
            }
        } else {
            for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < size; e += stride) {
                float2* o = grad + gp.offset[l] + e;
                const float2* p = ws + tp.ws_off[l] + e;
                float sx = 0.f, sy = 0.f;
                int r = 0;
                for (; r + 4 <= R; r += 4) {        // (loads in flight together, additions in slab order as before)
                    const float2 v0 = p[(int64_t)r * size], v1 = p[(int64_t)(r + 1) * size], v2 = p[(int64_t)(r + 2) * size],
                                 v3 = p[(int64_t)(r + 3) * size];
                    sx += v0.x; sy += v0.y; sx += v1.x; sy += v1.y; sx += v2.x; sy += v2.y; sx += v3.x; sy += v3.y;
                }
                for (; r < R; ++r) { const float2 v = p[(int64_t)r * size]; sx += v.x; sy += v.y; }
                if (tp.accumulate) { const float2 c = *o; sx += c.x; sy += c.y; }
                *o = make_float2(sx, sy);
            }
        }
        if (fixed) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) field_max = max(field_max, __shfl_xor(field_max, off));
            if ((threadIdx.x & 63) == 0 && field_max > 0) {
                if (overflow_flag && field_max >= (1 << 29)) atomicOr(overflow_flag, 1);
                // a RETURNING device-scope atomic: the wave waits until it has been performed at the memory side, so the
                // ticket below is ordered behind it without an agent-scope fence (which writes back the XCD's L2: ~20 us
                // over the 128 workgroups of this kernel, measured).  Only a wave that RAISES the level's maximum needs it
                // (same-address read-modify-writes retire one at a time, ~11 ns each; a look costs an L2 read).
                if (hr_state && field_max > __hip_atomic_load(&hr_state[PERF_MAX_LEVELS + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    sink = __hip_atomic_fetch_max(&hr_state[PERF_MAX_LEVELS + l], field_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!hr_state) return;
    // ---- headroom feedback by the last workgroup of the replicated rows to get here
    __shared__ int last_block;
    if (n_ticket_blocks > 0) {
        asm volatile("" : : "v"(sink));                    // (uses the atomic's return value: the wave waits for it)
        __syncthreads();
        if (threadIdx.x == 0) last_block = (atomicAdd(&hr_state[2 * PERF_MAX_LEVELS], 1) == n_ticket_blocks - 1) ? 1 : 0;
        __syncthreads();
        if (!last_block) return;
    }
    if ((int)threadIdx.x < gp.n_levels) {
        const int fm = atomicMax(&hr_state[PERF_MAX_LEVELS + threadIdx.x], 0);     // (an atomic read: the value sits at the memory side)
        hr_state[threadIdx.x] = headroom_feedback(hr_state[threadIdx.x], fm);
        hr_state[PERF_MAX_LEVELS + threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) hr_state[2 * PERF_MAX_LEVELS] = 0;
}

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

static inline unsigned grouped_grid(int64_t n) { return (unsigned)(div_up(n, 256) * 8); }

}  // namespace perf

using namespace perf;

extern "C" int perf_hashgrid_fwd(const perf_grid_desc* grid, const float* x01, const void* table16,
                                 void* feat16, int64_t n, const int64_t* n_dev, int dtype, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    PERF_REQUIRE(n >= 0 && n < (int64_t(1) << 31) * 16, "n out of range");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && table16 && feat16, "NULL pointer");
    // level group <-> XCD pinning only pays when every one of the 8 groups has a level (L >= 15); a grid of a few levels
    // (a rank's slice of a level-sharded table, the 5-level proposal field) would otherwise keep 1-3 XCDs busy
    const int xcd_affinity = gp.n_levels >= 15 ? 1 : 0;
    // chunks (of 256 samples) per level group in one launch; beyond that the workgroups loop
    int64_t chunks = div_up(n, 256);
    if (xcd_affinity && chunks > kFwdMaxChunks) chunks = kFwdMaxChunks;
    // ---- rotating level groups + run de-duplication (15/16-level grids).  (The measured-slower settings of this path -- no
    //      rotation, no de-duplication, the round-2 kernel for these grids, looping workgroups -- are tools/exp/r05_retired_variants.diff.)
    if (xcd_affinity && gp.n_levels <= 16) {
        // (one workgroup per chunk up to 4096 chunks per XCD: the rotation relies on chunks being served in dispatch order --
        //  512 looping workgroups per XCD measured 0.307 instead of 0.177 ms per 1 M samples: phases mix, every L2 sees every table)
        dim3 g((unsigned)(chunks * 8)), b(256);
        if (dtype == PERF_DTYPE_BF16)
            hipLaunchKernelGGL(hashgrid_fwd_v2_kernel<BF16>, g, b, 0, as_stream(stream), gp, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev);
        else if (dtype == PERF_DTYPE_FP16)
            hipLaunchKernelGGL(hashgrid_fwd_v2_kernel<FP16>, g, b, 0, as_stream(stream), gp, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev);
        else { set_error("perf_hashgrid_fwd: bad dtype %d", dtype); return PERF_E_INVALID; }
        PERF_LAUNCH_CHECK("perf_hashgrid_fwd");
        return PERF_OK;
    }
    dim3 g(xcd_affinity ? (unsigned)(chunks * 8) : (unsigned)(div_up(div_up(n, 256), 8) * 64)), b(256);
    if (dtype == PERF_DTYPE_BF16)
        hipLaunchKernelGGL(hashgrid_fwd_kernel<BF16>, g, b, 0, as_stream(stream), gp, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev, xcd_affinity);
    else if (dtype == PERF_DTYPE_FP16)
        hipLaunchKernelGGL(hashgrid_fwd_kernel<FP16>, g, b, 0, as_stream(stream), gp, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev, xcd_affinity);
    else { set_error("perf_hashgrid_fwd: bad dtype %d", dtype); return PERF_E_INVALID; }
    PERF_LAUNCH_CHECK("perf_hashgrid_fwd");
    return PERF_OK;
}

extern "C" int perf_hashgrid_fwd2(const perf_grid_desc* grid, const float* x01, const void* table16_a,
                                  const void* table16_b, void* feat16_a, void* feat16_b, int64_t n, int dtype,
                                  void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && table16_a && table16_b && feat16_a && feat16_b, "NULL pointer");
    dim3 g(grouped_grid(n)), b(256);
    if (dtype == PERF_DTYPE_BF16)
        hipLaunchKernelGGL(hashgrid_fwd2_kernel<BF16>, g, b, 0, as_stream(stream), gp, x01, (const uint32_t*)table16_a,
                           (const uint32_t*)table16_b, (uint32_t*)feat16_a, (uint32_t*)feat16_b, n);
    else if (dtype == PERF_DTYPE_FP16)
        hipLaunchKernelGGL(hashgrid_fwd2_kernel<FP16>, g, b, 0, as_stream(stream), gp, x01, (const uint32_t*)table16_a,
                           (const uint32_t*)table16_b, (uint32_t*)feat16_a, (uint32_t*)feat16_b, n);
    else { set_error("perf_hashgrid_fwd2: bad dtype %d", dtype); return PERF_E_INVALID; }
    PERF_LAUNCH_CHECK("perf_hashgrid_fwd2");
    return PERF_OK;
}

extern "C" int perf_hashgrid_corners(const perf_grid_desc* grid, const float* x01, int32_t* idx, int64_t n, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && idx, "NULL pointer");
    PERF_REQUIRE(gp.offset[gp.n_levels - 1] + gp.size[gp.n_levels - 1] < ((uint64_t)1 << 31), "perf_hashgrid_corners: table too large for int32 entries");
    hipLaunchKernelGGL(hashgrid_corners_kernel, dim3((unsigned)div_up(n, 256), gp.n_levels), dim3(256), 0, as_stream(stream), gp,
                       x01, idx, n);
    PERF_LAUNCH_CHECK("perf_hashgrid_corners");
    return PERF_OK;
}

extern "C" int perf_hashgrid_fwd_f32(const perf_grid_desc* grid, const float* x01, const float* table,
                                     float* feat, int64_t n, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && table && feat, "NULL pointer");
    hipLaunchKernelGGL(hashgrid_fwd_f32_kernel, dim3(grouped_grid(n)), dim3(256), 0, as_stream(stream), gp, x01,
                       (const float2*)table, (float2*)feat, n);
    PERF_LAUNCH_CHECK("perf_hashgrid_fwd_f32");
    return PERF_OK;
}

// levels whose owners can run the coded variant (multi-tile levels); returns their number
static int plan_codes(const GridParams& gp, int64_t n, TileParams* tp) {
    int slots = 0;
    for (int l = 0; l < PERF_MAX_LEVELS; ++l) {
        tp->code_slot[l] = -1;
        if (l >= gp.n_levels || n >= kMaxCodedSamples || ((tp->bitmap_levels >> l) & 1u)) continue;
        const int64_t nt = tp->tiles_of[l];         // (plan_tiles ran before)
        if (gp.hashed[l] ? (nt >= 2 && nt <= 255 && gp.res[l] + 2u < (uint32_t)kTileEntries) : (nt >= 2 && nt <= 64)) {
            tp->code_slot[l] = slots++;
        }
    }
    tp->n_pad = (n + 3) & ~(int64_t)3;
    return slots;
}

constexpr int64_t kShiftBytes = 256;        // per-level shifts the owners leave for the replica reduction

// bitmap rows of the levels plan_tiles marked (bitmap_levels); returns the bytes of [bitmaps][escape words] or 0
constexpr int64_t kBitmapMaxBytes = (int64_t)2 << 30;
static int64_t plan_bitmaps(const GridParams& gp, int64_t n, TileParams* tp, int* n_levels_out) {
    int rows = 0, idx = 0;
    for (int l = 0; l < PERF_MAX_LEVELS; ++l) {
        tp->bm_row[l] = -1; tp->bm_idx[l] = -1;
        if (l < gp.n_levels && ((tp->bitmap_levels >> l) & 1u)) { tp->bm_row[l] = rows; tp->bm_idx[l] = idx++; rows += tp->tiles_of[l]; }
    }
    int nt_max = 1;
    for (int l = 0; l < gp.n_levels; ++l) if ((tp->bitmap_levels >> l) & 1u) nt_max = tp->tiles_of[l] > nt_max ? tp->tiles_of[l] : nt_max;
    int samples = 1024;
    while (samples > 256 && (int64_t)nt_max * samples / 8 > 65536) samples >>= 1;
    tp->bm_samples = samples;
    tp->bm_blocks = div_up(n, samples);
    tp->bm_row_words = tp->bm_blocks * (samples / 32);
    *n_levels_out = idx;
    return (int64_t)rows * tp->bm_row_words * 4 + (int64_t)idx * tp->bm_blocks * 4;
}

extern "C" int64_t perf_hashgrid_bwd_workspace_bytes(const perf_grid_desc* grid, int64_t n) {
    GridParams gp;
    if (fill_params(grid, &gp)) return -1;
    TileParams tp; int nb; int64_t ws;
    int64_t ws2;
    plan_tiles(gp, false, &tp, &nb, &ws);
    plan_tiles(gp, true, &tp, &nb, &ws2, (n > 0 && n < kMaxCodedSamples) ? kBitmapMaxTiles : 0);
    const int slots = plan_codes(gp, n, &tp);
    int bm_levels = 0;
    int64_t bm_bytes = plan_bitmaps(gp, n, &tp, &bm_levels);
    if (bm_bytes > kBitmapMaxBytes) bm_bytes = 0;
    return (ws > ws2 ? ws : ws2) * (int64_t)sizeof(float2) + 16 + kShiftBytes + (int64_t)slots * tp.n_pad * 4 +
           (slots ? div_up(n, kCodeSamplesPerBlock) * 4 : 0) + (bm_bytes ? bm_bytes + 16 : 0);
}

extern "C" int perf_hashgrid_bwd(const perf_grid_desc* grid, const float* x01, const float* dfeat,
                                 float* grad_table, int64_t n, const int64_t* n_dev, int accumulate, const float* level_absmax,
                                 int32_t* overflow_flag, int32_t* headroom_state, const int32_t* shifts_dev, int raw_fields,
                                 const int32_t* redo_flag, void* workspace, int64_t workspace_bytes, void* stream) {
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    PERF_REQUIRE(grad_table, "NULL pointer");
    PERF_REQUIRE(n == 0 || (x01 && dfeat), "NULL pointer");
    const bool fixed = level_absmax != nullptr || shifts_dev != nullptr;
    if (redo_flag) {
        // The repair of a fixed-point call whose fields overflowed: ONE launch, predicated on the device flag, fp32 LDS
        // accumulation, every tile owned by a single workgroup that streams positions (no pre-pass, no replica reduction: the
        // slow-but-simple owners; the launch is a no-op dispatch in all but a handful of steps per million).
        PERF_REQUIRE(!fixed && !accumulate && !raw_fields, "perf_hashgrid_bwd: a redo call is an fp32, overwriting call");
        TileParams rp;
        int nb = 0;
        int64_t wse = 0;
        plan_tiles(gp, false, &rp, &nb, &wse, 0, true);
        if (rp.atomic_levels != 0u || nb == 0) { set_error("perf_hashgrid_bwd: the redo launch serves grids whose levels all fit LDS owners (<= 255 hashed / 64 dense tiles)"); return PERF_E_UNSUPPORTED; }
        rp.accumulate = 0; rp.raw_out = 0; rp.run_merge = 0; rp.n_pad = 0;
        for (int l = 0; l < PERF_MAX_LEVELS; ++l) rp.code_slot[l] = -1;
        const int lds_b = 2 * kTileEntries * (int)sizeof(float) + (kBwdThreads / 64) * kQueueCap * (int)sizeof(uint32_t);
        static std::once_flag redo_once;
        std::call_once(redo_once, [&]() {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hashgrid_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
        });
        if (n > 0)
            hashgrid_bwd_kernel<false><<<dim3(nb), dim3(kBwdThreads), lds_b, as_stream(stream)>>>(
                gp, rp, x01, (const float2*)dfeat, (float2*)grad_table, nullptr, nullptr, nullptr, headroom_state, nullptr, nullptr,
                nullptr, nullptr, nullptr, nullptr, n, n_dev, redo_flag);
        PERF_LAUNCH_CHECK("perf_hashgrid_bwd(redo)");
        return PERF_OK;
    }
    PERF_REQUIRE(!shifts_dev || !headroom_state, "perf_hashgrid_bwd: given units (shifts_dev) exclude the headroom feedback");
    PERF_REQUIRE(!raw_fields || (fixed && !accumulate), "perf_hashgrid_bwd: raw fields need the fixed-point mode and accumulate == 0");
    TileParams tp;
    int n_blocks = 0;
    int64_t ws_entries = 0;
    const bool aligned_ws = (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
    // workspace layout: [replica slabs (larger of both modes)][shifts][tile codes][escape words][bitmaps][their escape words]
    // (a workspace WITHOUT room for the codes / bitmaps selects the position-streaming owners / the global-atomics scatter: how the
    //  tests reach those paths)
    const bool want_bitmaps = n > 0 && n < kMaxCodedSamples && aligned_ws;
    int64_t slab_entries = 0;       // (the largest of the plans a call may end up with: the offsets below must not depend on the choice)
    { TileParams t2; int nb2; int64_t w2;
      plan_tiles(gp, fixed, &t2, &nb2, &w2); slab_entries = w2;
      plan_tiles(gp, !fixed, &t2, &nb2, &w2); if (w2 > slab_entries) slab_entries = w2;
      if (want_bitmaps) { plan_tiles(gp, fixed, &t2, &nb2, &w2, kBitmapMaxTiles); if (w2 > slab_entries) slab_entries = w2; } }
    const int64_t shifts_at = (slab_entries * (int64_t)sizeof(float2) + 15) & ~(int64_t)15;
    const int64_t codes_at = shifts_at + kShiftBytes;
    const int64_t esc_words = div_up(n, kCodeSamplesPerBlock);
    // levels of 256..2048 tiles take LDS owners fed by per-tile bitmaps when the workspace holds the bitmaps, global atomics otherwise
    int bitmap_tiles = 0, bm_levels = 0;
    int64_t bits_at = 0, bm_bytes = 0;
    if (want_bitmaps) {
        TileParams t0; int nb0; int64_t w0;
        plan_tiles(gp, fixed, &t0, &nb0, &w0, kBitmapMaxTiles);
        if (t0.bitmap_levels) {
            const int slots0 = plan_codes(gp, n, &t0);
            bm_bytes = plan_bitmaps(gp, n, &t0, &bm_levels);
            bits_at = (codes_at + (int64_t)slots0 * t0.n_pad * 4 + (slots0 ? esc_words * 4 : 0) + 15) & ~(int64_t)15;
            if (bm_bytes <= kBitmapMaxBytes && workspace_bytes >= bits_at + bm_bytes) bitmap_tiles = kBitmapMaxTiles;
        }
    }
    plan_tiles(gp, fixed, &tp, &n_blocks, &ws_entries, bitmap_tiles);
    PERF_REQUIRE(!(raw_fields || shifts_dev) || tp.atomic_levels == 0u,
                 "perf_hashgrid_bwd: raw fields / given units are not available for levels on the global-atomics scatter (more than 2048 tiles, or no room for the per-tile bitmaps in the workspace)");
    tp.run_merge = 1;
    tp.accumulate = accumulate;
    tp.raw_out = raw_fields ? 1 : 0;
    PERF_REQUIRE(workspace && workspace_bytes >= shifts_at + kShiftBytes,
                 "perf_hashgrid_bwd: workspace too small (need %lld bytes)", (long long)(shifts_at + kShiftBytes));
    int32_t* shifts_ws = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + shifts_at);
    // tile codes of the multi-tile levels (workspace permitting)
    const int slots = plan_codes(gp, n, &tp);
    uint32_t* codes = nullptr;
    uint32_t* escape = nullptr;
    uint32_t* bitmaps = nullptr;
    uint32_t* esc_bm = nullptr;
    if (tp.bitmap_levels) {
        (void)plan_bitmaps(gp, n, &tp, &bm_levels);
        bitmaps = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + bits_at);
        int rows = 0, nt_max = 0;
        for (int l = 0; l < gp.n_levels; ++l)
            if ((tp.bitmap_levels >> l) & 1u) { rows += tp.tiles_of[l]; nt_max = tp.tiles_of[l] > nt_max ? tp.tiles_of[l] : nt_max; }
        esc_bm = bitmaps + (int64_t)rows * tp.bm_row_words;
        static std::once_flag bm_once;
        std::call_once(bm_once, [&]() {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_bitmap_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        });
        tile_bitmap_kernel<<<dim3((unsigned)tp.bm_blocks, (unsigned)bm_levels), dim3(256), (size_t)nt_max * (tp.bm_samples / 8), as_stream(stream)>>>(
            gp, tp, x01, (const float2*)dfeat, bitmaps, esc_bm, n, n_dev);
        PERF_LAUNCH_CHECK("perf_hashgrid_bwd(bitmaps)");
    }
    if (slots > 0 && n > 0 && aligned_ws &&
        workspace_bytes >= codes_at + (int64_t)slots * tp.n_pad * 4 + esc_words * 4) {
        codes = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + codes_at);
        escape = codes + (int64_t)slots * tp.n_pad;
        if ((reinterpret_cast<uintptr_t>(codes) & 15) == 0)
            tile_codes4_kernel<<<dim3((unsigned)div_up(esc_words, 4)), dim3(256), 0, as_stream(stream)>>>(gp, tp, x01, (const float2*)dfeat,
                                                                                                            codes, escape, n, esc_words, n_dev);
        else
            tile_codes_kernel<<<dim3((unsigned)esc_words), dim3(256), 0, as_stream(stream)>>>(gp, tp, x01, (const float2*)dfeat,
                                                                                                 codes, escape, n, n_dev);
        PERF_LAUNCH_CHECK("perf_hashgrid_bwd(codes)");
    } else {
        for (int l = 0; l < PERF_MAX_LEVELS; ++l) tp.code_slot[l] = -1;
    }
    const int lds_bytes = 2 * kTileEntries * (int)sizeof(float) + (kBwdThreads / 64) * kQueueCap * (int)sizeof(uint32_t);
    static std::once_flag attr_once;                // one-time kernel attribute setup, safe under concurrent callers
    std::call_once(attr_once, [&]() {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hashgrid_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hashgrid_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    });
    if (n_blocks == 0) {
        // every level goes through the atomics fallback
    } else if (fixed)
        hashgrid_bwd_kernel<true><<<dim3(n_blocks), dim3(kBwdThreads), lds_bytes, as_stream(stream)>>>(
            gp, tp, x01, (const float2*)dfeat, (float2*)grad_table, (float2*)workspace, level_absmax, overflow_flag, headroom_state,
            shifts_dev, shifts_ws, codes, escape, bitmaps, esc_bm, n, n_dev, nullptr);
    else
        hashgrid_bwd_kernel<false><<<dim3(n_blocks), dim3(kBwdThreads), lds_bytes, as_stream(stream)>>>(
            gp, tp, x01, (const float2*)dfeat, (float2*)grad_table, (float2*)workspace, nullptr, nullptr, nullptr, nullptr, nullptr,
            codes, escape, bitmaps, esc_bm, n, n_dev, nullptr);
    PERF_LAUNCH_CHECK("perf_hashgrid_bwd");
    if (tp.atomic_levels && n > 0) {
        if (!accumulate)
            for (int l = 0; l < gp.n_levels; ++l)
                if ((tp.atomic_levels >> l) & 1u)
                    PERF_REQUIRE(hipMemsetAsync(grad_table + 2 * gp.offset[l], 0, (size_t)gp.size[l] * 2 * sizeof(float), as_stream(stream)) == hipSuccess,
                                 "perf_hashgrid_bwd: memset failed");
        if (level_absmax && !accumulate && (reinterpret_cast<uintptr_t>(grad_table) & 7) == 0) {      // fixed point: one 64-bit atomic per corner, order independent
            hashgrid_bwd_atomic_fixed_kernel<<<dim3((unsigned)div_up(n, 256), gp.n_levels), dim3(256), 0, as_stream(stream)>>>(
                gp, tp.atomic_levels, x01, (const float2*)dfeat, grad_table, level_absmax, headroom_state, n, n_dev);
            PERF_LAUNCH_CHECK("perf_hashgrid_bwd(atomics, fixed point)");
            hashgrid_bwd_unfix_kernel<<<dim3(1024, gp.n_levels), dim3(256), 0, as_stream(stream)>>>(
                gp, tp.atomic_levels, grad_table, level_absmax, headroom_state, overflow_flag, n, n_dev);
            PERF_LAUNCH_CHECK("perf_hashgrid_bwd(unfix)");
        } else {
            hashgrid_bwd_atomic_kernel<<<dim3((unsigned)div_up(n, 256), gp.n_levels), dim3(256), 0, as_stream(stream)>>>(
                gp, tp.atomic_levels, x01, (const float2*)dfeat, grad_table, n, n_dev);
            PERF_LAUNCH_CHECK("perf_hashgrid_bwd(atomics)");
        }
    } else if (tp.atomic_levels && !accumulate) {
        for (int l = 0; l < gp.n_levels; ++l)
            if ((tp.atomic_levels >> l) & 1u) (void)hipMemsetAsync(grad_table + 2 * gp.offset[l], 0, (size_t)gp.size[l] * 2 * sizeof(float), as_stream(stream));
    }
    const bool adapt = level_absmax && headroom_state && (n_blocks > 0 || (tp.atomic_levels && n > 0 && !accumulate));
    if (ws_entries > 0 || adapt) {      // replica sums, and the headroom feedback by the last workgroup
        constexpr int kReduceBlocks = 64;
        int n_tickets = 0;
        for (int l = 0; l < gp.n_levels; ++l) n_tickets += tp.replicas_of[l] > 1 ? reduce_blocks_of(gp.size[l], kReduceBlocks) : 0;
        hashgrid_bwd_reduce_kernel<<<dim3(kReduceBlocks, gp.n_levels), dim3(256), 0, as_stream(stream)>>>(
            gp, tp, (const float2*)workspace, (float2*)grad_table, adapt ? headroom_state : nullptr,
            shifts_dev ? shifts_dev : shifts_ws, fixed ? 1 : 0, overflow_flag, n_tickets);
        PERF_LAUNCH_CHECK("perf_hashgrid_bwd(reduce)");
    }
    return PERF_OK;
}

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
