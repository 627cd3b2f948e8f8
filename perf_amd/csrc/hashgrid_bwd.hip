// Hash-grid parameter gradient for gfx950 (tcnn kernel_grid_backward semantics): LDS tile owners, no global atomics.
#include <stdlib.h>
#include <mutex>
#include "common.hpp"
#include "grid_device.hpp"
#include "grid_fixed_point.hpp"
#include "mlp_reduce_device.hpp"
#include "step_book_device.hpp"

namespace perf {

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
constexpr int kCus = 248;                  // MI355X has 256 CUs and an owner workgroup (128 KiB of LDS) takes one -- but a launch of 252 did not get
                                           // one each: 731 us against 361 (call r06cu); 31 per XCD is what is asked for at most

// Load balance: a hashed level has 16 tiles, each receiving 1/16 of the level's 8 corner updates per
// sample; a coarse dense level has only 1..8 tiles receiving the same total.  Coarse tiles are therefore
// REPLICATED (R_l copies, each streaming 1/R_l of the samples) so that every workgroup takes about as long as
// a hashed-tile owner; replicas are summed by a small second kernel.  L16/T18, fixed-point mode:
// 8 + 8 + 4x3 + 8x3 + 12x16 = 244 workgroups (round 6: the third replica of level 3 from the CUs left over, see plan_tiles).
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
    // ---- CUs left over go to the dense levels of 5..16 tiles (round 6, per-workgroup times of a 1 M-sample call: the 8 tiles x 2
    // replicas of level 3 of L16 / T18 took 357 us each, every hashed owner 331-350 -- the launch ended with them; 16 of the 20 idle CUs
    // as two more replicas bring them to ~190).  Only while the launch still fits one workgroup per CU, and only in fixed-point mode
    // (integer slabs: the sums do not depend on how the samples are split).
    if (fixed && !no_replicas && !large_grid && tp->bitmap_levels == 0u && tp->atomic_levels == 0u) {
        for (int want = 4; want >= 3; --want)
            for (int l = 0; l < gp.n_levels; ++l) {
                const int nt = tp->tiles_of[l], r = tp->replicas_of[l];
                if (gp.hashed[l] || nt <= 4 || nt > 16 || r != rs[2] || want <= r) continue;
                if (nb + nt * (want - r) > kCus) continue;
                nb += nt * (want - r);
                tp->replicas_of[l] = want;
            }
        // (a fourth replica for the 4 tiles of level 2 -- 336 -> 269 us per workgroup, 248 workgroups -- made the launch SLOWER in an A/B on
        //  one box: 357.0 against 352.5 us, three runs each, call r06dd: the hashed owners end the launch and lose more to four more
        //  neighbours than level 2 gains)
        ws = 0;                                              // (slab offsets again)
        for (int l = 0; l < gp.n_levels; ++l) {
            tp->ws_off[l] = 0;
            if (tp->replicas_of[l] > 1) { tp->ws_off[l] = ws; ws += (int64_t)tp->replicas_of[l] * gp.size[l]; }
        }
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

// one workgroup = the 256 samples of one escape word, a sample per thread
__device__ __forceinline__ void tile_codes_block(const GridParams& gp, const TileParams& tp, const float* __restrict__ x01,
                                                 const float2* __restrict__ dfeat, uint32_t* __restrict__ codes,
                                                 uint32_t* __restrict__ escape, int64_t n, int64_t n_live) {
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

// (both pre-pass kernels can carry the MLP backward's second stage in workgroups of their own behind the `own_blocks` that compute
//  codes: mlp_reduce_device.hpp)
__global__ __launch_bounds__(256) void tile_codes_kernel(GridParams gp, TileParams tp, const float* __restrict__ x01,
                                                         const float2* __restrict__ dfeat, uint32_t* __restrict__ codes,
                                                         uint32_t* __restrict__ escape, int64_t n,
                                                         const int64_t* __restrict__ n_dev, MlpReduceJob job, unsigned own_blocks) {
    if (blockIdx.x >= own_blocks) { mlp_reduce_block(job, (int)(blockIdx.x - own_blocks)); return; }
    tile_codes_block(gp, tp, x01, dfeat, codes, escape, n, live_count(n, n_dev));       // n: capacity = stride of dfeat / codes
}

// The same codes, four consecutive samples per thread: 48 bytes of positions in three 16-byte loads and one 16-byte store per
// level instead of four 4-byte ones (the byte-code kernel spent most of a wave's life queueing stores: 16 per sample).  A wave
// covers the 256 samples of one escape word.  (The kernel above serves workspaces whose code rows are not 16-byte aligned.)
// The launch has one workgroup per escape word; with many live samples only the first quarter of them works (four words each).
// With FEW live samples (the reference-faithful training step: 16-35 k live rows of a 1 M-row capacity) four samples per thread
// leave a handful of waves walking 4 x 16 levels each -- 13.5 us against 7 for the one-sample-per-thread shape, which those
// launches therefore take (the count is on the device: the shape is chosen here, not by the host).
constexpr int64_t kCodes4MinLive = 65536;
__global__ __launch_bounds__(256) void tile_codes4_kernel(GridParams gp, TileParams tp, const float* __restrict__ x01,
                                                          const float2* __restrict__ dfeat, uint32_t* __restrict__ codes,
                                                          uint32_t* __restrict__ escape, int64_t n, int64_t n_words,
                                                          const int64_t* __restrict__ n_dev, MlpReduceJob job, unsigned own_blocks) {
    if (blockIdx.x >= own_blocks) { mlp_reduce_block(job, (int)(blockIdx.x - own_blocks)); return; }
    const int64_t n_live = live_count(n, n_dev);
    if (n_live < kCodes4MinLive) {                         // (uniform)
        tile_codes_block(gp, tp, x01, dfeat, codes, escape, n, n_live);
        return;
    }
    if ((int64_t)blockIdx.x * 4 >= n_words) return;
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
    // Hashed levels: which of a code's four bytes name this tile, as 0x80 in each such byte -- x = code ^ (t in every byte) has a zero byte
    // there, and ~(((x & 0x7f..) + 0x7f..) | x | 0x7f..) marks zero bytes exactly (no carry leaves a byte).  Five vector instructions per
    // code (gfx950's three-input bit operation folds the AND and the NOR) where four byte compares, four selects and two ORs took eleven; the
    // four bits of the queue entry are made from the marks by ONE dot product with (1, 2, 4, 8), and only on the lanes that hit.
    // (t4: this tile in every byte; a lane without a group of its own in the last iteration compares against 0xff.., which is no tile of
    //  a hashed level -- at most 255 of them -- so that validity costs one select per iteration instead of one per code)
    const uint32_t t4 = cx.t * 0x01010101u;
    auto test_marks = [&](uint32_t code, uint32_t t4) {
        const uint32_t x = code ^ t4;
        const uint32_t a = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
        return ~(a | x | 0x7f7f7f7fu);
    };
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
    auto enqueue_marks = [&](uint32_t marks, uint32_t i) {        // (hashed levels: `marks` of test_marks)
        const unsigned long long b = __ballot(marks != 0u);
        if (b) {
            const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            if (marks) queue[pos] = (i << 4) | __builtin_amdgcn_udot4(marks >> 7, 0x08040201u, 0u, false);
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
            const uint32_t t4v = valid ? t4 : 0xffffffffu;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (DENSE) enqueue(valid ? test(cs[s]) : 0u, (uint32_t)(4 * g0 + s));
                else enqueue_marks(test_marks(cs[s], t4v), (uint32_t)(4 * g0 + s));
            }
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
        if (i < n) cm = DENSE ? test(codes_l[i]) : test_marks(codes_l[i], t4);
        if (DENSE) enqueue(cm, (uint32_t)i);
        else enqueue_marks(cm, (uint32_t)i);
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
                                                                   const int32_t* __restrict__ redo_flag, perf_step_book book, int has_book) {
    // a predicated REDO launch (perf_hashgrid_bwd, redo_flag): nothing happens unless the fixed-point call before it raised
    // the flag -- the graph node costs a dispatch, the gradient table is left as that call wrote it.  The dispatch exists in every
    // training step: ONE of its threads does the step's bookkeeping on the way (perf_field_bwd_book) -- everything but clearing the
    // flag, which every workgroup of this launch reads right here (perf_adam_step_dev's clear_flag consumes it afterwards)
    if constexpr (!FIXED) {
        if (redo_flag && has_book && blockIdx.x == 0 && threadIdx.x == 0) step_bookkeeping_thread(book, false);
    }
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
#ifdef PERF_BWD_BLOCK_TIMES         // tools/exp/bwd_block_times.py: the launch is as long as its SLOWEST workgroup -- which one is it?
    const long long block_t0 = wall_clock64();
#endif
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
#ifdef PERF_BWD_BLOCK_TIMES         // (two workgroups per level report: a printf from every workgroup stretches the launch it measures)
    __syncthreads();
    if (FIXED && threadIdx.x == 0 && (t == 0u || t == n_tiles / 2u))
        printf("BLOCKT level %d tile %u rep %d of %d ticks %lld\n", l, t, rep, R, wall_clock64() - block_t0);
#endif
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
    if (FIXED && hr_state && R == 1) {  // largest |field| of the level, for the feedback (ONE atomic per workgroup; replicated
                                        // levels report the max of their SUMMED fields from the reduction kernel)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) field_max = max(field_max, __shfl_xor(field_max, off));
        // The 24 maxima share a cache line, and read-modify-writes on one line retire one at a time, ~11 ns each.  A wave only
        // needs one when it RAISES the maximum -- but the look does not help when everybody finishes together: with a few
        // thousand live samples (the reference-faithful step) all 192 hashed owners end within a microsecond, their 3,072 waves
        // all read the stale zero and all queue: the launch took 41 us at 1,024 live samples against 12 us at none (rocprofv3,
        // tools/exp/bwd_live_sweep.py).  The workgroup's waves therefore agree on ONE value first (LDS) -- at most 192 atomics.
        __shared__ int32_t wave_max[kBwdThreads / 64];
        if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = field_max;
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t m = 0;
#pragma unroll
            for (int w = 0; w < kBwdThreads / 64; ++w) m = max(m, wave_max[w]);
            if (m > 0 && m > __hip_atomic_load(&hr_state[PERF_MAX_LEVELS + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(&hr_state[PERF_MAX_LEVELS + l], m);
        }
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

}  // namespace perf

using namespace perf;

namespace perf {
__global__ void step_book_keep_flag_kernel(perf_step_book b) { step_bookkeeping_thread(b, false); }
}  // namespace perf
static void launch_step_book_keep_flag(const perf_step_book& b, void* stream) {
    hipLaunchKernelGGL(perf::step_book_keep_flag_kernel, dim3(1), dim3(1), 0, as_stream(stream), b);
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
    if (n == 0) {       // (the minimal workspace -- no codes, no bitmaps -- must still hold the replica slabs of whatever plan a call picks)
        TileParams t3; int nb3; int64_t ws3;
        plan_tiles(gp, true, &t3, &nb3, &ws3, kBitmapMaxTiles);
        if (ws3 > ws2) ws2 = ws3;
    }
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
    return perf_internal_hashgrid_bwd(grid, x01, dfeat, grad_table, n, n_dev, accumulate, level_absmax, overflow_flag, headroom_state, shifts_dev,
                                      raw_fields, redo_flag, workspace, workspace_bytes, stream, nullptr, nullptr);
}

// job (perf_field_bwd): the deferred second stage of the MLP backward that produced `dfeat` / `level_absmax`; it rides in the
// tile-code launch when there is one and gets a launch of its own otherwise -- in either case BEFORE the owners, which read level_absmax
int perf_internal_hashgrid_bwd(const perf_grid_desc* grid, const float* x01, const float* dfeat,
                               float* grad_table, int64_t n, const int64_t* n_dev, int accumulate, const float* level_absmax,
                               int32_t* overflow_flag, int32_t* headroom_state, const int32_t* shifts_dev, int raw_fields,
                               const int32_t* redo_flag, void* workspace, int64_t workspace_bytes, void* stream, const MlpReduceJob* job_in,
                               const perf_step_book* book) {
    PERF_REQUIRE(!book || redo_flag, "perf_hashgrid_bwd: the bookkeeping rides in a repair launch only");
    MlpReduceJob job{};
    if (job_in) job = *job_in;
    GridParams gp;
    int rc = fill_params(grid, &gp);
    if (rc) return rc;
    PERF_REQUIRE(!(redo_flag && job.n_blocks), "perf_hashgrid_bwd: a redo call carries no deferred work");
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
                nullptr, nullptr, nullptr, nullptr, n, n_dev, redo_flag, book ? *book : perf_step_book{}, book ? 1 : 0);
        else if (book)      // (no repair launch to ride in: the bookkeeping's own launch, the flag left for the caller all the same)
            launch_step_book_keep_flag(*book, stream);
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
        const unsigned own = (unsigned)esc_words, all = own + (unsigned)job.n_blocks;
        if ((reinterpret_cast<uintptr_t>(codes) & 15) == 0)
            tile_codes4_kernel<<<dim3(all), dim3(256), 0, as_stream(stream)>>>(gp, tp, x01, (const float2*)dfeat, codes, escape, n, esc_words, n_dev,
                                                                                job, own);
        else
            tile_codes_kernel<<<dim3(all), dim3(256), 0, as_stream(stream)>>>(gp, tp, x01, (const float2*)dfeat, codes, escape, n, n_dev, job, own);
        PERF_LAUNCH_CHECK("perf_hashgrid_bwd(codes)");
        job.n_blocks = 0;                          // (done)
    } else {
        for (int l = 0; l < PERF_MAX_LEVELS; ++l) tp.code_slot[l] = -1;
    }
    if (job.n_blocks) {                            // no tile-code launch to ride in
        perf_internal_launch_mlp_reduce(job, stream);
        PERF_LAUNCH_CHECK("perf_hashgrid_bwd(deferred MLP reduce)");
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
            shifts_dev, shifts_ws, codes, escape, bitmaps, esc_bm, n, n_dev, nullptr, perf_step_book{}, 0);
    else
        hashgrid_bwd_kernel<false><<<dim3(n_blocks), dim3(kBwdThreads), lds_bytes, as_stream(stream)>>>(
            gp, tp, x01, (const float2*)dfeat, (float2*)grad_table, (float2*)workspace, nullptr, nullptr, nullptr, nullptr, nullptr,
            codes, escape, bitmaps, esc_bm, n, n_dev, nullptr, perf_step_book{}, 0);
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
