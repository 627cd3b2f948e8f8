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
//
// This unit: the forward encode (perf_hashgrid_fwd / _fwd2 / _corners / _fwd_f32).  The parameter gradient is hashgrid_bwd.hip;
// input gradient, second order and the data-parallel unit / slot kernels are hashgrid_aux.hip.
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "grid_device.hpp"

namespace perf {


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


// ---- forward encode of DEEP grids (L > 16: BASELINE config 5's L = 20 tables sized to HBM) and of line-local tables ------------
// The tables of such a grid exceed every cache, so there is nothing to pin: what the level-group kernel above pins to an XCD --
// levels {g, 15-g, 16+g} one after the other -- leaves XCDs 0-3 with two HBM-bound levels each and XCDs 4-7 with none (measured,
// tools/exp/c5_encode_probe.hip, 4.2 M panorama samples: 2.36 ms at T = 2^28, 5.29 ms at 2^30).  Here a workgroup is ONE level of
// 256 x STEPS consecutive samples, and the work items are dealt so that
//   * XCD x = b % 8 serves level l for a TURN of consecutive samples (4,096: the neighbouring rays of a panorama batch) back to
//     back: their gathers meet in one L2 (a level's consecutive rays share cells at the coarse levels and 128-byte lines at the fine ones);
//   * which eighth of a stripe of eight turns an XCD serves rotates with the level ((x - l) & 7): every XCD serves every level for an
//     eighth of the samples -- balance whatever a level costs.                                      1.50 / 2.37 ms, tcnn layout
// Line-local levels (PERF_LAYOUT_LINE_LOCAL) give every sample FOUR LANES, one per (y, z) corner pair; a lane fetches the ALIGNED
// 16-byte x-run that holds its pair's first vertex -- both x corners unless the cell starts at a block's last vertex (a quarter of
// the samples: a 4-byte gather for those).  The four runs of a sample -- 2.3 lines on average -- are thus requested by ONE
// instruction (the texture addresser merges lanes that name the same line) instead of by four consecutive ones that find the line
// pending; a wave serves 16 samples per instruction and keeps four such groups in flight; the four partial sums of a sample meet
// through two DPP quad permutes.  Counters of the one-lane-per-sample form (profiles/r06_config5_counters.json): 53 % of the wave
// cycles waiting to ISSUE a memory instruction, the L1 stalled on pending lines 80 % of the time.
// Waves are LONG-LIVED: a wave serves 64 x STEPS samples in STEPS steps and requests the coordinates of the next step while the table
// lines of this one are in flight (a wave that serves 64 samples and retires spends a third of its life on start-up + the coordinate
// fetch with no table line requested: 0.89 ms per 4.2 M samples at T = 2^28 line-local; four steps: 0.74 ms).  Measured and not kept
// (profiles/r06_config5_long_waves.json): a ring of request slots refilled group by group (3-4 groups in flight at all times, but 95
// registers = 5 waves per SIMD: 0.90 ms; two slots at 7 waves: 0.80), eight waves forced with 5 spilled registers (0.79).
// OVERLAPPING RUNS (PERF_LAYOUT_LINE_OVERLAP, grid_device.hpp:overlap_x): the x corner pair of a cell is always inside one run, so a lane
// requests exactly its pair -- 8 bytes -- and only the last cell of a super-block row needs a second request: 0.705 -> 0.62 ms per 4.2 M
// samples at T = 2^28 (56 registers: eight waves per SIMD); with the coarse levels in the same launch 0.59-0.60.
// Placement is a speed assumption only; results do not depend on it.
constexpr int64_t kBigMaxStripes = 1 << 16;

__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    return v;
}

// The two kinds of level are two device functions with their own launches (a grid of one kind only) and one launch that holds both
// (hashgrid_fwd_big_mixed_kernel: 59-72 registers).
struct BigLevels { int32_t count; int32_t order; int64_t stripes; int32_t level[PERF_MAX_LEVELS]; };

// work item of workgroup b: XCD x = b % 8, its turn's sub-group, the level and the first stripe
struct BigItem { int xcd, sub, l; int64_t q0; };
__device__ __forceinline__ BigItem big_item(const BigLevels& lv, int bpt) {
    BigItem it;
    it.xcd = (int)(blockIdx.x & 7u);
    const int64_t j = (int64_t)(blockIdx.x >> 3);
    it.sub = (int)(j % bpt);                                     // a turn of an XCD = bpt workgroups = bpt x 256 x STEPS consecutive samples
    const int64_t t = j / bpt;
    if (lv.order == 1) {                                         // level-major: every turn of a level, then the next level
        it.l = lv.level[(int)(t / lv.stripes)];
        it.q0 = t % lv.stripes;
    } else {
        it.l = lv.level[(int)(t % lv.count)];
        it.q0 = t / lv.count;
    }
    return it;
}

typedef uint32_t uint2_a4 __attribute__((ext_vector_type(2), aligned(4)));   // (an 8-byte load from a 4-byte aligned address: all gfx950's global loads need)

template <typename T16, int STEPS>
__device__ __forceinline__ void big_gather_body(const GridParams& gp, const BigItem& w, const float* __restrict__ x01,
                                                const uint32_t* __restrict__ table, uint32_t* __restrict__ feat,
                                                int64_t n, int64_t n_live, int64_t grid_stripes, int bpt) {
    const int l = w.l;
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
    const uint32_t* tl = table + gp.offset[l];
    const bool dense = gp.hashed[l] == 0;
    const float sc = gp.scale[l];
    const uint32_t res = gp.res[l], r2 = res * res, size = gp.size[l];
    constexpr int64_t kGroup = 256 * STEPS;
    const int64_t stripe = 8 * (int64_t)bpt * kGroup;            // samples of one turn of all eight XCDs
    for (int64_t Q = w.q0; Q * stripe < n_live; Q += grid_stripes) {
        const int64_t base = (((Q << 3) + ((w.xcd - l) & 7)) * bpt + w.sub) * kGroup + threadIdx.x;
        if (base - threadIdx.x >= n_live) continue;
        int64_t i0 = base < n_live ? base : n_live - 1;
        float x = x01[3 * i0], y = x01[3 * i0 + 1], z = x01[3 * i0 + 2];
#pragma unroll
        for (int it = 0; it < STEPS; ++it) {
            const int64_t i = base + 256 * it;
            if (i - threadIdx.x >= n_live) break;                // (workgroup-uniform)
            uint32_t v[8];
            float f[3];
            // a dense level whose eight corners of THIS wave need no wrap (every wave but those that touch the table's last cells): the two x
            // corners are neighbours in the table -- ONE 8-byte request per (y, z) corner pair, no per-corner index arithmetic or bound test
            // (the launch of the coarse levels is bound by vector issue); same position arithmetic as corners_of
            bool fast = false;
            if (dense) {
                const float px = grid_pos(x, sc), py = grid_pos(y, sc), pz = grid_pos(z, sc);
                const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
                const uint32_t gx = (uint32_t)(int32_t)flx, gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
                const uint32_t base = gx + gy * res + gz * r2;
                fast = __all((int)(base + 1u + res + r2 < size)) != 0;      // (wave-uniform)
                if (fast) {
                    f[0] = px - flx; f[1] = py - fly; f[2] = pz - flz;
                    const uint32_t* t0 = tl + base;
                    const uint2_a4 p0 = *reinterpret_cast<const uint2_a4*>(t0), p1 = *reinterpret_cast<const uint2_a4*>(t0 + res),
                                   p2 = *reinterpret_cast<const uint2_a4*>(t0 + r2), p3 = *reinterpret_cast<const uint2_a4*>(t0 + res + r2);
                    v[0] = p0.x; v[1] = p0.y; v[2] = p1.x; v[3] = p1.y; v[4] = p2.x; v[5] = p2.y; v[6] = p3.x; v[7] = p3.y;
                }
            }
            if (!fast) {
                const Corners c = corners_of(x, y, z, sc, res, size, !dense);
                f[0] = c.f[0]; f[1] = c.f[1]; f[2] = c.f[2];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tl[c.idx[k]];
            }
            if (it + 1 < STEPS) {                                // the next step's coordinates, behind this step's gathers
                int64_t i1 = i + 256;
                if (i1 >= n_live) i1 = n_live - 1;
                x = x01[3 * i1]; y = x01[3 * i1 + 1]; z = x01[3 * i1 + 2];
            }
            float wt[8];
            corner_weights(f, smooth, wt);
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0 = fmaf(wt[k], T16::lo(v[k]), a0);
                a1 = fmaf(wt[k], T16::hi(v[k]), a1);
            }
            if (i < n_live) __builtin_nontemporal_store(T16::pack(a0, a1), &feat[(int64_t)l * n + i]);      // (streamed: read next by the MLP kernel, not here)
        }
    }
}

template <typename T16, int STEPS>
__global__ __launch_bounds__(256) void hashgrid_fwd_big_gather_kernel(GridParams gp, BigLevels lv, const float* __restrict__ x01,
                                                                      const uint32_t* __restrict__ table, uint32_t* __restrict__ feat,
                                                                      int64_t n, const int64_t* __restrict__ n_dev, int64_t grid_stripes, int bpt) {
    big_gather_body<T16, STEPS>(gp, big_item(lv, bpt), x01, table, feat, n, live_count(n, n_dev), grid_stripes, bpt);       // (n stays the level stride)
}

// line-local levels: lane = (sample s of a group of 16, corner pair r = ky + 2 kz); four groups of a wave in flight.  The four lanes
// of a quad fetch the coordinates of ONE sample each (group r's) one step ahead and hand them round by quad broadcasts: three
// registers per lane instead of twelve (70 in all: seven waves per SIMD).
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));   // quad_perm [K,K,K,K]
}
template <int K> struct QuadStep {
    template <typename F> static __device__ __forceinline__ void run(F&& f) { QuadStep<K - 1>::run(f); f(std::integral_constant<int, K - 1>{}); }
};
template <> struct QuadStep<0> { template <typename F> static __device__ __forceinline__ void run(F&&) {} };

template <typename T16, int STEPS, bool OVL>
__device__ __forceinline__ void big_local_body(const GridParams& gp, const GridLocal& gl, const BigItem& w, const float* __restrict__ x01,
                                               const uint32_t* __restrict__ table, uint32_t* __restrict__ feat,
                                               int64_t n, int64_t n_live, int64_t grid_stripes, int bpt) {
    const int l = w.l;
    const bool smooth = gp.interpolation == PERF_INTERP_SMOOTHSTEP;
    const uint32_t* tl = table + gp.offset[l];
    constexpr int64_t kGroup = 256 * STEPS;
    const int64_t stripe = 8 * (int64_t)bpt * kGroup;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t s = lane >> 2, r = lane & 3u;
    const float sc = gp.scale[l];
    const uint32_t size = gp.size[l];
    const bool hashed = gp.hashed[l] != 0;
    // local_vertex_index() with a lane's (y, z) part formed once per group and ONE pair of 32-bit multiplies whichever way the super-block is
    // addressed: multipliers, masks and the combining rule are uniform over the level
    const uint32_t sb_sh = gl.shx + gl.shy + gl.shz;
    const uint32_t mul_y = hashed ? kPrimeY : gl.nsx[l], mul_z = hashed ? kPrimeZ : gl.nsxy[l], slot_mask = (size >> sb_sh) - 1u;
    const uint32_t mkx = (1u << (gl.shx - 2)) - 1u, mky = (1u << (gl.shy - 2)) - 1u, mkz = (1u << (gl.shz - 1)) - 1u;
    for (int64_t Q = w.q0; Q * stripe < n_live; Q += grid_stripes) {
        const int64_t base = (((Q << 3) + ((w.xcd - l) & 7)) * bpt + w.sub) * kGroup + (int64_t)wave * (64 * STEPS);
        if (base >= n_live) continue;                                // (wave-uniform)
        // (wave-uniform bases + 32-bit lane offsets: addresses cost one register, not two)
        const float* __restrict__ xb = x01 + 3 * base;
        uint32_t* __restrict__ fb = feat + (int64_t)l * n + base;
        const int64_t left64 = n_live - base;
        const uint32_t left = left64 < 64 * STEPS ? (uint32_t)left64 : 64u * STEPS;      // live samples of this wave's share
        const uint32_t mo = 16u * r + s;                             // this lane's sample of a step: coordinates fetched, feature stored
        float nx, ny, nz;                                            // this lane's share of the next step's coordinates
        {
            const uint32_t o = 3u * (mo < left ? mo : left - 1u);    // (idle lanes of the last step repeat the last sample)
            nx = xb[o]; ny = xb[o + 1u]; nz = xb[o + 2u];
        }
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            if (64u * step >= left) break;                            // (wave-uniform)
            // the requests of the four groups: plain line-local levels, the ALIGNED 16-byte run that holds the pair's first vertex (+ a 4-byte
            // gather of the second one when the cell starts at the run's last vertex: a quarter of the samples); overlapping runs, exactly
            // the pair -- 8 bytes, 4-byte aligned, never across a run (+ the gather for the last cell of a super-block row: 1 in 24):
            // 56 registers instead of 70 (eight waves per SIMD instead of seven), half the bytes returned per request
            uint4 q[OVL ? 1 : 4];
            uint2_a4 q2[OVL ? 4 : 1];
            uint32_t e[4] = {0u, 0u, 0u, 0u}, lxp = 0u;               // (lxp, two bits per group: position of the first vertex in its run; OVL: 1 = second corner in e)
            float w0s[4], w1s[4];
            QuadStep<4>::run([&](auto itc) {
                constexpr int it = decltype(itc)::value;
                const float x = quad_bcast<it>(nx), y = quad_bcast<it>(ny), z = quad_bcast<it>(nz);
                const float px = grid_pos(x, sc), py = grid_pos(y, sc), pz = grid_pos(z, sc);
                const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
                float fx = px - flx, fy = py - fly, fz = pz - flz;
                const uint32_t gx = (uint32_t)(int32_t)flx, gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
                const uint32_t vy = gy + (r & 1u), vz = gz + (r >> 1);
                // (a level holds at most 2^30 entries: 32-bit BYTE offsets from the wave-uniform level base, one address register per request)
                const char* tb = reinterpret_cast<const char*>(tl);
                const uint32_t ya = (vy >> gl.shy) * mul_y, zb = (vz >> gl.shz) * mul_z;
                const uint32_t yz_in = (((vy >> 2) & mky) << (gl.shx - 2)) + (((vz >> 1) & mkz) << (gl.shx - 2 + gl.shy - 2));
                const uint32_t in_blk = ((vy & 3u) << 2) + ((vz & 1u) << 4);
                auto entry = [&](uint32_t X, uint32_t width) {        // entry of storage vertex (X, vy, vz): local_vertex_index(gl, l, size, hashed, X, vy, vz)
                    const uint32_t sx = X >> gl.shx;
                    const uint32_t slot = hashed ? ((sx ^ ya ^ zb) & slot_mask) : (sx + ya + zb);
                    const uint32_t i = (slot << sb_sh) + ((((X >> 2) & mkx) + yz_in) << 5) + (X & 3u) + in_blk;
                    return i < size - width ? i : size - width;       // (a point outside the unit cube must not read outside a densely addressed level; inside, never taken)
                };
                if constexpr (OVL) {
                    const uint32_t X = overlap_x(gx), sbm = (1u << gl.shx) - 1u;
                    const bool edge = (X & sbm) == sbm - 1u;         // the last cell of a super-block row: second corner = the next super-block's first vertex
                    lxp |= (edge ? 1u : 0u) << (2 * it);
                    q2[it] = *reinterpret_cast<const uint2_a4*>(tb + (entry(X, 2u) << 2));
                    if (edge) e[it] = *reinterpret_cast<const uint32_t*>(tb + (entry(X + 2u, 1u) << 2));
                } else {
                    lxp |= (gx & 3u) << (2 * it);
                    q[it] = *reinterpret_cast<const uint4*>(tb + (entry(gx & ~3u, 4u) << 2));   // (16-byte aligned by construction)
                    if ((gx & 3u) == 3u) e[it] = *reinterpret_cast<const uint32_t*>(tb + (entry(gx + 1u, 1u) << 2));
                }
                if (smooth) {
                    fx = fx * fx * (3.0f - 2.0f * fx); fy = fy * fy * (3.0f - 2.0f * fy); fz = fz * fz * (3.0f - 2.0f * fz);
                }
                const float wy = (r & 1u) ? fy : 1.0f - fy, wz = (r >> 1) ? fz : 1.0f - fz;
                w0s[it] = ((1.0f - fx) * wy) * wz; w1s[it] = (fx * wy) * wz;      // this lane's two corners with corner_weights' products ((wx * wy) * wz)
            });
            if (step + 1 < STEPS) {                                  // the next step's coordinates, behind this step's table lines
                const uint32_t i = 64u * (step + 1) + mo;
                const uint32_t o = 3u * (i < left ? i : left - 1u);
                nx = xb[o]; ny = xb[o + 1u]; nz = xb[o + 2u];
            }
            uint32_t mine = 0u;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint32_t lx = (lxp >> (2 * it)) & 3u;
                uint32_t a0, a1;
                if constexpr (OVL) {
                    a0 = q2[it].x;
                    a1 = lx ? e[it] : q2[it].y;
                } else {
                    a0 = lx == 0u ? q[it].x : (lx == 1u ? q[it].y : (lx == 2u ? q[it].z : q[it].w));
                    a1 = lx == 0u ? q[it].y : (lx == 1u ? q[it].z : (lx == 2u ? q[it].w : e[it]));
                }
                const float p0 = quad_sum(fmaf(w1s[it], T16::lo(a1), w0s[it] * T16::lo(a0)));     // ... then the sample's four lanes
                const float p1 = quad_sum(fmaf(w1s[it], T16::hi(a1), w0s[it] * T16::hi(a0)));
                if ((int)r == it) mine = T16::pack(p0, p1);          // (all four lanes hold the same sums: lane r keeps group r's)
            }
            const uint32_t io = 64u * step + mo;
            if (io < left) __builtin_nontemporal_store(mine, &fb[io]);
        }
    }
}

template <typename T16, int STEPS, bool OVL>
__global__ __launch_bounds__(256) void hashgrid_fwd_big_local_kernel(GridParams gp, GridLocal gl, BigLevels lv, const float* __restrict__ x01,
                                                                     const uint32_t* __restrict__ table, uint32_t* __restrict__ feat,
                                                                     int64_t n, const int64_t* __restrict__ n_dev, int64_t grid_stripes, int bpt) {
    big_local_body<T16, STEPS, OVL>(gp, gl, big_item(lv, bpt), x01, table, feat, n, live_count(n, n_dev), grid_stripes, bpt);
}

// both kinds of level in ONE launch (launch_big): a workgroup is one level of either kind
template <typename T16, int STEPS, bool OVL>
__global__ __launch_bounds__(256) void hashgrid_fwd_big_mixed_kernel(GridParams gp, GridLocal gl, BigLevels lv, const float* __restrict__ x01,
                                                                     const uint32_t* __restrict__ table, uint32_t* __restrict__ feat,
                                                                     int64_t n, const int64_t* __restrict__ n_dev, int64_t grid_stripes, int bpt) {
    const BigItem w = big_item(lv, bpt);
    const int64_t n_live = live_count(n, n_dev);
    if (gl.local[w.l]) big_local_body<T16, STEPS, OVL>(gp, gl, w, x01, table, feat, n, n_live, grid_stripes, bpt);
    else big_gather_body<T16, STEPS>(gp, w, x01, table, feat, n, n_live, grid_stripes, bpt);
}

// turns of kBigTurn consecutive samples: 4 steps x 256 samples x 4 workgroups (measured at T = 2^28, line-local: 1 / 2 / 4 / 8
// workgroups per turn 0.760 / 0.750 / 0.738 / 0.742 ms; 1 / 2 / 4 / 8 steps per wave 0.838 / 0.767 / 0.734 / 0.783 ms; overlapping runs,
// one launch: (steps, workgroups per turn) (4, 4) 0.597 ms, (4, 8) 0.606, (4, 2) 0.628, (4, 1) 0.675, (2, 4) 0.670, (2, 8) 0.661, (8, 2) 0.658, (8, 4) 0.637)
constexpr int kBigSteps = 4;
constexpr int kBigTurnGroups = 4;

template <typename T16>
static void launch_big(const GridParams& gp, const GridLocal& gl, const float* x01, const void* table16, void* feat16, int64_t n,
                       const int64_t* n_dev, void* stream) {
    const int64_t stripe = 8 * (int64_t)kBigTurnGroups * 256 * kBigSteps;
    int64_t stripes = div_up(n, stripe);
    if (stripes > kBigMaxStripes) stripes = kBigMaxStripes;
    BigLevels gather, local;
    gather.count = local.count = 0;
    for (int l = 0; l < PERF_MAX_LEVELS; ++l) gather.level[l] = local.level[l] = 0;
    for (int l = 0; l < gp.n_levels; ++l) {
        if (gl.local[l]) local.level[local.count++] = l; else gather.level[gather.count++] = l;
    }
    // Order of the work items (measured at T = 2^28, profiles/r06_config5_long_waves.json): beside line-local levels the levels
    // ALTERNATE from workgroup to workgroup (0.72 ms; a level at a time 0.79: cache-resident and HBM-bound levels share the chip at any
    // moment); tables beyond the caches in tcnn's layout are served a LEVEL AT A TIME (1.26 ms against 1.47 alternating).
    gather.order = local.order = local.count ? 0 : 1;
    gather.stripes = local.stripes = stripes;
    // (line-local levels first: the long launch; the few coarse levels' gathers find the coordinates in the caches)
    if (local.count && gather.count) {
        // both kinds of level in ONE launch, the coarse levels behind the line-local ones in the list the workgroups cycle through: the coarse
        // levels' workgroups are bound by vector arithmetic, the line-local ones by the L1's request rate -- side by side on a CU they hide
        // a third of the coarse levels' 75 us (two launches one after the other: 0.611 ms per 4.2 M samples at T = 2^28, one: 0.588; the
        // position of the coarse levels in the list does not matter: first 0.589, last 0.588, spread 0.593)
        BigLevels all = local;
        for (int k = 0; k < gather.count; ++k) all.level[all.count++] = gather.level[k];
        const dim3 g((unsigned)(stripes * all.count * 8 * kBigTurnGroups));
        if (gl.ovl)
            hipLaunchKernelGGL((hashgrid_fwd_big_mixed_kernel<T16, kBigSteps, true>), g, dim3(256), 0, as_stream(stream),
                               gp, gl, all, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev, stripes, kBigTurnGroups);
        else
            hipLaunchKernelGGL((hashgrid_fwd_big_mixed_kernel<T16, kBigSteps, false>), g, dim3(256), 0, as_stream(stream),
                               gp, gl, all, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev, stripes, kBigTurnGroups);
        return;
    }
    if (local.count) {
        const dim3 g((unsigned)(stripes * local.count * 8 * kBigTurnGroups));
        if (gl.ovl)
            hipLaunchKernelGGL((hashgrid_fwd_big_local_kernel<T16, kBigSteps, true>), g, dim3(256), 0, as_stream(stream),
                               gp, gl, local, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev, stripes, kBigTurnGroups);
        else
            hipLaunchKernelGGL((hashgrid_fwd_big_local_kernel<T16, kBigSteps, false>), g, dim3(256), 0, as_stream(stream),
                               gp, gl, local, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev, stripes, kBigTurnGroups);
    }
    if (!gather.count) return;
    // tables beyond the caches in tcnn's layout (no line-local level) are served best by waves of ONE step -- measured at T = 2^28 with
    // every level in tcnn's layout: 1.50 ms at one step, 1.58 at four, 1.87 at eight -- over the same turns; the coarse, cache-resident
    // levels beside line-local ones take the four steps
    if (local.count)
        hipLaunchKernelGGL((hashgrid_fwd_big_gather_kernel<T16, kBigSteps>), dim3((unsigned)(stripes * gather.count * 8 * kBigTurnGroups)), dim3(256), 0, as_stream(stream),
                           gp, gather, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev, stripes, kBigTurnGroups);
    else
        hipLaunchKernelGGL((hashgrid_fwd_big_gather_kernel<T16, 1>), dim3((unsigned)(stripes * gather.count * 8 * kBigTurnGroups * kBigSteps)), dim3(256), 0, as_stream(stream),
                           gp, gather, x01, (const uint32_t*)table16, (uint32_t*)feat16, n, n_dev, stripes, kBigTurnGroups * kBigSteps);
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
__global__ __launch_bounds__(256) void hashgrid_corners_kernel(GridParams gp, GridLocal gl, const float* __restrict__ x01,
                                                               int32_t* __restrict__ idx_out, int64_t n) {
    const int l = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || l >= gp.n_levels) return;
    const Corners c = corners_of_any(gp, gl, l, x01[3 * i], x01[3 * i + 1], x01[3 * i + 2]);
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

static inline unsigned grouped_grid(int64_t n) { return (unsigned)(div_up(n, 256) * 8); }

}  // namespace perf

using namespace perf;

extern "C" int perf_hashgrid_fwd(const perf_grid_desc* grid, const float* x01, const void* table16,
                                 void* feat16, int64_t n, const int64_t* n_dev, int dtype, void* stream) {
    GridParams gp;
    GridLocal gl;
    int rc = fill_params(grid, &gp, &gl);
    if (rc) return rc;
    PERF_REQUIRE(n >= 0 && n < (int64_t(1) << 31) * 16, "n out of range");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && table16 && feat16, "NULL pointer");
    if (gl.any || gp.n_levels > 16) {
        // deep grids and line-local tables: one level per workgroup, XCD-stable balanced (hashgrid_fwd_big_kernel)
        PERF_REQUIRE(!gl.any || ((uintptr_t)table16 & 15u) == 0, "perf_hashgrid_fwd: a line-local table must be 16-byte aligned");
        if (dtype != PERF_DTYPE_BF16 && dtype != PERF_DTYPE_FP16) { set_error("perf_hashgrid_fwd: bad dtype %d", dtype); return PERF_E_INVALID; }
        if (dtype == PERF_DTYPE_BF16) launch_big<BF16>(gp, gl, x01, table16, feat16, n, n_dev, stream);
        else launch_big<FP16>(gp, gl, x01, table16, feat16, n, n_dev, stream);
        PERF_LAUNCH_CHECK("perf_hashgrid_fwd");
        return PERF_OK;
    }
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
    GridLocal gl;
    int rc = fill_params(grid, &gp, &gl);
    if (rc) return rc;
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x01 && idx, "NULL pointer");
    PERF_REQUIRE(gp.offset[gp.n_levels - 1] + gp.size[gp.n_levels - 1] < ((uint64_t)1 << 31), "perf_hashgrid_corners: table too large for int32 entries");
    hipLaunchKernelGGL(hashgrid_corners_kernel, dim3((unsigned)div_up(n, 256), gp.n_levels), dim3(256), 0, as_stream(stream), gp, gl,
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
