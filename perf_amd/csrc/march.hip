// Occupancy-grid ray marching (nerfacc traverse_grids semantics as restated in SURVEY.md A.3 and
// oracle/perf_oracle.py:occ_march) for gfx950.
//
// One 64-lane wavefront owns one ray.  Lane k of chunk q tests lattice interval 64*q+k: the 64
// midpoints of a chunk are consecutive points on the ray, so their occupancy words are neighbours
// in the bit field (256^3 bits = 2 MiB: L2 resident).  A wave ballot turns the 64 tests into one
// uint64 keep-mask; pass 1 stores the masks and the per-ray popcount, an exclusive scan over rays
// gives the packed offsets, pass 2 expands the masks with a per-lane prefix popcount -- the
// "count -> scan -> write" protocol with bit-exact, t-sorted output and no atomics.
// All lattice/cell arithmetic is unfused fp32 (mul_rn/add_rn) so it matches numpy bit for bit.
#include <cstring>

#include "common.hpp"

namespace perf {

struct MarchParams {
    float lo[3], inv_ext[3];
    float hi[3];
    float far_plane, step;
    int32_t res, max_steps, mask_words;
    int32_t lattice_mode;    // PERF_LATTICE_*
    int32_t use_coarse;      // 1: skip 64-interval chunks whose midpoint lies in an empty DILATED block (kCoarseBlock^3 cells)
    float chunk_cells[3];    // fine cells a 64-interval chunk spans per unit of |d| along each axis (64 step res / extent)
};

// Coarse skip grid: bit b of `coarse` is set iff any fine cell in the 3x3x3 neighbourhood of B^3-block b is occupied
// (B = kCoarseBlock).  A chunk of 64 lattice intervals spans at most `span` fine cells; when span/2 + 1.5 <= B every
// midpoint of the chunk lies within one block of the block that holds the chunk's centre, so an empty dilated block proves
// the chunk empty (conservative: results are identical to the exhaustive test).  B = 4: at PeRF's step (5e-4, 4.1 cells per
// chunk) the condition holds with the smallest power of two, and a thin occupied shell keeps 3 blocks = 12 cells = ~3
// chunks of a ray alive instead of the ~6 that 8^3 blocks kept -- phase B of march_count is what a frame pays for.
constexpr int kCoarseShift = 2;
constexpr int kCoarseBlock = 1 << kCoarseShift;
__global__ __launch_bounds__(256) void coarse_build_kernel(const uint32_t* __restrict__ bits, int res,
                                                           uint32_t* __restrict__ coarse) {
    constexpr int B = kCoarseBlock;
    const int cr = res / B;
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= cr * cr * cr) return;
    const int bx = b / (cr * cr), by = (b / cr) % cr, bz = b % cr;
    bool any = false;
    for (int x = max(bx - 1, 0) * B; x < min(bx + 2, cr) * B && !any; ++x)
        for (int y = max(by - 1, 0) * B; y < min(by + 2, cr) * B && !any; ++y) {
            // the z run [z0, z1) of a row is contiguous in the bit field (B divides 32: a block's run never straddles a word)
            const int z0 = max(bz - 1, 0) * B, z1 = min(bz + 2, cr) * B;
            for (int z = z0; z < z1; z += B) {
                const uint32_t ci = (uint32_t)((x * res + y) * res + z);
                if ((bits[ci >> 5] >> (ci & 31)) & ((1u << B) - 1u)) { any = true; break; }
            }
        }
    if (any) atomicOr(&coarse[b >> 5], 1u << (b & 31));
}

// Fine skip grid, behind the coarse one in the same buffer (perf_occ_coarse_words counts both): bit b is set iff any fine cell in
// the 3x3x3 neighbourhood of the 2^3-block b is occupied.  A chunk that the coarse test lets through is tested again at TWO of
// its lattice points (k0 + 16, k0 + 48): every interval midpoint of the chunk lies within 16.5 intervals of one of them, i.e.
// within 0.258 span cells per axis -- while that (+ half a cell of slack) stays below the block size 2, the midpoint's block is the
// test point's block or a neighbour of it, so two empty dilated blocks prove the chunk empty (conservative: masks and counts are
// what the exhaustive test gives; the bit-exact marching tests run through it).  Around a thin occupied shell the coarse grid keeps
// 12 + 4.1 cells = ~4-5 chunks of a ray alive, the fine one 6 + 4.1 = ~2.5.
constexpr int kFineShift = 1;
constexpr int kFineBlock = 1 << kFineShift;
__host__ __device__ __forceinline__ int64_t coarse_words_of(int res) { const int64_t cr = res >> kCoarseShift; return (cr * cr * cr + 31) / 32; }
__host__ __device__ __forceinline__ int64_t fine_words_of(int res) { const int64_t fr = res >> kFineShift; return (fr * fr * fr + 31) / 32; }
__global__ __launch_bounds__(256) void fine_build_kernel(const uint32_t* __restrict__ bits, int res, uint32_t* __restrict__ fine) {
    constexpr int B = kFineBlock;
    const int fr = res / B;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= (int64_t)fr * fr * fr) return;
    const int bx = (int)(b / ((int64_t)fr * fr)), by = (int)((b / fr) % fr), bz = (int)(b % fr);
    bool any = false;
    const int z0 = max(bz - 1, 0) * B, z1 = min(bz + 2, fr) * B;
    for (int x = max(bx - 1, 0) * B; x < min(bx + 2, fr) * B && !any; ++x)
        for (int y = max(by - 1, 0) * B; y < min(by + 2, fr) * B && !any; ++y)
            for (int z = z0; z < z1; z += B) {                       // (B divides 32: a block's z run never straddles a word)
                const uint32_t ci = (uint32_t)((x * res + y) * res + z);
                if ((bits[ci >> 5] >> (ci & 31)) & ((1u << B) - 1u)) { any = true; break; }
            }
    if (any) atomicOr(&fine[b >> 5], 1u << (b & 31));
}

// the fine block of a point on the ray (same position arithmetic as the cell look-ups)
__device__ __forceinline__ uint32_t fine_block_at(const MarchParams& mp, const float o[3], const float d[3], float t, float rf, int fr) {
    int cb[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float p = add_rn(o[a], mul_rn(d[a], t));
        const float u = mul_rn(mul_rn(sub_rn(p, mp.lo[a]), mp.inv_ext[a]), rf);
        cb[a] = ((int)fminf(fmaxf(floorf(u), 0.0f), rf - 1.0f)) >> kFineShift;
    }
    return (uint32_t)((cb[0] * fr + cb[1]) * fr + cb[2]);
}

__device__ __forceinline__ float lattice_single(float t0, int k, float step) { return add_rn(t0, mul_rn((float)k, step)); }

// PERF_LATTICE_REPEATED: t_0 = t0, t_{j+1} = fl(t_j + step) -- the lattice a marcher that ADVANCES by `t += dt` produces
// (nerfacc's traverse_grids, as far as it can be read without the package: oracle/perf_oracle.py header).  Evaluated for an
// arbitrary k without k additions: inside one binade consecutive floats have consecutive bit patterns, and from the second
// in-binade addition on every addition moves by the same number of ulps (round-to-nearest-even of step / ulp: a tie lands
// on an even mantissa once and then stays even) -- so the walk jumps binade by binade with two real additions each.
// floor(n / d) for n < 2^24, d >= 1: the reciprocal estimate is off by at most one (the compiler's generic 32-bit division is
// ~4x the instructions, and the walk below divides once per binade)
__device__ __forceinline__ uint32_t div_u24(uint32_t n, uint32_t d) {
    uint32_t q = (uint32_t)((float)n * __frcp_rn((float)d));
    const uint32_t p = q * d;
    if (p > n) --q;                                              // (estimate one too large)
    else if (n - p >= d) ++q;                                    // (one too small)
    return q;
}

__device__ __forceinline__ float lattice(float t0, int k, float step, int mode);

__device__ __forceinline__ float lattice_repeated(float t0, int k, float step) {
    float t = t0;
    int left = k;
    while (left > 0) {
        const float t1 = add_rn(t, step);
        if (--left == 0) return t1;
        if ((__float_as_uint(t1) >> 23) != (__float_as_uint(t) >> 23)) { t = t1; continue; }   // entered a binade: one more real step first
        const float t2 = add_rn(t1, step);
        if (--left == 0) return t2;
        const uint32_t b1 = __float_as_uint(t1), b2 = __float_as_uint(t2);
        if ((b2 >> 23) != (b1 >> 23)) { t = t2; continue; }
        const uint32_t d = b2 - b1;                              // ulps per step from here to the end of the binade
        if (d == 0u) return t2;                                  // (step below half an ulp: the lattice is stuck)
        const uint32_t top = (b2 & 0xff800000u) + 0x00800000u;   // first bit pattern of the next binade
        uint32_t j = div_u24(top - 1u - b2, d);                  // further additions that stay inside
        if (j > (uint32_t)left) j = (uint32_t)left;
        t = __uint_as_float(b2 + j * d);
        left -= (int)j;
    }
    return t;
}

// The same walk, ONCE per ray, as a table of runs: run i covers lattice indices [ks[i], ks[i+1]) and holds
// t_k = bits(bs[i] + (k - ks[i]) * dd[i]) -- inside a binade consecutive lattice points are equidistant bit patterns.
// march_count_kernel evaluates the lattice at 3 indices per chunk in phase A and once per lane and live chunk in phase B,
// i.e. thousands of per-lane walks per ray at PeRF's 3,000 steps (measured: the kernel went from 13 to 85 us per 8,192
// rays when the repeated lattice became the default); with the table a ray is walked once -- on values that are uniform
// over the wave (the float additions run on the vector unit, everything else on the scalar unit) -- and an evaluation is
// a binary search over <= kMaxRuns run starts in LDS.  Returns the number of runs, or -1 when the table does not reach
// index k_need (more binades than the table holds: the caller falls back to the per-lane walk).
constexpr int kMaxRuns = 64;

// (the walk runs on values that are uniform over the wave: the float additions go through the vector unit and come back with
//  readfirstlane, everything else stays on the scalar unit; lane 0 stores.  Letting lanes 0..3 of one wave walk the block's
//  four rays and synchronising the block measured SLOWER: 26.9 vs 20.6 us per 8,192 rays -- the other waves wait out the
//  walk's latency)
// UNIFORM: one ray per WAVE (values uniform over the wave, kept on the scalar unit through readfirstlane; lane 0 stores);
// otherwise one ray per LANE (lattice_runs_kernel) -- the same arithmetic either way, so the tables are identical.
template <bool UNIFORM>
__device__ __forceinline__ int lattice_runs_build(float t0, float step, int k_need, int32_t* __restrict__ ks, uint32_t* __restrict__ bs,
                                                  uint32_t* __restrict__ dd, bool writer) {
    int n = 0;
    auto emit = [&](int k, uint32_t b, uint32_t d) {
        if (writer) { ks[n] = k; bs[n] = b; dd[n] = d; }
        ++n;
    };
    auto uni = [](uint32_t v) { return UNIFORM ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v) : v; };
    uint32_t tb = uni(__float_as_uint(t0));
    emit(0, tb, 0u);
    int K = 0;
    while (K < k_need) {
        if (n + 2 > kMaxRuns) return -1;
        const uint32_t b1 = uni(__float_as_uint(add_rn(__uint_as_float(tb), step)));
        if ((b1 >> 23) != (tb >> 23)) { emit(K + 1, b1, 0u); tb = b1; K += 1; continue; }    // entered a binade: one more real step first
        const uint32_t b2 = uni(__float_as_uint(add_rn(__uint_as_float(b1), step)));
        if ((b2 >> 23) != (b1 >> 23)) { emit(K + 1, b1, 0u); emit(K + 2, b2, 0u); tb = b2; K += 2; continue; }
        const uint32_t d = b2 - b1;
        emit(K + 1, b1, d);                                      // t_{K+1}, t_{K+2}, ... equidistant to the end of the binade
        if (d == 0u) return n;                                   // (step below half an ulp: the lattice is stuck at t_{K+1} for good)
        const uint32_t top = (b2 & 0xff800000u) + 0x00800000u;
        const uint32_t j = uni(div_u24(top - 1u - b2, d));
        tb = b2 + j * d;
        K += 2 + (int)j;
    }
    return n;
}

// Per-ray tables in device memory (perf_occ_lattice_runs): row r = [n | ks[kMaxRuns] | bs[kMaxRuns] | dd[kMaxRuns]] of
// kRunsStride 32-bit words (n = -1: the table did not fit, walk per lane).  The uniform build above costs a whole WAVE per ray --
// ~840 dependent vector instructions on uniform values, x 8,192 rays x 4 cycles each: 13 of the 27 us march_count took on a
// jittered training batch -- the kernel below gives every ray a LANE instead (128 waves for 8,192 rays).
constexpr int kRunsStride = 3 * kMaxRuns + 4;

// The table of a launch whose rays all start their lattice at the same t0 (eval renders: no stratified jitter): built ONCE
// on the host with the same IEEE single-precision additions and handed to the kernels as an argument.
struct SharedRuns {
    int32_t n;                    // 0: none (per-ray origins, or the single-rounding lattice); -1 cannot happen (the host falls back to 0)
    int32_t ks[kMaxRuns];
    uint32_t bs[kMaxRuns], dd[kMaxRuns];
};

static inline uint32_t host_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float host_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static void shared_runs_build(float t0, float step, int k_need, SharedRuns* sr) {
    sr->n = 0;
    int n = 0;
    auto emit = [&](int k, uint32_t b, uint32_t d) { sr->ks[n] = k; sr->bs[n] = b; sr->dd[n] = d; ++n; };
    uint32_t tb = host_bits(t0);
    emit(0, tb, 0u);
    int K = 0;
    while (K < k_need) {
        if (n + 2 > kMaxRuns) return;                            // (too many binades for the table: the kernels walk per lane)
        volatile float s1 = host_float(tb) + step;               // (volatile: one rounding to fp32 per addition, whatever the host compiler does)
        const uint32_t b1 = host_bits(s1);
        if ((b1 >> 23) != (tb >> 23)) { emit(K + 1, b1, 0u); tb = b1; K += 1; continue; }
        volatile float s2 = host_float(b1) + step;
        const uint32_t b2 = host_bits(s2);
        if ((b2 >> 23) != (b1 >> 23)) { emit(K + 1, b1, 0u); emit(K + 2, b2, 0u); tb = b2; K += 2; continue; }
        const uint32_t d = b2 - b1;
        emit(K + 1, b1, d);
        if (d == 0u) break;
        const uint32_t top = (b2 & 0xff800000u) + 0x00800000u;
        const uint32_t j = (top - 1u - b2) / d;
        tb = b2 + j * d;
        K += 2 + (int)j;
    }
    sr->n = n;
}

__device__ __forceinline__ float lattice(float t0, int k, float step, int mode) {
    return mode == PERF_LATTICE_REPEATED ? lattice_repeated(t0, k, step) : lattice_single(t0, k, step);
}

struct LatticeRuns {
    int n;                       // > 0: table; <= 0: evaluate directly (single lattice, or a table that did not fit)
    const int32_t* ks; const uint32_t* bs; const uint32_t* dd;
    float t0, step; int mode;
    const float* full;           // != NULL: t_k for every k of the launch (all rays on one lattice, perf_occ_lattice_table): one load
    __device__ __forceinline__ int find(int k) const {           // last run that starts at or before k
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (ks[mid] <= k) lo = mid; else hi = mid - 1;
        }
        return lo;
    }
    __device__ __forceinline__ float at(int r, int k) const { return __uint_as_float(bs[r] + (uint32_t)(k - ks[r]) * dd[r]); }
    __device__ __forceinline__ float operator()(int k) const {
        if (full) return full[k];
        if (n <= 0) return lattice(t0, k, step, mode);
        return at(find(k), k);
    }
    // a little further along the lattice than run r reaches: the next runs are tried before a search
    __device__ __forceinline__ float after(int& r, int k) const {
        if (full) return full[k];
        if (n <= 0) return lattice(t0, k, step, mode);
        while (r + 1 < n && ks[r + 1] <= k) ++r;
        return at(r, k);
    }
};

// t_k from a ray's row of the per-ray tables (device memory, L2 resident): binary search over the run starts
__device__ __forceinline__ float runs_row_at(const int32_t* __restrict__ row, int k, float t0, float step) {
    const int n = row[0];
    if (n <= 0) return lattice_repeated(t0, k, step);
    const int32_t* ks = row + 1;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ks[mid] <= k) lo = mid; else hi = mid - 1;
    }
    return __uint_as_float((uint32_t)row[1 + kMaxRuns + lo] + (uint32_t)(k - ks[lo]) * (uint32_t)row[1 + 2 * kMaxRuns + lo]);
}

// Lattice origin of ray r.  t0s == NULL: t0_base (the near plane).  t0_scale == 0: t0s[r] as given.  Otherwise t0s holds the
// stratified draw u in [0,1) and the origin is fl(u * t0_scale) (+ t0_base when that is not 0) -- the two torch ops of
// OccGridEstimator.sampling (near_plane + u * render_step_size), formed here so that no separate launch is needed.
__device__ __forceinline__ float lattice_origin(const float* __restrict__ t0s, int64_t r, float t0_scale, float t0_base) {
    if (!t0s) return t0_base;
    float t = t0s[r];
    if (t0_scale != 0.f) {
        t = mul_rn(t, t0_scale);
        if (t0_base != 0.f) t = add_rn(t, t0_base);
    }
    return t;
}

// Per-ray record in the mask buffer: [live words | chunk masks]: bit q of the live words says chunk q (lattice
// intervals 64q..64q+63) may hold samples; only live chunks have their mask word written (and later read).
__device__ __forceinline__ int n_live_words(int mask_words) { return (mask_words + 63) >> 6; }

// HEAD: the first `K` samples of every ray (the head of the two-phase sampler) are written right here, to rows r*K .. r*K+K-1
// of arrays of n_rays*K rows (rows beyond a ray's count are padded with sel = 0) -- what used to take a count clamp, a scan
// over the rays and a pass of march_write_kernel.  Same values as that pass writes (lattice + sample_point_store).
struct HeadOut {
    int32_t K;
    int64_t* ri; float* ts; float* te; int32_t* packed;
    float* x01; uint8_t* sel;
    Aabb bb;
};

template <bool HEAD>
__global__ __launch_bounds__(256) void march_count_kernel(MarchParams mp, const float* __restrict__ ro,
                                                          const float* __restrict__ rd, const float* __restrict__ t0s,
                                                          int64_t n_rays, const uint32_t* __restrict__ bits,
                                                          const uint32_t* __restrict__ coarse,
                                                          uint64_t* __restrict__ masks, int32_t* __restrict__ counts, HeadOut ho,
                                                          float t0_scale, float t0_base, SharedRuns sr, const float* __restrict__ lat_full) {
    const int lane = threadIdx.x & 63;
    // (the ray index is wave uniform: saying so turns the loads of the ray's origin, direction and lattice origin into
    //  scalar loads -- seven vector-memory instructions per ray less; the kernel is bound by VMEM issue, not by bytes)
    const int64_t r = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // repeated-addition lattice: the ray's walk as a table of runs in LDS -- ONE table for the block when all rays start at the
    // same t0 (built on the host), else every wave walks its own ray
    __shared__ int32_t s_ks[4][kMaxRuns];
    __shared__ uint32_t s_bs[4][kMaxRuns], s_dd[4][kMaxRuns];
    __shared__ int32_t s_head[HEAD ? 4 : 1][64];       // HEAD: lattice indices of the ray's first K samples (written after the chunk loop)
    if (mp.lattice_mode == PERF_LATTICE_REPEATED && sr.n > 0) {
        if ((int)threadIdx.x < sr.n) { s_ks[0][threadIdx.x] = sr.ks[threadIdx.x]; s_bs[0][threadIdx.x] = sr.bs[threadIdx.x]; s_dd[0][threadIdx.x] = sr.dd[threadIdx.x]; }
        __syncthreads();
    }
    if (r >= n_rays) return;
    const float o[3] = {ro[3 * r], ro[3 * r + 1], ro[3 * r + 2]};
    const float d[3] = {rd[3 * r], rd[3 * r + 1], rd[3 * r + 2]};
    const float t0 = lattice_origin(t0s, r, t0_scale, t0_base);
    // slab test (fminf/fmaxf drop NaNs like np.fmin/np.fmax)
    float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = __fdiv_rn(1.0f, d[a]);
        const float t1 = mul_rn(sub_rn(mp.lo[a], o[a]), inv), t2 = mul_rn(sub_rn(mp.hi[a], o[a]), inv);
        const float l = fminf(t1, t2), h = fmaxf(t1, t2);
        tmin = (a == 0) ? l : fmaxf(tmin, l);
        tmax = (a == 0) ? h : fminf(tmax, h);
    }
    const float lo = fmaxf(tmin, t0), hi = fminf(tmax, mp.far_plane);
    const int wv = threadIdx.x >> 6;
    const int tab = sr.n > 0 ? 0 : wv;
    LatticeRuns lat;
    lat.n = 0;
    lat.ks = s_ks[tab]; lat.bs = s_bs[tab]; lat.dd = s_dd[tab]; lat.t0 = t0; lat.step = mp.step; lat.mode = mp.lattice_mode;
    // per-ray origins with a table argument: the rows of perf_occ_lattice_runs (not a shared t_k array)
    const int32_t* ray_row = (t0s && lat_full) ? reinterpret_cast<const int32_t*>(lat_full) + r * (int64_t)kRunsStride : nullptr;
    lat.full = ray_row ? nullptr : lat_full;
    if (ray_row) {
        if (mp.lattice_mode == PERF_LATTICE_REPEATED) {
            s_ks[wv][lane] = ray_row[1 + lane]; s_bs[wv][lane] = (uint32_t)ray_row[1 + kMaxRuns + lane]; s_dd[wv][lane] = (uint32_t)ray_row[1 + 2 * kMaxRuns + lane];
            lat.n = ray_row[0];
            __builtin_amdgcn_wave_barrier();
        }
    } else if (mp.lattice_mode == PERF_LATTICE_REPEATED && !lat_full) {
        if (sr.n > 0) lat.n = sr.n;
        else {
            lat.n = lattice_runs_build<true>(t0, mp.step, mp.mask_words * 64 + 64, s_ks[wv], s_bs[wv], s_dd[wv], lane == 0);
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the coarse skip is only valid while a chunk spans few enough fine cells (see coarse_build_kernel): checked per ray
    // with its own direction, so unnormalised directions fall back to the exhaustive test instead of skipping cells
    float span_cells = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) span_cells = fmaxf(span_cells, fabsf(d[a]) * mp.chunk_cells[a]);
    const bool use_coarse = mp.use_coarse && (span_cells * 0.5f + 1.5f <= (float)kCoarseBlock);
    const bool use_fine = use_coarse && (span_cells * (16.5f / 64.0f) + 0.5f <= (float)kFineBlock);
    int32_t count = 0;
    const int res = mp.res;
    const float rf = (float)res;
    const uint32_t* fine = coarse ? coarse + coarse_words_of(res) : nullptr;
    const int nlw = n_live_words(mp.mask_words);
    uint64_t* rec = masks + r * (int64_t)(mp.mask_words + nlw);
    for (int g = 0; g < nlw; ++g) {
        // ---- phase A: lane q decides whether chunk 64g+q can hold a sample at all (range + dilated coarse grid):
        //      one ballot replaces up to 64 sequential chunk visits (what nerfacc's DDA skips cell by cell)
        const int q = g * 64 + lane;
        bool maybe = false;
        int run_at_chunk = 0;           // table of runs: the run that holds this lane's chunk start (phase B starts its look-ups there)
        if (q < mp.mask_words) {
            const int k0 = q * 64;
            int run = (lat.n > 0 && !lat.full) ? lat.find(k0) : 0;
            run_at_chunk = run;
            const float t_first = (lat.n > 0 && !lat.full) ? lat.at(run, k0) : lat(k0);
            const float t_mid = lat.after(run, k0 + 32), t_last = lat.after(run, k0 + 64);
            maybe = !(t_first > hi) && !(t_last < lo);
            if (maybe && use_coarse) {
                const float tc = t_mid;
                const int cr = res >> kCoarseShift;
                int cb[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float p = add_rn(o[a], mul_rn(d[a], tc));
                    const float u = mul_rn(mul_rn(sub_rn(p, mp.lo[a]), mp.inv_ext[a]), rf);
                    cb[a] = ((int)fminf(fmaxf(floorf(u), 0.0f), rf - 1.0f)) >> kCoarseShift;
                }
                const uint32_t bi = (uint32_t)((cb[0] * cr + cb[1]) * cr + cb[2]);
                maybe = (coarse[bi >> 5] >> (bi & 31)) & 1u;
                if (maybe && use_fine) {
                    const int fr = res >> kFineShift;
                    int run2 = run_at_chunk;                                 // (`run` has moved on to k0 + 64)
                    const uint32_t f1 = fine_block_at(mp, o, d, lat.after(run2, k0 + 16), rf, fr);
                    const uint32_t f3 = fine_block_at(mp, o, d, lat.after(run2, k0 + 48), rf, fr);
                    maybe = (((fine[f1 >> 5] >> (f1 & 31)) | (fine[f3 >> 5] >> (f3 & 31))) & 1u) != 0u;
                }
            }
        }
        uint64_t live = __ballot(maybe);
        uint64_t kept = 0;        // chunks that really hold samples
        // ---- phase B: the 64 lattice intervals of every surviving chunk, one per lane.  Four chunks per round: their
        //      occupancy words are independent gathers, issued together instead of one L2 latency after the other
        //      (a ray crosses ~5 live chunks around a surface: the dilated coarse grid is generous)
        for (uint64_t todo = live; todo;) {
            int qs[4]; uint32_t cis[4]; bool in_range[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                qs[u] = -1; cis[u] = 0u; in_range[u] = false;
                if (todo) {
                    qs[u] = g * 64 + (__ffsll((unsigned long long)todo) - 1);
                    todo &= todo - 1;
                    const int k = qs[u] * 64 + lane;
                    // (the chunk is wave uniform: ONE lane's run index for all -- v_readlane, whatever the exec mask)
                    int r_lo = __builtin_amdgcn_readlane(run_at_chunk, qs[u] & 63);
                    if (k < mp.max_steps) {
                        float ta;
                        // table of runs: look-ups start at the run of the chunk's first index instead of a binary search per lane (the
                        // ray's FIRST chunk crosses a dozen runs -- one or two per binade from 2^-11 up -- and keeps the search)
                        if (lat.n > 0 && !lat.full && qs[u] != 0) ta = lat.after(r_lo, k);
                        else ta = lat(k);
                        const float tb = mp.lattice_mode == PERF_LATTICE_REPEATED ? add_rn(ta, mp.step) : lattice_single(t0, k + 1, mp.step);
                        const float mid = mul_rn(add_rn(ta, tb), 0.5f);
                        if (mid >= lo && mid <= hi) {
                            int cell[3];
#pragma unroll
                            for (int a = 0; a < 3; ++a) {
                                const float p = add_rn(o[a], mul_rn(d[a], mid));
                                const float uu = mul_rn(mul_rn(sub_rn(p, mp.lo[a]), mp.inv_ext[a]), rf);
                                cell[a] = (int)fminf(fmaxf(floorf(uu), 0.0f), rf - 1.0f);
                            }
                            cis[u] = (uint32_t)((cell[0] * res + cell[1]) * res + cell[2]);
                            in_range[u] = true;
                        }
                    }
                }
            }
            uint32_t words[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) words[u] = in_range[u] ? bits[cis[u] >> 5] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (qs[u] < 0) continue;                 // (wave uniform)
                const bool mine = in_range[u] && ((words[u] >> (cis[u] & 31)) & 1u);
                const uint64_t m = __ballot(mine);
                if (m) {
                    kept |= 1ull << (qs[u] & 63);
                    if (lane == 0) rec[nlw + qs[u]] = m;
                    if (HEAD && count < ho.K && mine) {         // (chunks arrive in t order: `count` = samples before this chunk)
                        const int rank = count + __popcll(m & ((1ull << lane) - 1ull));
                        if (rank < ho.K) s_head[wv][rank] = qs[u] * 64 + lane;
                    }
                    count += __popcll(m);
                }
            }
        }
        if (lane == 0) rec[g] = kept;
    }
    if (lane == 0) counts[r] = count;
    if (HEAD) {
        const int have = count < ho.K ? count : ho.K;
        if (lane == 0) { ho.packed[2 * r] = (int32_t)(r * ho.K); ho.packed[2 * r + 1] = have; }
        __builtin_amdgcn_wave_barrier();
        if (lane < have) {                                       // the head rows: ONE copy of this code, outside the (unrolled) chunk loop
            const int64_t pos = r * ho.K + lane;
            const int k = s_head[wv][lane];
            const float a = lat(k), b = mp.lattice_mode == PERF_LATTICE_REPEATED ? add_rn(a, mp.step) : lattice_single(t0, k + 1, mp.step);
            ho.ts[pos] = a; ho.te[pos] = b; ho.ri[pos] = r;
            sample_point_store(ro + 3 * r, rd + 3 * r, a, b, ho.bb, ho.x01, ho.sel, pos);
        }
        if (lane >= have && lane < ho.K) {                       // padding rows: harmless inputs, selector 0
            const int64_t pos = r * ho.K + lane;
            ho.ts[pos] = 0.f; ho.te[pos] = 0.f; ho.ri[pos] = r;
            ho.x01[3 * pos] = 0.5f; ho.x01[3 * pos + 1] = 0.5f; ho.x01[3 * pos + 2] = 0.5f; ho.sel[pos] = 0;
        }
    }
}

// ---- large launches on ONE lattice (eval frames: no jitter, t_k precomputed): a ray per LANE for everything that is uniform
// per ray.  The wave-per-ray kernel above is VALU bound on such launches -- 488 vector instructions per ray on a 524,288-ray
// frame (SQ_ACTIVE_INST_VALU 95 %), of which ~140 are the ray's set-up (slab test with three IEEE divisions, skip-grid validity)
// executed by 64 lanes on uniform values and ~100 the chunk ballot that uses 47 of 64 lanes.  Here a wave takes 64 rays: set-up
// and the chunk decisions (range + dilated coarse grid; the chunk's lattice points are the same for every ray: scalar loads)
// run lane-parallel -- the coarse-grid words of eight chunks in flight per round --, then the wave walks its rays' surviving
// chunks exactly as above (64 intervals of a chunk, one per lane).  Same arithmetic per ray, same masks, counts and head rows.
// Needs enough rays to fill the chip with 64-ray waves (the host picks it from kSharedMinRays rays on) and one live word per
// ray (max_steps <= 4096).
constexpr int64_t kSharedMinRays = 262144;
constexpr int kSharedHeadMax = 16;                 // head rows per ray this kernel can hold (the host's STRIDED_HEAD_MAX)
template <bool HEAD>
__global__ __launch_bounds__(64) void march_count_shared_kernel(MarchParams mp, const float* __restrict__ ro, const float* __restrict__ rd,
                                                                int64_t n_rays, const uint32_t* __restrict__ bits,
                                                                const uint32_t* __restrict__ coarse, uint64_t* __restrict__ masks,
                                                                int32_t* __restrict__ counts, HeadOut ho, float t0_base,
                                                                const float* __restrict__ lat_full) {
    __shared__ int32_t s_head[HEAD ? 64 * kSharedHeadMax : 1];                // [ray of the wave][rank]: lattice index of the ray's first K samples
    const int lane = threadIdx.x;
    const int64_t r_base = (int64_t)blockIdx.x * 64;
    const int res = mp.res;
    const float rf = (float)res;
    // ---- per lane: the ray's range and its chunk decisions
    uint64_t live_mine = 0;
    float lo_mine = 0.f, hi_mine = 0.f;
    {
        const int64_t r = r_base + lane;
        if (r < n_rays) {
            const float o[3] = {ro[3 * r], ro[3 * r + 1], ro[3 * r + 2]};
            const float d[3] = {rd[3 * r], rd[3 * r + 1], rd[3 * r + 2]};
            float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float inv = __fdiv_rn(1.0f, d[a]);
                const float t1 = mul_rn(sub_rn(mp.lo[a], o[a]), inv), t2 = mul_rn(sub_rn(mp.hi[a], o[a]), inv);
                const float l = fminf(t1, t2), h = fmaxf(t1, t2);
                tmin = (a == 0) ? l : fmaxf(tmin, l);
                tmax = (a == 0) ? h : fminf(tmax, h);
            }
            lo_mine = fmaxf(tmin, t0_base); hi_mine = fminf(tmax, mp.far_plane);
            float span_cells = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) span_cells = fmaxf(span_cells, fabsf(d[a]) * mp.chunk_cells[a]);
            const bool use_coarse = mp.use_coarse && (span_cells * 0.5f + 1.5f <= (float)kCoarseBlock);
            const bool use_fine = use_coarse && (span_cells * (16.5f / 64.0f) + 0.5f <= (float)kFineBlock);
            const uint32_t* fine = coarse ? coarse + coarse_words_of(res) : nullptr;
            const int fr = res >> kFineShift;
            const int cr = res >> kCoarseShift;
            for (int q0 = 0; q0 < mp.mask_words; q0 += 8) {
                uint32_t bi[8]; bool in[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = q0 + u;
                    bi[u] = 0u; in[u] = false;
                    if (q < mp.mask_words) {                                 // (uniform)
                        const int k0 = q * 64;
                        const float t_first = lat_full[k0], t_mid = lat_full[k0 + 32], t_last = lat_full[k0 + 64];     // (uniform addresses)
                        in[u] = !(t_first > hi_mine) && !(t_last < lo_mine);
                        if (in[u] && use_coarse) {
                            int cb[3];
#pragma unroll
                            for (int a = 0; a < 3; ++a) {
                                const float p = add_rn(o[a], mul_rn(d[a], t_mid));
                                const float uu = mul_rn(mul_rn(sub_rn(p, mp.lo[a]), mp.inv_ext[a]), rf);
                                cb[a] = ((int)fminf(fmaxf(floorf(uu), 0.0f), rf - 1.0f)) >> kCoarseShift;
                            }
                            bi[u] = (uint32_t)((cb[0] * cr + cb[1]) * cr + cb[2]);
                        }
                    }
                }
                uint32_t w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = (in[u] && use_coarse) ? coarse[bi[u] >> 5] : 0u;
                // survivors of the coarse test: the fine grid at two lattice points of the chunk (few per ray: a divergent tail)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    bool alive = in[u] && (!use_coarse || ((w[u] >> (bi[u] & 31u)) & 1u));
                    if (alive && use_fine) {
                        const int k0 = (q0 + u) * 64;
                        const uint32_t f1 = fine_block_at(mp, o, d, lat_full[k0 + 16], rf, fr);
                        const uint32_t f3 = fine_block_at(mp, o, d, lat_full[k0 + 48], rf, fr);
                        alive = (((fine[f1 >> 5] >> (f1 & 31)) | (fine[f3 >> 5] >> (f3 & 31))) & 1u) != 0u;
                    }
                    if (alive) live_mine |= 1ull << (q0 + u);
                }
            }
        }
    }
    // ---- the wave walks its rays.  What a ray leaves behind besides its chunk masks -- the record's first word, its count, its head rows
    //      -- is kept by the ray's LANE and written after the walk by all lanes at once: one pass of coalesced stores and lane-parallel
    //      sample_point_store instead of 64 single-lane tails (~55 of the ~310 vector instructions per ray were that tail)
    const uint32_t live_lo32 = (uint32_t)live_mine, live_hi32 = (uint32_t)(live_mine >> 32);
    char* rec_wave = reinterpret_cast<char*>(masks + r_base * (int64_t)(mp.mask_words + 1));
    const uint32_t rec_stride = (uint32_t)__builtin_amdgcn_readfirstlane((mp.mask_words + 1) * 8);       // (a scalar: s * rec_stride stays on the scalar unit)
    int32_t count_mine = 0;
    uint32_t kept_lo = 0u, kept_hi = 0u;
    for (int s = 0; s < 64; ++s) {
        const int64_t r = r_base + s;
        if (r >= n_rays) break;
        const uint64_t live = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)live_lo32, s) |
                              ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)live_hi32, s) << 32);
        if (!live) continue;                                         // (count_mine / kept of that lane stay 0)
        const float lo = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(lo_mine), s));
        const float hi = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(hi_mine), s));
        // (the record of ray s from the wave's base + a 32-bit offset: formed as r * (mask_words + 1) the compiler multiplied 64-bit
        //  integers on the vector unit for every ray)
        char* rec = rec_wave + (uint32_t)s * rec_stride;
        int32_t count = 0;
        uint64_t kept = 0;
        const float o[3] = {ro[3 * r], ro[3 * r + 1], ro[3 * r + 2]};
        const float d[3] = {rd[3 * r], rd[3 * r + 1], rd[3 * r + 2]};
        for (uint64_t todo = live; todo;) {
            int qs[4]; uint32_t cis[4]; bool in_range[4]; float ta[4];
            // the lattice points of up to four surviving chunks are requested TOGETHER (as written before -- one load inside each chunk's
            // branch -- every chunk waited for its own round trip); the table holds 64 points beyond the last chunk: no bound test here
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                qs[u] = -1; ta[u] = 0.f;
                if (todo) {
                    qs[u] = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    ta[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lat_full) + (((uint32_t)qs[u] * 64u + (uint32_t)lane) << 2));
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cis[u] = 0u; in_range[u] = false;
                if (qs[u] < 0) continue;                 // (wave uniform)
                const int k = qs[u] * 64 + lane;
                if (k < mp.max_steps) {
                    const float tb = mp.lattice_mode == PERF_LATTICE_REPEATED ? add_rn(ta[u], mp.step) : lattice_single(t0_base, k + 1, mp.step);
                    const float mid = mul_rn(add_rn(ta[u], tb), 0.5f);
                    if (mid >= lo && mid <= hi) {
                        uint32_t cell[3];
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const float p = add_rn(o[a], mul_rn(d[a], mid));
                            const float uu = mul_rn(mul_rn(sub_rn(p, mp.lo[a]), mp.inv_ext[a]), rf);
                            cell[a] = (uint32_t)(int)fminf(fmaxf(floorf(uu), 0.0f), rf - 1.0f);
                        }
                        // (res <= 1024: both products fit 24-bit operands -- two full-rate multiply-adds instead of two 64-bit ones)
                        cis[u] = (uint32_t)__umul24((uint32_t)__umul24(cell[0], (uint32_t)res) + cell[1], (uint32_t)res) + cell[2];
                        in_range[u] = true;
                    }
                }
            }
            uint32_t words[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                words[u] = in_range[u] ? *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(bits) + ((cis[u] >> 5) << 2)) : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (qs[u] < 0) continue;                 // (wave uniform)
                const bool mine = in_range[u] && ((words[u] >> (cis[u] & 31)) & 1u);
                const uint64_t m = __ballot(mine);
                if (m) {
                    kept |= 1ull << qs[u];
                    if (lane == 0) *reinterpret_cast<uint64_t*>(rec + ((uint32_t)(1 + qs[u]) << 3)) = m;
                    if (HEAD && count < ho.K && mine) {
                        const int rank = count + __popcll(m & ((1ull << lane) - 1ull));
                        if (rank < ho.K) s_head[s * kSharedHeadMax + rank] = qs[u] * 64 + lane;
                    }
                    count += __popcll(m);
                }
            }
        }
        if (lane == s) { count_mine = count; kept_lo = (uint32_t)kept; kept_hi = (uint32_t)(kept >> 32); }
    }
    // ---- per lane again: record heads, counts, head rows
    const int64_t r = r_base + lane;
    if (r >= n_rays) return;
    masks[r * (int64_t)(mp.mask_words + 1)] = (uint64_t)kept_lo | ((uint64_t)kept_hi << 32);
    counts[r] = count_mine;
    if (HEAD) {
        const int have = count_mine < ho.K ? count_mine : ho.K;
        ho.packed[2 * r] = (int32_t)(r * ho.K); ho.packed[2 * r + 1] = have;
        for (int j = 0; j < ho.K; ++j) {                             // (row j of 64 rays at a time)
            const int64_t pos = r * ho.K + j;
            if (j < have) {
                const int k = s_head[lane * kSharedHeadMax + j];
                const float a = lat_full[k], b = mp.lattice_mode == PERF_LATTICE_REPEATED ? add_rn(a, mp.step) : lattice_single(t0_base, k + 1, mp.step);
                ho.ts[pos] = a; ho.te[pos] = b; ho.ri[pos] = r;
                sample_point_store(ro + 3 * r, rd + 3 * r, a, b, ho.bb, ho.x01, ho.sel, pos);
            } else {                                                 // padding rows: harmless inputs, selector 0
                ho.ts[pos] = 0.f; ho.te[pos] = 0.f; ho.ri[pos] = r;
                ho.x01[3 * pos] = 0.5f; ho.x01[3 * pos + 1] = 0.5f; ho.x01[3 * pos + 2] = 0.5f; ho.sel[pos] = 0;
            }
        }
    }
}

__global__ __launch_bounds__(256) void march_write_kernel(const float* __restrict__ t0s, int64_t n_rays, float step,
                                                          int32_t mask_words, const uint64_t* __restrict__ masks,
                                                          const int32_t* __restrict__ counts,
                                                          const int32_t* __restrict__ offsets, int64_t capacity,
                                                          int64_t* __restrict__ ray_indices, float* __restrict__ ts,
                                                          float* __restrict__ te, int32_t* __restrict__ packed,
                                                          const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                          Aabb bb, float* __restrict__ x01, uint8_t* __restrict__ sel,
                                                          int32_t rank_lo, float t0_scale, float t0_base, int lattice_mode, SharedRuns sr,
                                                          const float* __restrict__ lat_full) {
    // (all rays on one lattice -- eval renders --: the host-built table of runs, see march_count_kernel; per-ray origins walk)
    __shared__ int32_t w_ks[kMaxRuns];
    __shared__ uint32_t w_bs[kMaxRuns], w_dd[kMaxRuns];
    if (sr.n > 0) {
        if ((int)threadIdx.x < sr.n) { w_ks[threadIdx.x] = sr.ks[threadIdx.x]; w_bs[threadIdx.x] = sr.bs[threadIdx.x]; w_dd[threadIdx.x] = sr.dd[threadIdx.x]; }
        __syncthreads();
    }
    const int32_t* ray_rows = (t0s && lat_full) ? reinterpret_cast<const int32_t*>(lat_full) : nullptr;      // per-ray tables (see march_count_kernel)
    if (ray_rows) lat_full = nullptr;
    LatticeRuns tab;
    tab.n = sr.n; tab.ks = w_ks; tab.bs = w_bs; tab.dd = w_dd; tab.t0 = 0.f; tab.step = step; tab.mode = lattice_mode; tab.full = lat_full;
    // Writes the samples of rank [rank_lo, rank_lo + counts[r]) of every ray (rank = position among the ray's samples in t
    // order) to offsets[r]...: rank_lo = 0 and counts = the march counts is the plain expansion; the two-phase sampler
    // writes the first K samples of every ray first and the rest of the rays that are still alive later.
    // With at most 16 samples to write for each ray of an aligned group of four, every ray gets a quarter wave whose lanes
    // walk the 64 bits of a mask word in four steps; otherwise every ray has its own wave (the kernel has no cross-lane
    // operation: a team is only a lane -> (ray, bit) mapping).
    // (two launch shapes as in composite.hip:for_rays_of_wave: n_rays / 4 waves for large batches -- wave w serves rays
    //  4w..4w+3 --, one wave per ray for small ones, where in the quarter-wave case the first wave of an aligned group of
    //  four rays serves all four and the other three retire at once)
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool four_per_wave = (int64_t)gridDim.x * 4 < n_rays;          // n_rays / 4 waves launched (large batches), else one per ray
    const int64_t r0 = four_per_wave ? w * 4 : (w & ~(int64_t)3);
    if (r0 >= n_rays) return;
    const int64_t rq = r0 + (lane >> 4);
    const bool small = __ballot(rq < n_rays && counts[rq] > 16) == 0ull;
    if (small && !four_per_wave && (w & 3) != 0) return;
    const int W = small ? 16 : 64;
    const int l = small ? (lane & 15) : lane;
    for (int q = 0; q < ((four_per_wave && !small) ? 4 : 1); ++q) {
        const int64_t r = small ? rq : (four_per_wave ? r0 + q : w);
        if (r >= n_rays) continue;
        int32_t cnt = counts[r];
        const int32_t off = offsets[r];
        if ((int64_t)off + cnt > capacity) cnt = (int32_t)(capacity > off ? capacity - off : 0);   // truncated batch
        if (l == 0) { packed[2 * r] = off; packed[2 * r + 1] = cnt; }
        if (cnt == 0) continue;
        const float t0 = lattice_origin(t0s, r, t0_scale, t0_base);
        int64_t run = (int64_t)off - rank_lo;            // output position of rank 0 (may lie before `off`)
        const int64_t end = (int64_t)off + cnt;
        const int nlw = n_live_words(mask_words);
        const uint64_t* rec = masks + r * (int64_t)(mask_words + nlw);
        for (int g = 0; g < nlw && run < end; ++g) {
            for (uint64_t todo = rec[g]; todo && run < end; todo &= todo - 1) {
                const int qq = g * 64 + (__ffsll((unsigned long long)todo) - 1);
                const uint64_t m = rec[nlw + qq];
                for (int bit = l; bit < 64; bit += W) {
                    if (!((m >> bit) & 1ull)) continue;
                    const uint64_t below = (bit == 0) ? 0ull : (~0ull >> (64 - bit));
                    const int64_t pos = run + __popcll(m & below);
                    if (pos >= off && pos < end) {
                        const int k = qq * 64 + bit;
                        const float a = lat_full ? lat_full[k]
                                                 : (tab.n > 0 ? tab.at(tab.find(k), k)
                                                              : ((ray_rows && lattice_mode == PERF_LATTICE_REPEATED) ? runs_row_at(ray_rows + r * (int64_t)kRunsStride, k, t0, step)
                                                                                                                     : lattice(t0, k, step, lattice_mode)));
                        const float b = lattice_mode == PERF_LATTICE_REPEATED ? add_rn(a, step) : lattice_single(t0, k + 1, step);
                        ts[pos] = a;
                        te[pos] = b;
                        ray_indices[pos] = r;
                        if (x01) sample_point_store(rays_o + 3 * r, rays_d + 3 * r, a, b, bb, x01, sel, pos);     // (= perf_points_from_rays)
                    }
                }
                run += __popcll(m);
            }
        }
    }
}

// Large launches on ONE lattice (eval frames, as march_count_shared_kernel): a wave takes 64 rays.  Every ray's LANE reads its count and
// offset and stores its packed_info pair -- coalesced -- and the wave then walks only the rays that HAVE samples to write, four at a
// time on quarter waves (or one after the other with all 64 lanes when one of the four writes more than 16 samples).  On an eval frame of a trained scene most rays end inside their head (the two-phase sampler's tail
// counts are zero): march_write_kernel gave each of them a quarter wave that loaded two words, stored two and retired.  Same rows,
// same values.
__global__ __launch_bounds__(256) void march_write_shared_kernel(int64_t n_rays, float step, int32_t mask_words, const uint64_t* __restrict__ masks,
                                                                 const int32_t* __restrict__ counts, const int32_t* __restrict__ offsets,
                                                                 int64_t capacity, int64_t* __restrict__ ray_indices, float* __restrict__ ts,
                                                                 float* __restrict__ te, int32_t* __restrict__ packed,
                                                                 const float* __restrict__ rays_o, const float* __restrict__ rays_d, Aabb bb,
                                                                 float* __restrict__ x01, uint8_t* __restrict__ sel, int32_t rank_lo,
                                                                 float t0_base, int lattice_mode, const float* __restrict__ lat_full) {
    const int lane = threadIdx.x & 63;
    const int64_t r_base = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * 64;
    if (r_base >= n_rays) return;
    const int64_t rl = r_base + lane;
    int32_t cnt_mine = 0, off_mine = 0;
    if (rl < n_rays) {
        cnt_mine = counts[rl]; off_mine = offsets[rl];
        if ((int64_t)off_mine + cnt_mine > capacity) cnt_mine = (int32_t)(capacity > off_mine ? capacity - off_mine : 0);   // truncated batch
        packed[2 * rl] = off_mine; packed[2 * rl + 1] = cnt_mine;
    }
    const int nlw = n_live_words(mask_words);
    // one ray's samples, by a team of W lanes that walks the 64 bits of a mask word in 64 / W steps (march_write_kernel's loop)
    auto walk = [&](auto wc, int64_t r, int32_t cnt, int32_t off, int l) {
        constexpr int W = decltype(wc)::value;
        int64_t run = (int64_t)off - rank_lo;            // output position of rank 0 (may lie before `off`)
        const int64_t end = (int64_t)off + cnt;
        const uint64_t* rec = masks + r * (int64_t)(mask_words + nlw);
        for (int g = 0; g < nlw && run < end; ++g) {
            for (uint64_t todo = rec[g]; todo && run < end; todo &= todo - 1) {
                const int qq = g * 64 + (__ffsll((unsigned long long)todo) - 1);
                const uint64_t m = rec[nlw + qq];
                for (int bit = l; bit < 64; bit += W) {
                    if (!((m >> bit) & 1ull)) continue;
                    const uint64_t below = (bit == 0) ? 0ull : (~0ull >> (64 - bit));
                    const int64_t pos = run + __popcll(m & below);
                    if (pos >= off && pos < end) {
                        const int k = qq * 64 + bit;
                        const float a = lat_full[k];
                        const float b = lattice_mode == PERF_LATTICE_REPEATED ? add_rn(a, step) : lattice_single(t0_base, k + 1, step);
                        ts[pos] = a;
                        te[pos] = b;
                        ray_indices[pos] = r;
                        if (x01) sample_point_store(rays_o + 3 * r, rays_d + 3 * r, a, b, bb, x01, sel, pos);
                    }
                }
                run += __popcll(m);
            }
        }
    };
    // the rays that have samples, four at a time: a quarter wave each while all four write at most 16 samples (the tail of a two-phase
    // sampler, a thin shell), else one after the other with all 64 lanes
    for (uint64_t rays = __ballot(cnt_mine > 0); rays;) {
        int sq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sq[j] = -1;
            if (rays) { sq[j] = __ffsll((unsigned long long)rays) - 1; rays &= rays - 1; }
        }
        const int quarter = lane >> 4;
        const int s_mine = quarter == 0 ? sq[0] : (quarter == 1 ? sq[1] : (quarter == 2 ? sq[2] : sq[3]));
        const int32_t cnt_q = __shfl(cnt_mine, s_mine < 0 ? 0 : s_mine), off_q = __shfl(off_mine, s_mine < 0 ? 0 : s_mine);
        if (__ballot(s_mine >= 0 && cnt_q > 16) == 0ull) {
            if (s_mine >= 0) walk(std::integral_constant<int, 16>{}, r_base + s_mine, cnt_q, off_q, lane & 15);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sq[j] >= 0)
                    walk(std::integral_constant<int, 64>{}, r_base + sq[j], __builtin_amdgcn_readlane(cnt_mine, sq[j]), __builtin_amdgcn_readlane(off_mine, sq[j]), lane);
        }
    }
}

// ---------------- exclusive scan of int32 (counts -> offsets) -----------------------------------
constexpr int kScanBlock = 1024;   // elements per block (256 threads x 4)

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int y = __shfl_up(v, off);
        if (lane >= off) v += y;
    }
    return v;
}

// block-wide exclusive scan of one int per thread (256 threads); returns exclusive value, total in *total
__device__ __forceinline__ int block_excl_scan(int v, int* lds4, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = wave_incl_scan(v, lane);
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w2 = 0; w2 < wave; ++w2) base += lds4[w2];
    *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const int32_t* __restrict__ in, int64_t n, int64_t* __restrict__ sums) {
    __shared__ int lds4[4];
    const int64_t base = (int64_t)blockIdx.x * kScanBlock + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < n) s += in[base + k];
    int total;
    block_excl_scan(s, lds4, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// Second (and last) launch of the large scan: block b adds up the sums of the blocks before it itself (at most a few thousand
// 8-byte values from L2 -- a third launch that scans them cost 7.4 us, twice per 512x1024 frame), then scans its own 1024
// elements; the last block leaves the total.
// (Beyond kScanApplyMaxBlocks blocks -- more than 2 M elements -- the re-summing would grow quadratically: 8 M elements = 8,192 blocks =
//  33 M loads.  A middle launch then turns the block sums into exclusive prefixes once (scan_sums_kernel) and every block reads its own.)
constexpr int64_t kScanApplyMaxBlocks = 2048;
__global__ __launch_bounds__(1024) void scan_sums_kernel(int64_t* __restrict__ sums, int64_t nb) {
    __shared__ long long wave_tot[16];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
        const int64_t j = b0 + threadIdx.x;
        const long long v = j < nb ? sums[j] : 0;
        long long inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const long long t = __shfl_up(inc, off); if (lane >= off) inc += t; }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        long long before = carry_s;
        for (int w2 = 0; w2 < wave; ++w2) before += wave_tot[w2];
        if (j < nb) sums[j] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n,
                                                         const int64_t* __restrict__ sums, int64_t* __restrict__ total_out,
                                                         int64_t bias, int64_t* __restrict__ total_biased, int sums_are_prefixes) {
    __shared__ int lds4[4];
    __shared__ long long pre4[4];
    long long before = 0;
    if (sums_are_prefixes) { if (threadIdx.x == 0) before = sums[blockIdx.x]; }
    else for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += 256) before += sums[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) before += __shfl_xor(before, off);
    if ((threadIdx.x & 63) == 0) pre4[threadIdx.x >> 6] = before;
    const int64_t base = (int64_t)blockIdx.x * kScanBlock + threadIdx.x * 4;
    int v[4];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    int total;
    const int ex0 = block_excl_scan(s, lds4, &total);       // (its barriers order pre4 as well)
    const long long prefix = (pre4[0] + pre4[1]) + (pre4[2] + pre4[3]);
    int ex = ex0 + (int)prefix;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        *total_out = prefix + total;
        if (total_biased) *total_biased = prefix + total + bias;
    }
}

// n <= kScanSmall: ONE launch of ceil(n / 1024) workgroups.  Block b first sums every element BEFORE its segment itself
// (redundant, coalesced reads of at most 256 KiB from L2 -- far cheaper than a second launch or a cross-block handshake),
// then scans its own 1024 elements.  (The 8,192 per-ray counts of a training batch take 8 blocks; a single-workgroup
// scan of the 32,768 counts of an eval batch cost 27 us per call, 0.86 ms per 512x1024 frame.)
constexpr int kScanSmall = 65536;
__global__ __launch_bounds__(1024) void scan_small_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n,
                                                          int64_t* __restrict__ total_out, int64_t bias, int64_t* __restrict__ total_biased) {
    __shared__ long long wave_sum[16];
    __shared__ int wave_tot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t seg_lo = (int64_t)blockIdx.x * 1024;
    // ---- sum of everything before the segment
    // (seg_lo is a multiple of 1024: 16-byte loads, four independent accumulators -- the loop is latency bound)
    long long s = 0;
    {
        const int4* in4 = reinterpret_cast<const int4*>(in);
        const int64_t n4 = seg_lo >> 2;
        long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int64_t i = threadIdx.x;
        if ((reinterpret_cast<uintptr_t>(in) & 15) == 0) {
            for (; i + 3 * 1024 < n4; i += 4 * 1024) {
                const int4 a = in4[i], b = in4[i + 1024], c = in4[i + 2048], d = in4[i + 3072];
                s0 += (long long)a.x + a.y + a.z + a.w; s1 += (long long)b.x + b.y + b.z + b.w;
                s2 += (long long)c.x + c.y + c.z + c.w; s3 += (long long)d.x + d.y + d.z + d.w;
            }
            for (; i < n4; i += 1024) { const int4 a = in4[i]; s0 += (long long)a.x + a.y + a.z + a.w; }
        } else {
            for (int64_t j = threadIdx.x; j < seg_lo; j += 1024) s0 += in[j];
        }
        s = (s0 + s1) + (s2 + s3);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) wave_sum[wave] = s;
    // ---- scan of the segment: wave scans + the 16 wave totals
    const int64_t i = seg_lo + threadIdx.x;
    const int v = (i < n) ? in[i] : 0;
    const int inc = wave_incl_scan(v, lane);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    long long base = 0;
    int before = 0, seg_total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) { base += wave_sum[w2]; if (w2 < wave) before += wave_tot[w2]; seg_total += wave_tot[w2]; }
    if (i < n) out[i] = (int)base + before + inc - v;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        *total_out = base + seg_total;
        if (total_biased) *total_biased = base + seg_total + bias;
    }
}

}  // namespace perf

using namespace perf;

static inline int chunk_words(int32_t max_steps) { return max_steps > 0 ? (max_steps + 63) / 64 : 0; }

// uint64 words per ray in the mask buffer: one mask word per 64-interval chunk plus the live-chunk words
extern "C" int64_t perf_occ_mask_words(int32_t max_steps) {
    const int mw = chunk_words(max_steps);
    return mw + (mw + 63) / 64;
}

extern "C" int64_t perf_occ_coarse_words(int32_t res) {
    if (res <= 0 || (res % 8) != 0) return 0;
    return coarse_words_of(res) + fine_words_of(res);           // the dilated 4^3-block grid, then the dilated 2^3-block grid
}

extern "C" int perf_occ_build_coarse(const uint32_t* occ_bits, int32_t res, uint32_t* coarse, void* stream) {
    PERF_REQUIRE(occ_bits && coarse && res > 0 && (res % 8) == 0, "perf_occ_build_coarse: res must be a positive multiple of 8");
    const int64_t words = perf_occ_coarse_words(res);
    hipError_t e = hipMemsetAsync(coarse, 0, words * sizeof(uint32_t), as_stream(stream));
    if (e != hipSuccess) { set_error("perf_occ_build_coarse: memset failed"); return PERF_E_LAUNCH; }
    const int cr = res / kCoarseBlock;
    hipLaunchKernelGGL(coarse_build_kernel, dim3((unsigned)div_up((int64_t)cr * cr * cr, 256)), dim3(256), 0, as_stream(stream),
                       occ_bits, (int)res, coarse);
    const int64_t fr = res / kFineBlock;
    hipLaunchKernelGGL(fine_build_kernel, dim3((unsigned)div_up(fr * fr * fr, 256)), dim3(256), 0, as_stream(stream),
                       occ_bits, (int)res, coarse + coarse_words_of(res));
    PERF_LAUNCH_CHECK("perf_occ_build_coarse");
    return PERF_OK;
}

static int march_count_launch(const float* rays_o, const float* rays_d, const float* t0, float t0_scale, float t0_base, int64_t n_rays,
                              const uint32_t* occ_bits, const uint32_t* occ_coarse, int32_t res, const float* aabb, float far_plane,
                              float step, int32_t max_steps, int32_t lattice_mode, const float* lattice_table, uint64_t* masks, int32_t* counts,
                              const HeadOut* head, void* stream) {
    PERF_REQUIRE(lattice_mode == PERF_LATTICE_SINGLE || lattice_mode == PERF_LATTICE_REPEATED, "bad lattice mode %d", (int)lattice_mode);
    PERF_REQUIRE(n_rays >= 0 && res > 0 && res <= 1024 && max_steps > 0 && step > 0.f, "perf_occ_march_count: bad arguments");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(rays_o && rays_d && occ_bits && aabb && masks && counts, "NULL pointer");
    MarchParams mp;
    for (int a = 0; a < 3; ++a) {
        mp.lo[a] = aabb[a]; mp.hi[a] = aabb[3 + a];
        mp.inv_ext[a] = 1.0f / (aabb[3 + a] - aabb[a]);
    }
    mp.far_plane = far_plane; mp.step = step; mp.res = res; mp.max_steps = max_steps; mp.lattice_mode = lattice_mode;
    mp.mask_words = chunk_words(max_steps);
    for (int a = 0; a < 3; ++a) mp.chunk_cells[a] = 64.0f * step * mp.inv_ext[a] * (float)res;
    mp.use_coarse = (occ_coarse != nullptr && (res % 8) == 0) ? 1 : 0;      // (+ the per-ray span test in the kernel)
    SharedRuns sr;
    sr.n = 0;
    PERF_REQUIRE(!lattice_table || t0 == nullptr || lattice_mode == PERF_LATTICE_REPEATED, "per-ray lattice tables (t0 != NULL) exist for the repeated lattice only");
    if (lattice_mode == PERF_LATTICE_REPEATED && t0 == nullptr && !lattice_table) shared_runs_build(t0_base, step, mp.mask_words * 64 + 64, &sr);
    if (t0 == nullptr && lattice_table != nullptr && mp.mask_words <= 64 && n_rays >= kSharedMinRays && (!head || head->K <= kSharedHeadMax)) {      // (see march_count_shared_kernel)
        if (head)
            hipLaunchKernelGGL(march_count_shared_kernel<true>, dim3((unsigned)div_up(n_rays, 64)), dim3(64), 0, as_stream(stream), mp, rays_o, rays_d,
                               n_rays, occ_bits, occ_coarse, masks, counts, *head, t0_base, lattice_table);
        else
            hipLaunchKernelGGL(march_count_shared_kernel<false>, dim3((unsigned)div_up(n_rays, 64)), dim3(64), 0, as_stream(stream), mp, rays_o, rays_d,
                               n_rays, occ_bits, occ_coarse, masks, counts, HeadOut{}, t0_base, lattice_table);
        PERF_LAUNCH_CHECK("perf_occ_march_count");
        return PERF_OK;
    }
    if (head)
        hipLaunchKernelGGL(march_count_kernel<true>, dim3((unsigned)div_up(n_rays, 4)), dim3(256), 0, as_stream(stream), mp, rays_o,
                           rays_d, t0, n_rays, occ_bits, occ_coarse, masks, counts, *head, t0_scale, t0_base, sr, lattice_table);
    else
        hipLaunchKernelGGL(march_count_kernel<false>, dim3((unsigned)div_up(n_rays, 4)), dim3(256), 0, as_stream(stream), mp, rays_o,
                           rays_d, t0, n_rays, occ_bits, occ_coarse, masks, counts, HeadOut{}, t0_scale, t0_base, sr, lattice_table);
    PERF_LAUNCH_CHECK("perf_occ_march_count");
    return PERF_OK;
}

extern "C" int perf_occ_march_count(const float* rays_o, const float* rays_d, const float* t0, float t0_scale, float t0_base,
                                    int64_t n_rays, const uint32_t* occ_bits, const uint32_t* occ_coarse, int32_t res,
                                    const float* aabb, float far_plane, float step, int32_t max_steps, int32_t lattice_mode,
                                    const float* lattice_table, uint64_t* masks, int32_t* counts, void* stream) {
    return march_count_launch(rays_o, rays_d, t0, t0_scale, t0_base, n_rays, occ_bits, occ_coarse, res, aabb, far_plane, step, max_steps,
                              lattice_mode, lattice_table, masks, counts, nullptr, stream);
}

extern "C" int perf_occ_march_count_head(const float* rays_o, const float* rays_d, const float* t0, float t0_scale, float t0_base,
                                         int64_t n_rays, const uint32_t* occ_bits, const uint32_t* occ_coarse, int32_t res, const float* aabb,
                                         float far_plane, float step, int32_t max_steps, int32_t lattice_mode, const float* lattice_table,
                                         uint64_t* masks, int32_t* counts, int32_t head_k, int64_t* ray_indices, float* t_starts, float* t_ends, int32_t* packed_info,
                                         const float* points_aabb6, float* x01, uint8_t* sel, void* stream) {
    PERF_REQUIRE(head_k >= 1 && head_k <= 64, "perf_occ_march_count_head: head_k must be in [1, 64]");
    PERF_REQUIRE(n_rays == 0 || (ray_indices && t_starts && t_ends && packed_info && points_aabb6 && x01 && sel), "NULL pointer");
    PERF_REQUIRE(n_rays * (int64_t)head_k < ((int64_t)1 << 31), "perf_occ_march_count_head: n_rays * head_k exceeds int32 offsets");
    HeadOut ho;
    ho.K = head_k; ho.ri = ray_indices; ho.ts = t_starts; ho.te = t_ends; ho.packed = packed_info; ho.x01 = x01; ho.sel = sel;
    if (n_rays > 0) for (int a = 0; a < 3; ++a) { ho.bb.lo[a] = points_aabb6[a]; ho.bb.hi[a] = points_aabb6[3 + a]; }
    return march_count_launch(rays_o, rays_d, t0, t0_scale, t0_base, n_rays, occ_bits, occ_coarse, res, aabb, far_plane, step, max_steps,
                              lattice_mode, lattice_table, masks, counts, &ho, stream);
}

extern "C" int64_t perf_scan_workspace_bytes(int64_t n) { return (div_up(n > 0 ? n : 1, kScanBlock) + 1) * (int64_t)sizeof(int64_t); }

extern "C" int perf_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t* total, int64_t n, int64_t total_bias,
                                       int64_t* total_biased, void* workspace, int64_t workspace_bytes, void* stream) {
    PERF_REQUIRE(n >= 0 && total, "perf_exclusive_scan_i32: bad arguments");
    if (n == 0) {
        PERF_REQUIRE(!total_biased, "perf_exclusive_scan_i32: a biased total needs n > 0");
        hipError_t e = hipMemsetAsync(total, 0, sizeof(int64_t), as_stream(stream));
        if (e != hipSuccess) { set_error("memset failed"); return PERF_E_LAUNCH; }
        return PERF_OK;
    }
    PERF_REQUIRE(in && out && workspace, "NULL pointer");
    PERF_REQUIRE(workspace_bytes >= perf_scan_workspace_bytes(n), "scan workspace too small");
    PERF_REQUIRE(in != out, "perf_exclusive_scan_i32: in-place scan is not supported");
    if (n <= kScanSmall) {
        hipLaunchKernelGGL(scan_small_kernel, dim3((unsigned)div_up(n, 1024)), dim3(1024), 0, as_stream(stream), in, out, n, total, total_bias,
                           total_biased);
        PERF_LAUNCH_CHECK("perf_exclusive_scan_i32");
        return PERF_OK;
    }
    const int64_t nb = div_up(n, kScanBlock);
    int64_t* sums = (int64_t*)workspace;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), in, n, sums);
    const int prefixes = nb > kScanApplyMaxBlocks ? 1 : 0;
    if (prefixes) hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, as_stream(stream), sums, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), in, out, n, sums, total, total_bias,
                       total_biased, prefixes);
    PERF_LAUNCH_CHECK("perf_exclusive_scan_i32");
    return PERF_OK;
}

extern "C" int perf_occ_march_write(const float* t0, float t0_scale, float t0_base, int64_t n_rays, float step, int32_t max_steps,
                                    int32_t lattice_mode, const float* lattice_table, const uint64_t* masks,
                                    const int32_t* counts, const int32_t* offsets, int64_t capacity, int64_t* ray_indices,
                                    float* t_starts, float* t_ends, int32_t* packed_info, void* stream) {
    PERF_REQUIRE(n_rays >= 0 && max_steps > 0 && capacity >= 0, "perf_occ_march_write: bad arguments");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(masks && counts && offsets && packed_info, "NULL pointer");
    PERF_REQUIRE(capacity == 0 || (ray_indices && t_starts && t_ends), "NULL sample arrays");
    SharedRuns sr;
    sr.n = 0;
    PERF_REQUIRE(!lattice_table || t0 == nullptr || lattice_mode == PERF_LATTICE_REPEATED, "per-ray lattice tables (t0 != NULL) exist for the repeated lattice only");
    if (lattice_mode == PERF_LATTICE_REPEATED && t0 == nullptr && !lattice_table) shared_runs_build(t0_base, step, chunk_words(max_steps) * 64 + 64, &sr);
    if (t0 == nullptr && lattice_table != nullptr && n_rays >= kSharedMinRays)        // (see march_write_shared_kernel)
        hipLaunchKernelGGL(march_write_shared_kernel, dim3((unsigned)div_up(n_rays, 256)), dim3(256), 0, as_stream(stream), n_rays, step,
                           (int32_t)chunk_words(max_steps), masks, counts, offsets, capacity, ray_indices, t_starts, t_ends, packed_info,
                           (const float*)nullptr, (const float*)nullptr, Aabb{}, (float*)nullptr, (uint8_t*)nullptr, 0, t0_base, (int)lattice_mode, lattice_table);
    else
    hipLaunchKernelGGL(march_write_kernel, dim3((unsigned)(n_rays / 4 >= 8192 ? div_up(n_rays, 16) : div_up(n_rays, 4))), dim3(256), 0, as_stream(stream), t0, n_rays,
                       step, (int32_t)chunk_words(max_steps), masks, counts, offsets, capacity, ray_indices, t_starts,
                       t_ends, packed_info, (const float*)nullptr, (const float*)nullptr, Aabb{}, (float*)nullptr, (uint8_t*)nullptr, 0, t0_scale, t0_base, (int)lattice_mode, sr, lattice_table);
    PERF_LAUNCH_CHECK("perf_occ_march_write");
    return PERF_OK;
}

extern "C" int perf_occ_march_write_points(const float* t0, float t0_scale, float t0_base, int64_t n_rays, float step, int32_t max_steps,
                                           int32_t lattice_mode, const float* lattice_table, const uint64_t* masks,
                                           const int32_t* counts, const int32_t* offsets, int64_t capacity, int64_t* ray_indices,
                                           float* t_starts, float* t_ends, int32_t* packed_info, const float* rays_o,
                                           const float* rays_d, const float* aabb6, float* x01, uint8_t* sel, int32_t rank_lo,
                                           void* stream) {
    PERF_REQUIRE(n_rays >= 0 && max_steps > 0 && capacity >= 0 && rank_lo >= 0, "perf_occ_march_write_points: bad arguments");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(masks && counts && offsets && packed_info && rays_o && rays_d && aabb6, "NULL pointer");
    PERF_REQUIRE(capacity == 0 || (ray_indices && t_starts && t_ends && x01), "NULL sample arrays");
    Aabb bb;
    for (int k = 0; k < 3; ++k) { bb.lo[k] = aabb6[k]; bb.hi[k] = aabb6[3 + k]; }
    SharedRuns sr;
    sr.n = 0;
    PERF_REQUIRE(!lattice_table || t0 == nullptr || lattice_mode == PERF_LATTICE_REPEATED, "per-ray lattice tables (t0 != NULL) exist for the repeated lattice only");
    if (lattice_mode == PERF_LATTICE_REPEATED && t0 == nullptr && !lattice_table) shared_runs_build(t0_base, step, chunk_words(max_steps) * 64 + 64, &sr);
    if (t0 == nullptr && lattice_table != nullptr && n_rays >= kSharedMinRays)        // (see march_write_shared_kernel)
        hipLaunchKernelGGL(march_write_shared_kernel, dim3((unsigned)div_up(n_rays, 256)), dim3(256), 0, as_stream(stream), n_rays, step,
                           (int32_t)chunk_words(max_steps), masks, counts, offsets, capacity, ray_indices, t_starts, t_ends, packed_info,
                           rays_o, rays_d, bb, x01, sel, rank_lo, t0_base, (int)lattice_mode, lattice_table);
    else
    hipLaunchKernelGGL(march_write_kernel, dim3((unsigned)(n_rays / 4 >= 8192 ? div_up(n_rays, 16) : div_up(n_rays, 4))), dim3(256), 0, as_stream(stream), t0, n_rays,
                       step, (int32_t)chunk_words(max_steps), masks, counts, offsets, capacity, ray_indices, t_starts,
                       t_ends, packed_info, rays_o, rays_d, bb, x01, sel, rank_lo, t0_scale, t0_base, (int)lattice_mode, sr, lattice_table);
    PERF_LAUNCH_CHECK("perf_occ_march_write_points");
    return PERF_OK;
}

// ---- per-ray tables of runs: one LANE per ray ------------------------------------------------------------------------------
namespace perf {
// One LANE per ray.  The walk as a loop WITHOUT long divergent bodies: every iteration makes the two real additions and the
// division whatever happens next (lanes cross their binades at different indices; with the three cases as branches a wave
// executed all three bodies in nearly every iteration: 31 us for 8,192 rays), then advances by one of the three cases.  Same
// arithmetic, same tables as lattice_runs_build.
__device__ __forceinline__ int lattice_runs_build_lane(float t0, float step, int k_need, int32_t* __restrict__ row) {
    int32_t* ks = row + 1;
    uint32_t* bs = reinterpret_cast<uint32_t*>(row + 1 + kMaxRuns);
    uint32_t* dd = reinterpret_cast<uint32_t*>(row + 1 + 2 * kMaxRuns);
    int n = 0;
    uint32_t tb = __float_as_uint(t0);
    ks[0] = 0; bs[0] = tb; dd[0] = 0u; n = 1;
    int K = 0;
    bool stuck = false;
    while (K < k_need && !stuck) {
        if (n + 2 > kMaxRuns) return -1;
        const uint32_t b1 = __float_as_uint(add_rn(__uint_as_float(tb), step));
        const uint32_t b2 = __float_as_uint(add_rn(__uint_as_float(b1), step));
        const bool c1 = (b1 >> 23) != (tb >> 23);            // the first addition entered a binade: one more real step first
        const bool c2 = (b2 >> 23) != (b1 >> 23);
        const uint32_t d = b2 - b1;
        const uint32_t top = (b2 & 0xff800000u) + 0x00800000u;
        const uint32_t j = div_u24(top - 1u - b2, d ? d : 1u);
        // run K+1 (always emitted): a single point (c1, c2) or the arithmetic run to the end of the binade
        ks[n] = K + 1; bs[n] = b1; dd[n] = (c1 || c2) ? 0u : d;
        ++n;
        if (!c1 && c2) { ks[n] = K + 2; bs[n] = b2; dd[n] = 0u; ++n; }
        stuck = !c1 && !c2 && d == 0u;                       // (step below half an ulp: the lattice stays at t_{K+1} for good)
        tb = c1 ? b1 : (c2 ? b2 : b2 + j * d);
        K += c1 ? 1 : (c2 ? 2 : 2 + (int)j);
    }
    return n;
}

__global__ __launch_bounds__(64) void lattice_runs_kernel(const float* __restrict__ t0s, float t0_scale, float t0_base, int64_t n_rays,
                                                          float step, int32_t k_need, int32_t* __restrict__ rows) {
    const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rays) return;
    int32_t* row = rows + r * (int64_t)kRunsStride;
    row[0] = lattice_runs_build_lane(lattice_origin(t0s, r, t0_scale, t0_base), step, k_need, row);
}
}  // namespace perf

extern "C" int64_t perf_occ_lattice_runs_len(int64_t n_rays) { return (n_rays > 0 ? n_rays : 0) * (int64_t)kRunsStride; }

extern "C" int perf_occ_lattice_runs(const float* t0, float t0_scale, float t0_base, int64_t n_rays, float step, int32_t max_steps,
                                     int32_t* runs, void* stream) {
    PERF_REQUIRE(n_rays >= 0 && max_steps > 0 && step > 0.f, "perf_occ_lattice_runs: bad arguments");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(t0 && runs, "NULL pointer");
    hipLaunchKernelGGL(perf::lattice_runs_kernel, dim3((unsigned)div_up(n_rays, 64)), dim3(64), 0, as_stream(stream), t0, t0_scale, t0_base, n_rays,
                       step, (int32_t)(chunk_words(max_steps) * 64 + 64), runs);
    PERF_LAUNCH_CHECK("perf_occ_lattice_runs");
    return PERF_OK;
}

// t_k, k = 0 .. n-1, of the lattice that starts at t0 -- for launches whose rays all share it (eval renders: no stratified jitter).
// The marching entry points take the table as `lattice_table`: a lattice point is then ONE load instead of a walk (repeated
// addition) -- a 512x1024 eval frame marches 0.5 M rays on the same 3,000 points.
namespace perf {
__global__ __launch_bounds__(256) void lattice_table_kernel(float t0, float step, int32_t n, int mode, float* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = lattice(t0, k, step, mode);
}
}  // namespace perf

extern "C" int64_t perf_occ_lattice_table_len(int32_t max_steps) { return (int64_t)chunk_words(max_steps) * 64 + 66; }

extern "C" int perf_occ_lattice_table(float t0, float step, int32_t max_steps, int32_t lattice_mode, float* table, void* stream) {
    PERF_REQUIRE(table && max_steps > 0 && step > 0.f, "perf_occ_lattice_table: bad arguments");
    PERF_REQUIRE(lattice_mode == PERF_LATTICE_SINGLE || lattice_mode == PERF_LATTICE_REPEATED, "bad lattice mode %d", (int)lattice_mode);
    const int32_t n = (int32_t)perf_occ_lattice_table_len(max_steps);
    hipLaunchKernelGGL(perf::lattice_table_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, as_stream(stream), t0, step, n,
                       (int)lattice_mode, table);
    PERF_LAUNCH_CHECK("perf_occ_lattice_table");
    return PERF_OK;
}
