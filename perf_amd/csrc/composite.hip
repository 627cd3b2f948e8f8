// Packed-ray compositing for gfx950: per-ray exclusive scans, visibility compaction, weights and
// segmented accumulation (nerfacc render_weight_from_density / accumulate_along_rays /
// render_visibility_from_density), the distortion loss (A.5) and their backward passes.
//
// One 64-lane wavefront owns one ray: samples of a ray are contiguous in the packed arrays
// (packed_info[r] = (start, count)), so every access is a coalesced 256-byte row, prefix sums are
// wave scans (Kogge-Stone over 64 lanes + a scalar carry between chunks) and the per-ray reductions
// need no atomics (the reference's index_add_ does).  The scan uses a fixed association order with
// unfused adds so that thresholds on it (early termination) are bit-exact against
// oracle/perf_oracle.py:packed_exclusive_sum_canonical.
#include "common.hpp"

namespace perf {

__device__ __forceinline__ float wave_incl_scan_f(float x, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float y = __shfl_up(x, off);
        if (lane >= off) x = add_rn(x, y);
    }
    return x;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// exclusive prefix within the ray: returns ex for this lane's sample, updates carry
__device__ __forceinline__ float chunk_excl(float v, int lane, float& carry) {
    const float inc = wave_incl_scan_f(v, lane);
    float prev = __shfl_up(inc, 1);
    if (lane == 0) prev = 0.f;
    const float ex = add_rn(carry, prev);
    carry = add_rn(carry, __shfl(inc, 63));
    return ex;
}

// ---- ray teams ------------------------------------------------------------------------------------------------------
// A wave serves FOUR rays.  When all four hold at most 16 samples each gets a quarter wave (16 lanes): a trained scene
// keeps a sample or two per ray, a whole wavefront per ray then idles 60 of its 64 lanes (eval frames: 5 x 10^5 rays).
// Otherwise the wave walks its four rays one after the other with all 64 lanes.  Both ways produce the same bits: the
// canonical scan is Kogge-Stone inside 64-sample chunks, and for <= 16 samples its offsets 16 and 32 -- like the butterfly
// steps 32 and 16 of the reductions -- only ever add exact zeros.
template <int W> struct Team { static constexpr int width = W; };

template <int W>
__device__ __forceinline__ float team_incl_scan_f(float x, int l) {
#pragma unroll
    for (int off = 1; off < W; off <<= 1) {
        const float y = __shfl_up(x, off, W);
        if (l >= off) x = add_rn(x, y);
    }
    return x;
}
template <int W>
__device__ __forceinline__ float team_sum(float x) {
#pragma unroll
    for (int off = W / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, W);
    return x;
}
template <int W>
__device__ __forceinline__ float team_chunk_excl(float v, int l, float& carry) {
    const float inc = team_incl_scan_f<W>(v, l);
    float prev = __shfl_up(inc, 1, W);
    if (l == 0) prev = 0.f;
    const float ex = add_rn(carry, prev);
    carry = add_rn(carry, __shfl(inc, W - 1, W));
    return ex;
}
// ballot restricted to the caller's team (bit k = lane k of the team)
template <int W>
__device__ __forceinline__ unsigned long long team_ballot(bool p) {
    const unsigned long long b = __ballot(p);
    if (W == 64) return b;
    return (b >> (((threadIdx.x & 63) / W) * W)) & ((1ull << W) - 1ull);
}

// Two launch shapes, told apart by the grid size (team_grid); a third one -- sixteen rays per wave, BY16 -- has its own instantiations:
//  * packed (n_rays / 4 waves; chosen when that still fills the GPU, i.e. eval batches): wave w serves rays 4w..4w+3 --
//    a quarter wave each when all four hold <= 16 samples, one after the other with all 64 lanes otherwise;
//  * one wave per ray (small batches: 8,192 training rays of 128 samples must not lose three quarters of their waves): the
//    waves of an aligned group of four rays take the same decision; in the quarter-wave case the group's first wave serves
//    all four rays and the other three retire at once.
// Calls body(Team<4, 16 or 64>, ray, lane-in-team).
constexpr int64_t kPackedMinWaves = 8192;       // 256 CUs x 32 waves

// BY16 (its own kernel instantiations, launched with n_rays / 16 waves from 524,288 rays on -- a 512 x 1024 frame as one batch): wave w serves
// rays 16w..16w+15, with 4-lane teams when all sixteen hold <= 4 samples (with n_rays / 4 waves three of every four waves of such a frame
// loaded sixteen counts and retired), else its four groups of four one after the other.  NOT a branch of the other shapes' code: with the
// three bodies instantiated a second time, or with the shapes folded into one loop, the kernels ran 12-40 % slower on rays of 128 samples
// in EVERY shape (tools/exp/team_shape_ab.py, three library builds: visibility_count 14.0 -> 19.5 us at 32,768 rays).
template <bool BY16 = false, typename CountFn, typename Body>
__device__ __forceinline__ void for_rays_of_wave(int64_t n_rays, CountFn count_of, Body body) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if constexpr (BY16) {
        const int64_t g16 = w * 16;
        if (g16 >= n_rays) return;
        const int64_t r16 = g16 + (lane >> 2);
        const int c16 = r16 < n_rays ? count_of(r16) : 0;
        if (__ballot(c16 > 4) == 0ull) {
            if (r16 < n_rays) body(Team<4>{}, r16, lane & 3);
            return;
        }
        for (int g4 = 0; g4 < 4; ++g4) {
            const int64_t rb = g16 + 4 * g4;
            if (rb >= n_rays) break;
            const int64_t rq = rb + (lane >> 4);
            const int cq = rq < n_rays ? count_of(rq) : 0;
            if (__ballot(cq > 16) == 0ull) {
                if (rq < n_rays) body(Team<16>{}, rq, lane & 15);
            } else {
                for (int q = 0; q < 4; ++q)
                    if (rb + q < n_rays) body(Team<64>{}, rb + q, lane);
            }
        }
        return;
    }
    const bool packed = (int64_t)gridDim.x * 4 < n_rays;
    const int64_t r0 = packed ? w * 4 : (w & ~(int64_t)3);
    if (r0 >= n_rays) return;
    // sixteen rays of at most 4 samples each (a trained scene renders with one or two kept samples per ray): the first wave
    // of their aligned group serves all sixteen with 4-lane teams, the other waves of the group retire.  Same bits again:
    // for <= 4 samples the Kogge-Stone offsets 4..32 and the butterfly steps 32..4 only ever add exact zeros.
    {
        const int64_t g16 = packed ? (w >> 2) * 16 : (w & ~(int64_t)15);
        const bool leader = packed ? (w & 3) == 0 : (w & 15) == 0;
        const int64_t r16 = g16 + (lane >> 2);
        const int c16 = r16 < n_rays ? count_of(r16) : 0;
        if (__ballot(c16 > 4) == 0ull) {
            if (leader && r16 < n_rays) body(Team<4>{}, r16, lane & 3);
            return;
        }
    }
    const int64_t rq = r0 + (lane >> 4);
    const int cq = rq < n_rays ? count_of(rq) : 0;
    if (__ballot(cq > 16) == 0ull) {
        if (!packed && (w & 3) != 0) return;
        if (rq < n_rays) body(Team<16>{}, rq, lane & 15);
    } else if (packed) {
        for (int q = 0; q < 4; ++q)
            if (r0 + q < n_rays) body(Team<64>{}, r0 + q, lane);
    } else if (w < n_rays) {
        body(Team<64>{}, w, lane);
    }
}
static inline dim3 team_grid(int64_t n_rays) {
    return dim3((unsigned)(n_rays / 4 >= kPackedMinWaves ? div_up(n_rays, 16) : div_up(n_rays, 4)));
}
// the BY16 instantiations take launches of this many rays and more (measured at 262,144 rays, tools/exp/team_shape_ab.py: rays of two
// samples gain -- visibility 13 -> 9 us, compositing 19 -> 13 --, rays of 128 samples lose 9-16 %: sixteen rays one after the other per
// wave leave fewer requests in flight; the eval render's batches stay with four rays per wave)
constexpr int64_t kBy16MinRays = 16 * 4 * kPackedMinWaves;          // 524,288
static inline bool team_by16(int64_t n_rays) { return n_rays >= kBy16MinRays; }
static inline dim3 team_grid16(int64_t n_rays) { return dim3((unsigned)div_up(n_rays, 64)); }

template <bool BY16>
__global__ __launch_bounds__(256) void visibility_count_kernel(const float* __restrict__ sig, const float* __restrict__ ts,
                                                               const float* __restrict__ te, const int32_t* __restrict__ packed,
                                                               int64_t n_rays, float thr, int32_t* __restrict__ new_counts,
                                                               float* __restrict__ exsum, const int32_t* __restrict__ march_counts,
                                                               int32_t head_k, int32_t* __restrict__ tail_counts) {
    for_rays_of_wave<BY16>(n_rays, [&](int64_t r) { return packed[2 * r + 1]; }, [&](auto team, int64_t r, int l) {
        constexpr int W = decltype(team)::width;
        const int64_t start = packed[2 * r];
        const int cnt = packed[2 * r + 1];
        float carry = 0.f;
        int kept = 0;
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int i = c0 + l;
            const bool valid = i < cnt;
            float sd = 0.f;
            if (valid) sd = mul_rn(sig[start + i], sub_rn(te[start + i], ts[start + i]));
            const float ex = team_chunk_excl<W>(sd, l, carry);
            if (valid && exsum) exsum[start + i] = ex;
            kept += __popcll(team_ballot<W>(valid && (ex <= thr)));
        }
        if (l == 0) {
            new_counts[r] = kept;
            if (tail_counts) {          // two-phase sampler: the tail of a ray whose whole head survived (perf_head_tail_counts)
                const int32_t c = march_counts[r], h = c < head_k ? c : head_k;
                tail_counts[r] = (kept == h && c > head_k) ? c - head_k : 0;
            }
        }
    });
}

// optional copy of the level-major features the density pass of the sampler produced (reused by the gradient pass: the
// reference evaluates the density field twice on the kept samples, nerf_renderer.py:145-148 and :166-168 -- same
// parameters, same positions, bit-identical features)
struct FeatCopy {
    const uint32_t* in_h; int64_t stride_h;       // head (or only) source
    const uint32_t* in_t; int64_t stride_t;       // tail source (two-phase sampler)
    uint32_t* out; int64_t stride_out;
    int32_t n_levels;
    int32_t* src_index_out;                       // instead of a copy: the source row of every kept sample (one-phase compaction)
};

__global__ __launch_bounds__(256) void compact_prefix_kernel(const int32_t* __restrict__ packed, const int32_t* __restrict__ new_counts,
                                                             const int32_t* __restrict__ new_offsets, int64_t n_rays,
                                                             const float* __restrict__ ts_in, const float* __restrict__ te_in,
                                                             const float* __restrict__ sig_in, int64_t* __restrict__ ri_out,
                                                             float* __restrict__ ts_out, float* __restrict__ te_out,
                                                             float* __restrict__ sig_out, int32_t* __restrict__ packed_out,
                                                             const float* __restrict__ x01_in, const uint8_t* __restrict__ sel_in,
                                                             float* __restrict__ x01_out, uint8_t* __restrict__ sel_out,
                                                             FeatCopy fc) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int64_t src = packed[2 * r];
    const int cnt = new_counts[r];
    const int64_t dst = new_offsets[r];
    if (lane == 0) { packed_out[2 * r] = (int32_t)dst; packed_out[2 * r + 1] = cnt; }
    for (int i = lane; i < cnt; i += 64) {
        ts_out[dst + i] = ts_in[src + i];
        te_out[dst + i] = te_in[src + i];
        ri_out[dst + i] = r;
        if (sig_out) sig_out[dst + i] = sig_in[src + i];
        if (sel_out) sel_out[dst + i] = sel_in[src + i];
    }
    if (x01_out)            // positions of the kept samples: 3*cnt contiguous floats
        for (int i = lane; i < 3 * cnt; i += 64) x01_out[3 * dst + i] = x01_in[3 * src + i];
    if (fc.out)             // level-major encoded features of the kept samples (one packed 2x16-bit dword per level)
        for (int l = 0; l < fc.n_levels; ++l)
            for (int i = lane; i < cnt; i += 64) fc.out[(int64_t)l * fc.stride_out + dst + i] = fc.in_h[(int64_t)l * fc.stride_h + src + i];
    if (fc.src_index_out)   // ... or where they are (perf_mlp_bwd reads them in place)
        for (int i = lane; i < cnt; i += 64) fc.src_index_out[dst + i] = (int32_t)(src + i);
}

template <bool BY16>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ sig, const float* __restrict__ rgb,
                                                            const float* __restrict__ ts, const float* __restrict__ te,
                                                            const int32_t* __restrict__ packed, int64_t n_rays,
                                                            float* __restrict__ weights, float* __restrict__ trans,
                                                            float* __restrict__ alphas, float* __restrict__ opacity,
                                                            float* __restrict__ distance, float* __restrict__ color,
                                                            float* __restrict__ distloss) {
    for_rays_of_wave<BY16>(n_rays, [&](int64_t r) { return packed[2 * r + 1]; }, [&](auto team, int64_t r, int l) {
        constexpr int TW = decltype(team)::width;
        const int64_t start = packed[2 * r];
        const int cnt = packed[2 * r + 1];
        float carry = 0.f;
        float a_op = 0.f, a_d = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f;
        float cW = 0.f, cWM = 0.f, a_dl = 0.f;          // distortion loss of the ray (same arithmetic as distloss_fwd_kernel)
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int i = c0 + l;
            const bool valid = i < cnt;
            float sd = 0.f, t0 = 0.f, t1 = 0.f;
            if (valid) { t0 = ts[start + i]; t1 = te[start + i]; sd = mul_rn(sig[start + i], sub_rn(t1, t0)); }
            const float ex = team_chunk_excl<TW>(sd, l, carry);
            float w_dl = 0.f;
            if (valid) {
                const float T = expf(-ex);
                const float al = 1.0f - expf(-sd);
                const float w = T * al;
                w_dl = w;
                if (weights) weights[start + i] = w;
                if (trans) trans[start + i] = T;
                if (alphas) alphas[start + i] = al;
                a_op += w;
                a_d += w * ((t0 + t1) * 0.5f);
                if (rgb) {
                    a_r += w * rgb[3 * (start + i)];
                    a_g += w * rgb[3 * (start + i) + 1];
                    a_b += w * rgb[3 * (start + i) + 2];
                }
            }
            if (distloss) {
                const float m = (t0 + t1) * 0.5f, d = t1 - t0;
                const float Wp = team_chunk_excl<TW>(w_dl, l, cW);
                const float WMp = team_chunk_excl<TW>(w_dl * m, l, cWM);
                if (valid) a_dl += d * w_dl * w_dl * (1.0f / 3.0f) + 2.0f * w_dl * (m * Wp - WMp);
            }
        }
        if (distloss) { a_dl = team_sum<TW>(a_dl); if (l == 0) distloss[r] = a_dl; }
        a_op = team_sum<TW>(a_op); a_d = team_sum<TW>(a_d);
        if (rgb) { a_r = team_sum<TW>(a_r); a_g = team_sum<TW>(a_g); a_b = team_sum<TW>(a_b); }
        if (l == 0) {
            if (opacity) opacity[r] = a_op;
            if (distance) distance[r] = a_d;
            if (rgb && color) { color[3 * r] = a_r; color[3 * r + 1] = a_g; color[3 * r + 2] = a_b; }
        }
    });
}

// d sigma_i = delta_i * ( (G_i T_i + gA_i) (1 - alpha_i) - sum_{j>i} (G_j w_j + gT_j T_j) ),
//   G = g_w + g_op + g_dist * t_mid,  gT = g_trans, gA = g_alphas (both optional)
// d rgb_i   = w_i * g_color[r]     (colour is accumulated with detached weights)
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ sig, const float* __restrict__ ts,
                                                            const float* __restrict__ te, const int32_t* __restrict__ packed,
                                                            int64_t n_rays, const float* __restrict__ weights,
                                                            const float* __restrict__ trans, const float* __restrict__ g_w,
                                                            const float* __restrict__ g_T, const float* __restrict__ g_al,
                                                            const float* __restrict__ g_op, const float* __restrict__ g_dist,
                                                            const float* __restrict__ g_col, float* __restrict__ d_sig,
                                                            float* __restrict__ d_rgb, const float* __restrict__ tot_w,
                                                            const float* __restrict__ tot_wm, float dl_scale,
                                                            const float* __restrict__ dl_scale_dev) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int64_t start = packed[2 * r];
    const int cnt = packed[2 * r + 1];
    if (cnt == 0) return;
    const float gop = g_op ? g_op[r] : 0.f, gd = g_dist ? g_dist[r] : 0.f;
    // distortion-loss gradient w.r.t. the weights formed here instead of being read from g_w (tot_w != NULL):
    // d loss / d w_i = scale ( 2/3 d_i w_i + 2 ( m_i (W_i - Wsuf_i) - (WM_i - WMsuf_i) ) ), prefixes = totals - suffixes
    const bool with_dl = tot_w != nullptr;
    const float totW = with_dl ? tot_w[r] : 0.f, totWM = with_dl ? tot_wm[r] : 0.f;
    if (with_dl && dl_scale_dev) dl_scale *= dl_scale_dev[0];
    float sufW = 0.f, sufWM = 0.f;                      // sums over all later chunks
    float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f;
    if (g_col && d_rgb) { gc0 = g_col[3 * r]; gc1 = g_col[3 * r + 1]; gc2 = g_col[3 * r + 2]; }
    float carry = 0.f;   // sum of G_j w_j over all later chunks
    const int n_chunks = (cnt + 63) / 64;
    for (int q = n_chunks - 1; q >= 0; --q) {
        const int i = q * 64 + lane;
        const bool valid = i < cnt;
        float w = 0.f, T = 0.f, t0 = 0.f, t1 = 0.f, s = 0.f, G = 0.f, gT = 0.f, gA = 0.f;
        if (valid) {
            w = weights[start + i]; T = trans[start + i]; t0 = ts[start + i]; t1 = te[start + i]; s = sig[start + i];
            G = gop + gd * ((t0 + t1) * 0.5f) + (g_w ? g_w[start + i] : 0.f);
            if (g_T) gT = g_T[start + i];
            if (g_al) gA = g_al[start + i];
        }
        if (with_dl) {
            const float m = (t0 + t1) * 0.5f, wm = w * m;
            float sw = w, swm = wm;                     // inclusive suffix sums within the chunk
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float y0 = __shfl_down(sw, off), y1 = __shfl_down(swm, off);
                if (lane + off < 64) { sw += y0; swm += y1; }
            }
            const float Wsuf = sufW + (sw - w), WMsuf = sufWM + (swm - wm);
            const float W = totW - Wsuf - w, WM = totWM - WMsuf - wm;
            if (valid) G += dl_scale * ((2.0f / 3.0f) * (t1 - t0) * w + 2.0f * (m * (W - Wsuf) - (WM - WMsuf)));
            sufW += __shfl(sw, 0); sufWM += __shfl(swm, 0);
        }
        const float qv = G * w + gT * T;
        // inclusive suffix sum within the chunk
        float suf = qv;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            float y = __shfl_down(suf, off);
            if (lane + off < 64) suf += y;
        }
        const float later = carry + (suf - qv);
        if (valid) {
            const float delta = t1 - t0;
            if (d_sig) d_sig[start + i] = delta * ((G * T + gA) * expf(-s * delta) - later);
            if (d_rgb) { d_rgb[3 * (start + i)] = w * gc0; d_rgb[3 * (start + i) + 1] = w * gc1; d_rgb[3 * (start + i) + 2] = w * gc2; }
        }
        carry += __shfl(suf, 0);
    }
}

// distortion loss, per ray:  sum_i ( d_i w_i^2 / 3 + 2 w_i (m_i W_i - WM_i) ),  W, WM exclusive prefixes
__global__ __launch_bounds__(256) void distloss_fwd_kernel(const float* __restrict__ w, const float* __restrict__ ts,
                                                           const float* __restrict__ te, const int32_t* __restrict__ packed,
                                                           int64_t n_rays, float* __restrict__ loss) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int64_t start = packed[2 * r];
    const int cnt = packed[2 * r + 1];
    float cW = 0.f, cWM = 0.f, acc = 0.f;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < cnt;
        float wi = 0.f, m = 0.f, d = 0.f;
        if (valid) { wi = w[start + i]; const float a = ts[start + i], b = te[start + i]; m = (a + b) * 0.5f; d = b - a; }
        const float W = chunk_excl(wi, lane, cW);
        const float WM = chunk_excl(wi * m, lane, cWM);
        if (valid) acc += d * wi * wi * (1.0f / 3.0f) + 2.0f * wi * (m * W - WM);
    }
    acc = wave_sum(acc);
    if (lane == 0) loss[r] = acc;
}

// d loss / d w_i = scale * ( 2/3 d_i w_i + 2 ( m_i (W_i - Wsuf_i) - (WM_i - WMsuf_i) ) )
__global__ __launch_bounds__(256) void distloss_bwd_kernel(const float* __restrict__ w, const float* __restrict__ ts,
                                                           const float* __restrict__ te, const int32_t* __restrict__ packed,
                                                           int64_t n_rays, float scale, const float* __restrict__ scale_dev,
                                                           float* __restrict__ g_w) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    if (scale_dev) scale *= scale_dev[0];
    const int64_t start = packed[2 * r];
    const int cnt = packed[2 * r + 1];
    float totW = 0.f, totWM = 0.f;
    for (int i = lane; i < cnt; i += 64) {
        const float wi = w[start + i];
        totW += wi; totWM += wi * ((ts[start + i] + te[start + i]) * 0.5f);
    }
    totW = wave_sum(totW); totWM = wave_sum(totWM);
    float cW = 0.f, cWM = 0.f;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < cnt;
        float wi = 0.f, m = 0.f, d = 0.f;
        if (valid) { wi = w[start + i]; const float a = ts[start + i], b = te[start + i]; m = (a + b) * 0.5f; d = b - a; }
        const float W = chunk_excl(wi, lane, cW);
        const float WM = chunk_excl(wi * m, lane, cWM);
        if (valid) {
            const float Wsuf = totW - W - wi, WMsuf = totWM - WM - wi * m;
            g_w[start + i] = scale * ((2.0f / 3.0f) * d * wi + 2.0f * (m * (W - Wsuf) - (WM - WMsuf)));
        }
    }
}

// accumulate_along_rays: out[r, c] = sum_i w_i * v[i, c]  (v == NULL: C = 1, out[r] = sum_i w_i)
__global__ __launch_bounds__(256) void accumulate_kernel(const float* __restrict__ w, const float* __restrict__ v,
                                                         const int32_t* __restrict__ packed, int64_t n_rays, int C,
                                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int64_t start = packed[2 * r];
    const int cnt = packed[2 * r + 1];
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
        for (int i = lane; i < cnt; i += 64) acc += w[start + i] * (v ? v[(start + i) * C + c] : 1.0f);
        acc = wave_sum(acc);
        if (lane == 0) out[r * C + c] = acc;
    }
}

// packed_info from sorted ray_indices: start[r] = lower_bound(ray_indices, r)
__global__ __launch_bounds__(256) void pack_info_kernel(const int64_t* __restrict__ ri, int64_t n, int64_t n_rays,
                                                        int32_t* __restrict__ packed) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rays) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ri[mid] < r) lo = mid + 1; else hi = mid; }
    const int64_t first = lo;
    hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ri[mid] <= r) lo = mid + 1; else hi = mid; }
    packed[2 * r] = (int32_t)first; packed[2 * r + 1] = (int32_t)(lo - first);
}

static inline dim3 ray_grid(int64_t n_rays) { return dim3((unsigned)div_up(n_rays, 4)); }

}  // namespace perf

using namespace perf;

extern "C" int perf_visibility_count(const float* sigmas, const float* t_starts, const float* t_ends,
                                     const int32_t* packed_info, int64_t n_rays, float thr, int32_t* new_counts,
                                     float* exsum, const int32_t* march_counts, int32_t head_samples, int32_t* tail_counts,
                                     void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info && new_counts, "NULL pointer");
    PERF_REQUIRE(!tail_counts || (march_counts && head_samples >= 1), "perf_visibility_count: tail counts need the march counts and the head size");
    if (team_by16(n_rays)) hipLaunchKernelGGL(visibility_count_kernel<true>, team_grid16(n_rays), dim3(256), 0, as_stream(stream), sigmas, t_starts, t_ends,
                       packed_info, n_rays, thr, new_counts, exsum, march_counts, head_samples, tail_counts);
    else hipLaunchKernelGGL(visibility_count_kernel<false>, team_grid(n_rays), dim3(256), 0, as_stream(stream), sigmas, t_starts, t_ends,
                       packed_info, n_rays, thr, new_counts, exsum, march_counts, head_samples, tail_counts);
    PERF_LAUNCH_CHECK("perf_visibility_count");
    return PERF_OK;
}

extern "C" int perf_compact_prefix(const int32_t* packed_info, const int32_t* new_counts, const int32_t* new_offsets,
                                   int64_t n_rays, const float* ts_in, const float* te_in, const float* sig_in,
                                   int64_t* ray_indices_out, float* ts_out, float* te_out, float* sig_out,
                                   int32_t* packed_out, const float* x01_in, const uint8_t* sel_in, float* x01_out,
                                   uint8_t* sel_out, const void* feat_in, int64_t feat_stride_in, void* feat_out,
                                   int64_t feat_stride_out, int32_t n_levels, int32_t* src_index_out, void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info && new_counts && new_offsets && packed_out, "NULL pointer");
    PERF_REQUIRE((sig_in == nullptr) == (sig_out == nullptr), "sig_in/sig_out must both be given or both be NULL");
    PERF_REQUIRE((x01_in == nullptr) == (x01_out == nullptr) && (sel_in == nullptr) == (sel_out == nullptr),
                 "x01/sel in and out must both be given or both be NULL");
    PERF_REQUIRE((feat_in == nullptr) == (feat_out == nullptr), "feat in and out must both be given or both be NULL");
    FeatCopy fc{(const uint32_t*)feat_in, feat_stride_in, nullptr, 0, (uint32_t*)feat_out, feat_stride_out, n_levels, src_index_out};
    hipLaunchKernelGGL(compact_prefix_kernel, ray_grid(n_rays), dim3(256), 0, as_stream(stream), packed_info, new_counts,
                       new_offsets, n_rays, ts_in, te_in, sig_in, ray_indices_out, ts_out, te_out, sig_out, packed_out, x01_in, sel_in,
                       x01_out, sel_out, fc);
    PERF_LAUNCH_CHECK("perf_compact_prefix");
    return PERF_OK;
}

extern "C" int perf_composite_fwd(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                                  const int32_t* packed_info, int64_t n_rays, float* weights, float* trans, float* alphas,
                                  float* opacity, float* distance, float* color, void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info, "NULL pointer");
    if (team_by16(n_rays)) hipLaunchKernelGGL(composite_fwd_kernel<true>, team_grid16(n_rays), dim3(256), 0, as_stream(stream), sigmas, rgbs, t_starts,
                       t_ends, packed_info, n_rays, weights, trans, alphas, opacity, distance, color, (float*)nullptr);
    else hipLaunchKernelGGL(composite_fwd_kernel<false>, team_grid(n_rays), dim3(256), 0, as_stream(stream), sigmas, rgbs, t_starts,
                       t_ends, packed_info, n_rays, weights, trans, alphas, opacity, distance, color, (float*)nullptr);
    PERF_LAUNCH_CHECK("perf_composite_fwd");
    return PERF_OK;
}

extern "C" int perf_composite_distloss_fwd(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                                           const int32_t* packed_info, int64_t n_rays, float* weights, float* trans,
                                           float* opacity, float* distance, float* color, float* distloss_per_ray, void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info && distloss_per_ray, "NULL pointer");
    if (team_by16(n_rays)) hipLaunchKernelGGL(composite_fwd_kernel<true>, team_grid16(n_rays), dim3(256), 0, as_stream(stream), sigmas, rgbs, t_starts,
                       t_ends, packed_info, n_rays, weights, trans, (float*)nullptr, opacity, distance, color, distloss_per_ray);
    else hipLaunchKernelGGL(composite_fwd_kernel<false>, team_grid(n_rays), dim3(256), 0, as_stream(stream), sigmas, rgbs, t_starts,
                       t_ends, packed_info, n_rays, weights, trans, (float*)nullptr, opacity, distance, color, distloss_per_ray);
    PERF_LAUNCH_CHECK("perf_composite_distloss_fwd");
    return PERF_OK;
}

extern "C" int perf_composite_bwd(const float* sigmas, const float* t_starts, const float* t_ends,
                                  const int32_t* packed_info, int64_t n_rays, const float* weights, const float* trans,
                                  const float* g_weights, const float* g_trans, const float* g_alphas,
                                  const float* g_opacity, const float* g_distance,
                                  const float* g_color, float* d_sigmas, float* d_rgbs, void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info && weights && trans, "NULL pointer");
    hipLaunchKernelGGL(composite_bwd_kernel, ray_grid(n_rays), dim3(256), 0, as_stream(stream), sigmas, t_starts, t_ends,
                       packed_info, n_rays, weights, trans, g_weights, g_trans, g_alphas, g_opacity, g_distance, g_color, d_sigmas, d_rgbs,
                       (const float*)nullptr, (const float*)nullptr, 0.0f, (const float*)nullptr);
    PERF_LAUNCH_CHECK("perf_composite_bwd");
    return PERF_OK;
}

extern "C" int perf_composite_distloss_bwd(const float* sigmas, const float* t_starts, const float* t_ends,
                                           const int32_t* packed_info, int64_t n_rays, const float* weights, const float* trans,
                                           const float* opacity, const float* distance, const float* g_opacity,
                                           const float* g_distance, float distloss_scale, const float* distloss_scale_dev,
                                           float* d_sigmas, void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info && weights && trans && opacity && distance && d_sigmas, "NULL pointer");
    hipLaunchKernelGGL(composite_bwd_kernel, ray_grid(n_rays), dim3(256), 0, as_stream(stream), sigmas, t_starts, t_ends,
                       packed_info, n_rays, weights, trans, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       g_opacity, g_distance, (const float*)nullptr, d_sigmas, (float*)nullptr, opacity, distance, distloss_scale,
                       distloss_scale_dev);
    PERF_LAUNCH_CHECK("perf_composite_distloss_bwd");
    return PERF_OK;
}

extern "C" int perf_distloss_fwd(const float* w, const float* t_starts, const float* t_ends, const int32_t* packed_info,
                                 int64_t n_rays, float* loss_per_ray, void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info && loss_per_ray, "NULL pointer");
    hipLaunchKernelGGL(distloss_fwd_kernel, ray_grid(n_rays), dim3(256), 0, as_stream(stream), w, t_starts, t_ends,
                       packed_info, n_rays, loss_per_ray);
    PERF_LAUNCH_CHECK("perf_distloss_fwd");
    return PERF_OK;
}

extern "C" int perf_distloss_bwd(const float* w, const float* t_starts, const float* t_ends, const int32_t* packed_info,
                                 int64_t n_rays, float scale, const float* scale_dev, float* g_w, void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info && g_w, "NULL pointer");
    hipLaunchKernelGGL(distloss_bwd_kernel, ray_grid(n_rays), dim3(256), 0, as_stream(stream), w, t_starts, t_ends,
                       packed_info, n_rays, scale, scale_dev, g_w);
    PERF_LAUNCH_CHECK("perf_distloss_bwd");
    return PERF_OK;
}

extern "C" int perf_accumulate_fwd(const float* weights, const float* values, const int32_t* packed_info, int64_t n_rays,
                                   int32_t n_channels, float* out, void* stream) {
    PERF_REQUIRE(n_rays >= 0 && n_channels >= 1, "bad arguments");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info && out, "NULL pointer");
    hipLaunchKernelGGL(accumulate_kernel, ray_grid(n_rays), dim3(256), 0, as_stream(stream), weights, values, packed_info,
                       n_rays, (int)n_channels, out);
    PERF_LAUNCH_CHECK("perf_accumulate_fwd");
    return PERF_OK;
}

extern "C" int perf_pack_info(const int64_t* ray_indices, int64_t n, int64_t n_rays, int32_t* packed_info, void* stream) {
    PERF_REQUIRE(n >= 0 && n_rays >= 0, "bad arguments");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_info && (n == 0 || ray_indices), "NULL pointer");
    hipLaunchKernelGGL(pack_info_kernel, dim3((unsigned)div_up(n_rays, 256)), dim3(256), 0, as_stream(stream), ray_indices, n,
                       n_rays, packed_info);
    PERF_LAUNCH_CHECK("perf_pack_info");
    return PERF_OK;
}

// ---- two-phase early termination -----------------------------------------------------------------------------------
// render_visibility_from_density keeps, per ray, the leading samples whose exclusive sum of sigma*delta stays below
// -ln(eps): a PREFIX, because the sum never decreases.  Samples behind the first one that exceeds the threshold are dropped
// whatever their density is -- so their density need not be evaluated.  The sampler therefore evaluates the first K samples
// of every ray ("head"), decides the rays that terminate inside their head (or have no more samples), and evaluates the
// rest ("tail") only for the rays that are still alive.  In a trained scene a ray is opaque after a sample or two: the
// density pass shrinks from ~40 to K samples per ray.  Results are bit-identical to the one-phase path: the canonical scan
// value of sample i only involves samples <= i.
namespace perf {

// mode 0: out = min(counts, K)                                             (head counts)
// mode 1: out = (kept_head == min(counts, K) && counts > K) ? counts - K : 0     (tail counts of the rays still alive)
__global__ __launch_bounds__(256) void head_tail_counts_kernel(const int32_t* __restrict__ counts, int64_t n_rays, int32_t K,
                                                               const int32_t* __restrict__ kept_head, int32_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rays) return;
    const int32_t c = counts[r], h = c < K ? c : K;
    out[r] = kept_head ? ((kept_head[r] == h && c > K) ? c - K : 0) : h;
}

// sample i of ray r lives in the head arrays for i < head count, in the tail arrays behind that
struct TwoSource {
    const float* sig_h; const float* ts_h; const float* te_h; const int32_t* packed_h;
    const float* sig_t; const float* ts_t; const float* te_t; const int32_t* packed_t;
};

template <bool BY16>
__global__ __launch_bounds__(256) void visibility_count2_kernel(TwoSource src, int64_t n_rays, float thr, int32_t* __restrict__ new_counts) {
    for_rays_of_wave<BY16>(n_rays, [&](int64_t r) { return src.packed_h[2 * r + 1] + src.packed_t[2 * r + 1]; }, [&](auto team, int64_t r, int l) {
        constexpr int TW = decltype(team)::width;
        const int64_t sh = src.packed_h[2 * r], st = src.packed_t[2 * r];
        const int ch = src.packed_h[2 * r + 1], ct = src.packed_t[2 * r + 1];
        const int cnt = ch + ct;
        float carry = 0.f;
        int kept = 0;
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int i = c0 + l;
            const bool valid = i < cnt;
            float sd = 0.f;
            if (valid) {
                const bool head = i < ch;
                const int64_t j = head ? sh + i : st + (i - ch);
                const float sg = head ? src.sig_h[j] : src.sig_t[j];
                const float a = head ? src.ts_h[j] : src.ts_t[j], b = head ? src.te_h[j] : src.te_t[j];
                sd = mul_rn(sg, sub_rn(b, a));
            }
            const float ex = team_chunk_excl<TW>(sd, l, carry);
            const unsigned long long ok = team_ballot<TW>(valid && (ex <= thr));
            kept += __popcll(ok);
            if (ok != team_ballot<TW>(valid)) break;        // the prefix ended in this chunk (uniform over the team)
        }
        if (l == 0) new_counts[r] = kept;
    });
}

template <bool BY16>
__global__ __launch_bounds__(256) void compact_prefix2_kernel(TwoSource src, const float* __restrict__ x01_h, const uint8_t* __restrict__ sel_h,
                                                              const float* __restrict__ x01_t, const uint8_t* __restrict__ sel_t,
                                                              const int32_t* __restrict__ new_counts, const int32_t* __restrict__ new_offsets,
                                                              int64_t n_rays, int64_t capacity, int64_t* __restrict__ ri_out,
                                                              float* __restrict__ ts_out, float* __restrict__ te_out, float* __restrict__ sig_out,
                                                              float* __restrict__ x01_out, uint8_t* __restrict__ sel_out,
                                                              int32_t* __restrict__ packed_out, FeatCopy fc) {
    for_rays_of_wave<BY16>(n_rays, [&](int64_t r) { return new_counts[r]; }, [&](auto team, int64_t r, int l) {
        constexpr int TW = decltype(team)::width;
        const int64_t sh = src.packed_h[2 * r], st = src.packed_t[2 * r];
        const int ch = src.packed_h[2 * r + 1];
        int cnt = new_counts[r];
        const int64_t dst = new_offsets[r];
        if (dst + cnt > capacity) cnt = (int)(capacity > dst ? capacity - dst : 0);       // truncated batch
        if (l == 0) { packed_out[2 * r] = (int32_t)dst; packed_out[2 * r + 1] = cnt; }
        for (int i = l; i < cnt; i += TW) {
            const bool head = i < ch;
            const int64_t j = head ? sh + i : st + (i - ch);
            ts_out[dst + i] = head ? src.ts_h[j] : src.ts_t[j];
            te_out[dst + i] = head ? src.te_h[j] : src.te_t[j];
            ri_out[dst + i] = r;
            if (sig_out) sig_out[dst + i] = head ? src.sig_h[j] : src.sig_t[j];
            if (sel_out) sel_out[dst + i] = head ? sel_h[j] : sel_t[j];
            if (x01_out) {
                const float* p = head ? x01_h + 3 * j : x01_t + 3 * j;
                x01_out[3 * (dst + i)] = p[0]; x01_out[3 * (dst + i) + 1] = p[1]; x01_out[3 * (dst + i) + 2] = p[2];
            }
        }
        if (fc.out)
            for (int lv = 0; lv < fc.n_levels; ++lv)
                for (int i = l; i < cnt; i += TW)
                    fc.out[(int64_t)lv * fc.stride_out + dst + i] = (i < ch) ? fc.in_h[(int64_t)lv * fc.stride_h + sh + i]
                                                                             : fc.in_t[(int64_t)lv * fc.stride_t + st + (i - ch)];
    });
}
}  // namespace perf

extern "C" int perf_head_tail_counts(const int32_t* counts, int64_t n_rays, int32_t head_samples, const int32_t* kept_head,
                                     int32_t* out_counts, void* stream) {
    PERF_REQUIRE(n_rays >= 0 && head_samples >= 1, "perf_head_tail_counts: bad arguments");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(counts && out_counts, "NULL pointer");
    hipLaunchKernelGGL(perf::head_tail_counts_kernel, dim3((unsigned)perf::div_up(n_rays, 256)), dim3(256), 0, perf::as_stream(stream),
                       counts, n_rays, head_samples, kept_head, out_counts);
    PERF_LAUNCH_CHECK("perf_head_tail_counts");
    return PERF_OK;
}

extern "C" int perf_visibility_count2(const float* sig_h, const float* ts_h, const float* te_h, const int32_t* packed_h,
                                      const float* sig_t, const float* ts_t, const float* te_t, const int32_t* packed_t,
                                      int64_t n_rays, float thr, int32_t* new_counts, void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_h && packed_t && new_counts, "NULL pointer");
    perf::TwoSource src{sig_h, ts_h, te_h, packed_h, sig_t, ts_t, te_t, packed_t};
    if (perf::team_by16(n_rays)) hipLaunchKernelGGL(perf::visibility_count2_kernel<true>, perf::team_grid16(n_rays), dim3(256), 0, perf::as_stream(stream), src, n_rays, thr, new_counts);
    else hipLaunchKernelGGL(perf::visibility_count2_kernel<false>, perf::team_grid(n_rays), dim3(256), 0, perf::as_stream(stream), src, n_rays, thr, new_counts);
    PERF_LAUNCH_CHECK("perf_visibility_count2");
    return PERF_OK;
}

extern "C" int perf_compact_prefix2(const float* sig_h, const float* ts_h, const float* te_h, const int32_t* packed_h,
                                    const float* x01_h, const uint8_t* sel_h, const float* sig_t, const float* ts_t,
                                    const float* te_t, const int32_t* packed_t, const float* x01_t, const uint8_t* sel_t,
                                    const int32_t* new_counts, const int32_t* new_offsets, int64_t n_rays, int64_t capacity,
                                    int64_t* ray_indices_out, float* ts_out, float* te_out, float* sig_out, float* x01_out,
                                    uint8_t* sel_out, int32_t* packed_out, const void* feat_h, int64_t feat_stride_h,
                                    const void* feat_t, int64_t feat_stride_t, void* feat_out, int64_t feat_stride_out,
                                    int32_t n_levels, void* stream) {
    PERF_REQUIRE(n_rays >= 0 && capacity >= 0, "perf_compact_prefix2: bad arguments");
    PERF_REQUIRE(!feat_out || (feat_h && feat_t), "perf_compact_prefix2: feat_out needs both feature sources");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(packed_h && packed_t && new_counts && new_offsets && packed_out, "NULL pointer");
    PERF_REQUIRE(capacity == 0 || (ray_indices_out && ts_out && te_out), "NULL sample arrays");
    perf::TwoSource src{sig_h, ts_h, te_h, packed_h, sig_t, ts_t, te_t, packed_t};
    if (perf::team_by16(n_rays)) hipLaunchKernelGGL(perf::compact_prefix2_kernel<true>, perf::team_grid16(n_rays), dim3(256), 0, perf::as_stream(stream), src, x01_h, sel_h,
                       x01_t, sel_t, new_counts, new_offsets, n_rays, capacity, ray_indices_out, ts_out, te_out, sig_out, x01_out,
                       sel_out, packed_out,
                       perf::FeatCopy{(const uint32_t*)feat_h, feat_stride_h, (const uint32_t*)feat_t, feat_stride_t, (uint32_t*)feat_out,
                                      feat_stride_out, n_levels});
    else hipLaunchKernelGGL(perf::compact_prefix2_kernel<false>, perf::team_grid(n_rays), dim3(256), 0, perf::as_stream(stream), src, x01_h, sel_h,
                       x01_t, sel_t, new_counts, new_offsets, n_rays, capacity, ray_indices_out, ts_out, te_out, sig_out, x01_out,
                       sel_out, packed_out,
                       perf::FeatCopy{(const uint32_t*)feat_h, feat_stride_h, (const uint32_t*)feat_t, feat_stride_t, (uint32_t*)feat_out,
                                      feat_stride_out, n_levels});
    PERF_LAUNCH_CHECK("perf_compact_prefix2");
    return PERF_OK;
}

// ---- eval tail of NeRFOCCRenderer.render (nerf_renderer.py:195-197): distance += 5 (1 - opacity), rgb += 0.5 (1 - opacity).
// A batch without any sample returns before that tail in the reference (:156-162: zeros, is_valid False); with device-side
// counts the same decision is taken here from *n_dev.
namespace perf {
__global__ __launch_bounds__(256) void render_finish_eval_kernel(const float* __restrict__ opacity, float* __restrict__ distance,
                                                                 float* __restrict__ color, int64_t n_rays,
                                                                 const int64_t* __restrict__ n_dev) {
    if (n_dev && n_dev[0] <= 0) return;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rays) return;
    const float rest = 1.0f - opacity[r];
    if (distance) distance[r] = distance[r] + 5.0f * rest;
    if (color) {
        const float c = 0.5f * rest;
        color[3 * r] += c; color[3 * r + 1] += c; color[3 * r + 2] += c;
    }
}
}  // namespace perf

extern "C" int perf_render_finish_eval(const float* opacity, float* distance, float* color, int64_t n_rays,
                                       const int64_t* n_dev, void* stream) {
    PERF_REQUIRE(n_rays >= 0, "n_rays < 0");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(opacity, "NULL pointer");
    hipLaunchKernelGGL(perf::render_finish_eval_kernel, dim3((unsigned)perf::div_up(n_rays, 256)), dim3(256), 0,
                       perf::as_stream(stream), opacity, distance, color, n_rays, n_dev);
    PERF_LAUNCH_CHECK("perf_render_finish_eval");
    return PERF_OK;
}

// ---- hierarchical (inverse-CDF) resampling: nerfacc importance_sampling / PropNetEstimator.sampling, SURVEY.md A.6.
// Dead in the reference (nerf_renderer.py:60-73 raises NameError before use), so the semantics are this repo's own
// restatement (oracle/perf_oracle.py:pdf_resample):  edges u_j = (j + tau_r) / (n_out + 1), j = 0..n_out, tau_r = 0.5
// or the per-ray stratified draw; t_j = s_k + (u_j - cdf_k) / (cdf_{k+1} - cdf_k) * (s_{k+1} - s_k) for the interval
// k with cdf_k <= u_j < cdf_{k+1}.  One thread per output edge, binary search in the ray's (L1-resident) CDF row.
namespace perf {
__global__ __launch_bounds__(256) void pdf_resample_kernel(const float* __restrict__ s_in, const float* __restrict__ cdf,
                                                           const float* __restrict__ tau, int64_t n_rays, int n_in,
                                                           int n_out, float* __restrict__ s_out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = n_rays * (int64_t)(n_out + 1);
    if (t >= total) return;
    const int64_t r = t / (n_out + 1);
    const int j = (int)(t % (n_out + 1));
    const float* c = cdf + r * (int64_t)(n_in + 1);
    const float* s = s_in + r * (int64_t)(n_in + 1);
    const float u = __fdiv_rn(add_rn((float)j, tau ? tau[r] : 0.5f), (float)(n_out + 1));
    int lo = 0, hi = n_in;                 // largest k in [0, n_in-1] with c[k] <= u
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c[mid] <= u) lo = mid; else hi = mid; }
    const float c0 = c[lo], c1 = c[lo + 1], s0 = s[lo], s1 = s[lo + 1];
    const float den = sub_rn(c1, c0);
    float v = s0;
    if (den > 0.f) v = add_rn(s0, mul_rn(__fdiv_rn(sub_rn(u, c0), den), sub_rn(s1, s0)));
    s_out[t] = fminf(fmaxf(v, s0), s1);
}
}  // namespace perf

extern "C" int perf_pdf_resample(const float* s_in, const float* cdf, const float* tau, int64_t n_rays, int32_t n_in,
                                 int32_t n_out, float* s_out, void* stream) {
    PERF_REQUIRE(n_rays >= 0 && n_in >= 1 && n_out >= 1, "perf_pdf_resample: bad arguments");
    if (n_rays == 0) return PERF_OK;
    PERF_REQUIRE(s_in && cdf && s_out, "NULL pointer");
    const int64_t total = n_rays * (int64_t)(n_out + 1);
    hipLaunchKernelGGL(perf::pdf_resample_kernel, dim3((unsigned)perf::div_up(total, 256)), dim3(256), 0, perf::as_stream(stream),
                       s_in, cdf, tau, n_rays, (int)n_in, (int)n_out, s_out);
    PERF_LAUNCH_CHECK("perf_pdf_resample");
    return PERF_OK;
}

// ---- training-step loss heads (modules/scene/nerf.py:208-252 geometry, :281-293 colour), fused ---------------------
// One single-workgroup kernel replaces ~25 tiny elementwise/reduction launches of the autograd formulation: it applies
// the training-time noise/background terms of nerf_renderer.py:185-194, the smooth-L1 losses, the loss scale and
// produces the per-ray gradients the compositing backward consumes plus the scalars the distortion-loss backward needs.
namespace perf {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// smooth_l1(x, beta): 0.5 x^2 / beta for |x| < beta else |x| - 0.5 beta;  derivative x/beta or sign(x)
__device__ __forceinline__ float sl1(float x, float beta) { const float a = fabsf(x); return a < beta ? 0.5f * x * x / beta : a - 0.5f * beta; }
__device__ __forceinline__ float sl1_grad(float x, float beta) { return fabsf(x) < beta ? x / beta : (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f)); }

// Loss heads: per-ray work is spread over kLossBlocks workgroups that leave their partial sums behind the results in
// `scalars` (PERF_LOSS_SCALARS floats), a one-workgroup kernel then folds them in a fixed order (deterministic).
constexpr int kLossBlocks = 64;

// scalars out: [0] depth loss (mean over the global batch), [1] distortion loss, [2] scale for perf_distloss_bwd
__global__ __launch_bounds__(256) void geo_loss_kernel(const float* __restrict__ opacity, const float* __restrict__ distance,
                                                       const float* __restrict__ gt, const float* __restrict__ noise,
                                                       const float* __restrict__ dl_per_ray, const int32_t* __restrict__ packed,
                                                       int64_t n_rays, float inv_bs, float depth_w, float loss_scale,
                                                       float* __restrict__ g_op, float* __restrict__ g_dist,
                                                       float* __restrict__ scalars, float dist_w, const float* __restrict__ ratio_dev,
                                                       int data_parallel, int* __restrict__ ticket) {
    __shared__ float red[4];
    __shared__ int last_s;
    if (threadIdx.x == 0) last_s = -1;
    __syncthreads();
    float dsum = 0.f, lsum = 0.f;
    int last = -1;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * 256) {
        const float op = opacity[r];
        const float nz = noise ? (noise[r] * 2.0f - 1.0f) : 0.0f;
        const float pre = distance[r] + nz * (1.0f - op);
        const float d = fmaxf(pre, 0.0f);
        const float diff = d - gt[r];
        dsum += sl1(diff, 1e-2f);
        const float gd = (pre > 0.0f) ? sl1_grad(diff, 1e-2f) * inv_bs * depth_w * loss_scale : 0.0f;
        g_dist[r] = gd;
        g_op[r] = -nz * gd;
        lsum += dl_per_ray[r];
        if (packed[2 * r + 1] > 0) last = (int)r;
    }
    atomicMax(&last_s, last);
    const float depth = block_sum_256(dsum, red);
    const float distl = block_sum_256(lsum, red);
    // ---- the partial sums go out with returning device-scope atomics (performed at the memory side before the ticket is
    //      taken: no agent-scope fence, which would write back the XCD's L2); the last workgroup to take a ticket folds them
    //      in a fixed order (deterministic) and leaves *ticket at 0 for the next call
    __shared__ int is_last;
    if (threadIdx.x == 0) {
        float* part = scalars + 4 + 3 * blockIdx.x;
        float sink = atomicExch(&part[0], depth) + atomicExch(&part[1], distl) + atomicExch(&part[2], (float)last_s);  // ray indices < 2^24: exact
        asm volatile("" : : "v"(sink));
        is_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!is_last) return;
    if (threadIdx.x < 64) {
        const int n_blocks = (int)gridDim.x;
        float ds = 0.f, ls = 0.f, lastf = -1.f;
        if ((int)threadIdx.x < n_blocks) {
            float* part = scalars + 4 + 3 * threadIdx.x;
            ds = __hip_atomic_load(&part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ls = __hip_atomic_load(&part[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lastf = __hip_atomic_load(&part[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {       // fixed butterfly: deterministic
            ds += __shfl_xor(ds, off); ls += __shfl_xor(ls, off); lastf = fmaxf(lastf, __shfl_xor(lastf, off));
        }
        if (threadIdx.x == 0) {
            // flatten_eff_distloss divides by ray_id.max()+1 (the last ray that has samples); data-parallel runs (the local
            // batch is a slice of the global one) normalise by the global batch instead
            const float inv_n = data_parallel ? inv_bs : 1.0f / (lastf + 1.0f > 0.f ? lastf + 1.0f : 1.0f);
            const float ratio = ratio_dev ? ratio_dev[0] : 1.0f;
            scalars[0] = ds * inv_bs;
            scalars[1] = ls * inv_n;
            scalars[2] = inv_n * dist_w * ratio * loss_scale;
            *ticket = 0;
        }
    }
}

// colour head: colors' = col + bg * (1 - opacity); loss = mean smooth_l1(colors' - gt, 0.05) over bs*3; g_col = d loss / d col
__global__ __launch_bounds__(256) void app_loss_kernel(const float* __restrict__ opacity, const float* __restrict__ color,
                                                       const float* __restrict__ bg, const float* __restrict__ gt,
                                                       int64_t n_rays, float inv_n, float color_w, float loss_scale,
                                                       float* __restrict__ g_col, float* __restrict__ scalars) {
    __shared__ float red[4];
    float sum = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rays * 3; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / 3;
        const float c = color[i] + (bg ? bg[i] : 0.0f) * (1.0f - opacity[r]);
        const float diff = c - gt[i];
        sum += sl1(diff, 5e-2f);
        g_col[i] = sl1_grad(diff, 5e-2f) * inv_n * color_w * loss_scale;
    }
    const float tot = block_sum_256(sum, red);
    if (threadIdx.x == 0) scalars[4 + blockIdx.x] = tot;
}

__global__ __launch_bounds__(64) void app_loss_final_kernel(int n_blocks, float inv_n, float* __restrict__ scalars) {
    float sum = ((int)threadIdx.x < n_blocks) ? scalars[4 + threadIdx.x] : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (threadIdx.x == 0) scalars[0] = sum * inv_n;
}


// ---- the whole ray head of a training step in ONE launch -------------------------------------------------------------------
// compositing forward (+ the per-ray distortion-loss sums) -> the loss head of the ray -> compositing backward, one wavefront
// per ray.  Nothing in the loss couples two rays except two normalisers: the batch size (a launch argument) and, for the
// distortion loss, flatten_eff_distloss's "last ray that holds a sample" -- which every wave finds for itself with one load
// from the end of packed_info (a ballot over the last 64 rays; it walks further back only over rays without samples).  The
// loss VALUES are reports: their per-ray terms are left in memory for the caller to sum when somebody asks.  Same arithmetic,
// expression by expression, as composite_fwd_kernel (Team<64>) -> geo_loss_kernel / app_loss_kernel -> composite_bwd_kernel;
// the last chunk of the forward loop is the first of the backward loop and stays in registers.
struct HeadLoss {
    const float* gt;            // geometry: distances [R]; colour: colours [R, 3]
    const float* noise;         // geometry: U[0,1) per ray or NULL
    const float* bg;            // colour: background [R, 3] or NULL
    const float* ratio_dev;     // geometry: the distortion-loss ramp (device scalar) or NULL
    float inv_bs;               // 1 / global batch (colour: 1 / (3 global batch))
    float w0;                   // depth_weight (colour: color_weight)
    float dist_w;
    float loss_scale;
    int data_parallel;
};

// ray teams for batches whose per-sample arrays hold fewer than 32 rows per ray (sample_rows <= 0: unknown -> teams)
static inline bool head_by_teams(int64_t n_rays, int64_t sample_rows) { return sample_rows <= 0 || sample_rows < 32 * n_rays; }

template <bool APP, bool TEAMS>
__global__ __launch_bounds__(256) void train_head_kernel(const float* __restrict__ sig, const float* __restrict__ rgb,
                                                         const float* __restrict__ ts, const float* __restrict__ te,
                                                         const int32_t* __restrict__ packed, int64_t n_rays, HeadLoss hl,
                                                         float* __restrict__ weights, float* __restrict__ trans,
                                                         float* __restrict__ opacity, float* __restrict__ distance,
                                                         float* __restrict__ color, float* __restrict__ terms,
                                                         float* __restrict__ distloss, float* __restrict__ inv_n_out,
                                                         float* __restrict__ d_sig, float* __restrict__ d_rgb) {
    // The one thing the loss shares between rays besides the batch size -- 1 / (last ray that holds a sample + 1) -- is found by the WAVE,
    // before its lanes split into ray teams: a ballot over the last 64 rays (it walks further back only over rays without samples).
    float inv_n = hl.inv_bs;
    if (!APP && !hl.data_parallel) {
        const int wl = threadIdx.x & 63;
        const int in_tail = (n_rays - 1 - wl >= 0) ? packed[2 * (n_rays - 1 - wl) + 1] : 0;
        float lastf = -1.f;
        unsigned long long m = __ballot(in_tail > 0);
        for (int64_t hi = n_rays; ; ) {
            if (m) { lastf = (float)(hi - 1 - __builtin_ctzll(m)); break; }
            hi -= 64;
            if (hi <= 0) break;
            const int64_t q = hi - 1 - wl;
            m = __ballot(q >= 0 && packed[2 * q + 1] > 0);
        }
        inv_n = 1.0f / (lastf + 1.0f > 0.f ? lastf + 1.0f : 1.0f);
    }
    // Ray teams as in the compositing kernels (for_rays_of_wave): a training batch late in an episode keeps a sample or two per ray -- a
    // wavefront per ray then spends its ~80 cross-lane steps (six 64-wide scans and sums each way) on two live lanes; 16- and 4-lane
    // teams take 4 and 2 steps per scan inside one DPP row.  Same bits: for <= W samples the steps beyond W only ever add exact zeros.
    auto head_of_ray = [&](auto team, int64_t r, int l) {
        constexpr int W = decltype(team)::width;
        const int64_t start = packed[2 * r];
        const int cnt = packed[2 * r + 1];
        const int n_chunks = (cnt + W - 1) / W;
        // what the loss head reads is requested NOW: the kernel is a chain of dependent round trips (packed_info -> samples ->
        // loss inputs -> backward) for a few samples per ray, and these do not depend on the forward pass
        float in_noise = 0.f, in_gt[3] = {0.f, 0.f, 0.f}, in_bg[3] = {0.f, 0.f, 0.f}, in_ratio = 1.0f;
        if (!APP) {
            if (hl.noise) in_noise = hl.noise[r];
            in_gt[0] = hl.gt[r];
            if (hl.ratio_dev) in_ratio = hl.ratio_dev[0];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { in_gt[k] = hl.gt[3 * r + k]; if (hl.bg) in_bg[k] = hl.bg[3 * r + k]; }
        }
        // ---- forward
        float carry = 0.f;
        float a_op = 0.f, a_d = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f;
        float cW = 0.f, cWM = 0.f, a_dl = 0.f;
        float w_l = 0.f, T_l = 0.f, t0_l = 0.f, t1_l = 0.f, s_l = 0.f;       // the last chunk's sample of this lane
        for (int c0 = 0; c0 < cnt; c0 += W) {
            const int i = c0 + l;
            const bool valid = i < cnt;
            float sd = 0.f, t0 = 0.f, t1 = 0.f, s = 0.f;
            if (valid) { t0 = ts[start + i]; t1 = te[start + i]; s = sig[start + i]; sd = mul_rn(s, sub_rn(t1, t0)); }
            const float ex = team_chunk_excl<W>(sd, l, carry);
            float w_dl = 0.f, T = 0.f;
            if (valid) {
                T = expf(-ex);
                const float al = 1.0f - expf(-sd);
                const float w = T * al;
                w_dl = w;
                weights[start + i] = w;
                trans[start + i] = T;
                a_op += w;
                a_d += w * ((t0 + t1) * 0.5f);
                if (rgb) {
                    a_r += w * rgb[3 * (start + i)];
                    a_g += w * rgb[3 * (start + i) + 1];
                    a_b += w * rgb[3 * (start + i) + 2];
                }
            }
            if (!APP) {
                const float m = (t0 + t1) * 0.5f, d = t1 - t0;
                const float Wp = team_chunk_excl<W>(w_dl, l, cW);
                const float WMp = team_chunk_excl<W>(w_dl * m, l, cWM);
                if (valid) a_dl += d * w_dl * w_dl * (1.0f / 3.0f) + 2.0f * w_dl * (m * Wp - WMp);
            }
            w_l = w_dl; T_l = T; t0_l = t0; t1_l = t1; s_l = s;
        }
        if (!APP) a_dl = team_sum<W>(a_dl);
        a_op = team_sum<W>(a_op); a_d = team_sum<W>(a_d);
        if (rgb) { a_r = team_sum<W>(a_r); a_g = team_sum<W>(a_g); a_b = team_sum<W>(a_b); }
        if (l == 0) {
            if (!APP) distloss[r] = a_dl;
            opacity[r] = a_op;
            distance[r] = a_d;
            if (rgb && color) { color[3 * r] = a_r; color[3 * r + 1] = a_g; color[3 * r + 2] = a_b; }
        }
        // ---- the ray's loss head
        float gop = 0.f, gd = 0.f, gc0 = 0.f, gc1 = 0.f, gc2 = 0.f, dl_scale = 0.f;
        if (!APP) {
            const float op = a_op;
            const float nz = hl.noise ? (in_noise * 2.0f - 1.0f) : 0.0f;
            const float pre = a_d + nz * (1.0f - op);
            const float d = fmaxf(pre, 0.0f);
            const float diff = d - in_gt[0];
            gd = (pre > 0.0f) ? sl1_grad(diff, 1e-2f) * hl.inv_bs * hl.w0 * hl.loss_scale : 0.0f;
            gop = -nz * gd;
            if (l == 0) terms[r] = sl1(diff, 1e-2f);
            const float ratio = in_ratio;
            dl_scale = inv_n * hl.dist_w * ratio * hl.loss_scale;
            if (r == 0 && l == 0 && inv_n_out) inv_n_out[0] = inv_n;
        } else {
            const float om = 1.0f - a_op;
            const float acc[3] = {a_r, a_g, a_b};
            float g[3], term = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float c = acc[k] + (hl.bg ? in_bg[k] : 0.0f) * om;
                const float diff = c - in_gt[k];
                term += sl1(diff, 5e-2f);
                g[k] = sl1_grad(diff, 5e-2f) * hl.inv_bs * hl.w0 * hl.loss_scale;
            }
            gc0 = g[0]; gc1 = g[1]; gc2 = g[2];
            if (l == 0) terms[r] = term;
        }
        if (cnt == 0) return;
        // ---- backward (composite_bwd_kernel; geometry: with the distortion-loss gradient formed here, colour: d rgb only)
        const float totW = a_op, totWM = a_d;
        float sufW = 0.f, sufWM = 0.f, bcarry = 0.f;
        for (int q = n_chunks - 1; q >= 0; --q) {
            const int i = q * W + l;
            const bool valid = i < cnt;
            float w = 0.f, T = 0.f, t0 = 0.f, t1 = 0.f, s = 0.f;
            if (q == n_chunks - 1) { w = w_l; T = T_l; t0 = t0_l; t1 = t1_l; s = s_l; }
            else if (valid) { w = weights[start + i]; T = trans[start + i]; t0 = ts[start + i]; t1 = te[start + i]; s = sig[start + i]; }
            if (APP) {
                if (valid) { d_rgb[3 * (start + i)] = w * gc0; d_rgb[3 * (start + i) + 1] = w * gc1; d_rgb[3 * (start + i) + 2] = w * gc2; }
                continue;
            }
            float G = 0.f;
            if (valid) G = gop + gd * ((t0 + t1) * 0.5f);
            {
                const float m = (t0 + t1) * 0.5f, wm = w * m;
                float sw = w, swm = wm;
#pragma unroll
                for (int off = 1; off < W; off <<= 1) {
                    const float y0 = __shfl_down(sw, off, W), y1 = __shfl_down(swm, off, W);
                    if (l + off < W) { sw += y0; swm += y1; }
                }
                const float Wsuf = sufW + (sw - w), WMsuf = sufWM + (swm - wm);
                const float Wt = totW - Wsuf - w, WM = totWM - WMsuf - wm;
                if (valid) G += dl_scale * ((2.0f / 3.0f) * (t1 - t0) * w + 2.0f * (m * (Wt - Wsuf) - (WM - WMsuf)));
                sufW += __shfl(sw, 0, W); sufWM += __shfl(swm, 0, W);
            }
            const float qv = G * w;
            float suf = qv;
#pragma unroll
            for (int off = 1; off < W; off <<= 1) {
                const float y = __shfl_down(suf, off, W);
                if (l + off < W) suf += y;
            }
            const float later = bcarry + (suf - qv);
            if (valid) {
                const float delta = t1 - t0;
                d_sig[start + i] = delta * ((G * T) * expf(-s * delta) - later);
            }
            bcarry += __shfl(suf, 0, W);
        }
    };
    // TEAMS = false (chosen by the host for batches whose sample arrays hold >= 32 rows per ray: the fixed-count benchmark step, the
    // first steps of an episode): a wavefront per ray and nothing else -- 50 registers and a third of the code; with the team shapes in the
    // same kernel (77 registers) rays of 128 samples took 7 % (geometry) / 35 % (colour) longer
    if constexpr (TEAMS) {
        for_rays_of_wave(n_rays, [&](int64_t q) { return packed[2 * q + 1]; }, head_of_ray);
    } else {
        const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (r < n_rays) head_of_ray(Team<64>{}, r, (int)(threadIdx.x & 63));
    }
}

}  // namespace perf

extern "C" int perf_geo_loss(const float* opacity, const float* distance, const float* gt_distance, const float* noise,
                             const float* distloss_per_ray, const int32_t* packed_info, int64_t n_rays, int64_t global_batch,
                             float depth_weight, float distortion_weight, const float* ratio_dev, float loss_scale,
                             float* g_opacity, float* g_distance, float* scalars, int32_t* ticket, void* stream) {
    PERF_REQUIRE(n_rays > 0 && global_batch > 0, "perf_geo_loss: empty batch");
    PERF_REQUIRE(opacity && distance && gt_distance && distloss_per_ray && packed_info && g_opacity && g_distance && scalars && ticket, "NULL pointer");
    const int nb = (int)((n_rays + 255) / 256 < perf::kLossBlocks ? (n_rays + 255) / 256 : perf::kLossBlocks);
    const float inv_bs = 1.0f / (float)global_batch;
    hipLaunchKernelGGL(perf::geo_loss_kernel, dim3(nb), dim3(256), 0, perf::as_stream(stream), opacity, distance, gt_distance, noise,
                       distloss_per_ray, packed_info, n_rays, inv_bs, depth_weight, loss_scale, g_opacity, g_distance, scalars,
                       distortion_weight, ratio_dev, (int)(n_rays != global_batch), ticket);
    PERF_LAUNCH_CHECK("perf_geo_loss");
    return PERF_OK;
}

extern "C" int perf_app_loss(const float* opacity, const float* color, const float* bg_color, const float* gt_color,
                             int64_t n_rays, int64_t global_batch, float color_weight, float loss_scale, float* g_color,
                             float* scalars, void* stream) {
    PERF_REQUIRE(n_rays > 0 && global_batch > 0, "perf_app_loss: empty batch");
    PERF_REQUIRE(opacity && color && gt_color && g_color && scalars, "NULL pointer");
    const int nb = (int)((n_rays * 3 + 255) / 256 < perf::kLossBlocks ? (n_rays * 3 + 255) / 256 : perf::kLossBlocks);
    const float inv_n = 1.0f / (float)(global_batch * 3);
    hipLaunchKernelGGL(perf::app_loss_kernel, dim3(nb), dim3(256), 0, perf::as_stream(stream), opacity, color, bg_color, gt_color,
                       n_rays, inv_n, color_weight, loss_scale, g_color, scalars);
    hipLaunchKernelGGL(perf::app_loss_final_kernel, dim3(1), dim3(64), 0, perf::as_stream(stream), nb, inv_n, scalars);
    PERF_LAUNCH_CHECK("perf_app_loss");
    return PERF_OK;
}

extern "C" int perf_train_head_geo(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                                   const int32_t* packed_info, int64_t n_rays, int64_t sample_rows, const float* gt_distance, const float* noise,
                                   int64_t global_batch, float depth_weight, float distortion_weight, const float* ratio_dev,
                                   float loss_scale, float* weights, float* trans, float* opacity, float* distance, float* color,
                                   float* depth_terms, float* distloss_per_ray, float* inv_n_out, float* d_sigmas, void* stream) {
    PERF_REQUIRE(n_rays > 0 && global_batch > 0, "perf_train_head_geo: empty batch");
    PERF_REQUIRE(sigmas && t_starts && t_ends && packed_info && gt_distance && weights && trans && opacity && distance && depth_terms &&
                 distloss_per_ray && d_sigmas, "NULL pointer");
    PERF_REQUIRE(!rgbs || color, "perf_train_head_geo: rgbs without a colour output");
    perf::HeadLoss hl{gt_distance, noise, nullptr, ratio_dev, 1.0f / (float)global_batch, depth_weight, distortion_weight, loss_scale,
                      (int)(n_rays != global_batch)};
    if (perf::head_by_teams(n_rays, sample_rows))
        hipLaunchKernelGGL((perf::train_head_kernel<false, true>), perf::team_grid(n_rays), dim3(256), 0, perf::as_stream(stream), sigmas, rgbs, t_starts,
                           t_ends, packed_info, n_rays, hl, weights, trans, opacity, distance, color, depth_terms, distloss_per_ray, inv_n_out,
                           d_sigmas, (float*)nullptr);
    else
        hipLaunchKernelGGL((perf::train_head_kernel<false, false>), ray_grid(n_rays), dim3(256), 0, perf::as_stream(stream), sigmas, rgbs, t_starts,
                           t_ends, packed_info, n_rays, hl, weights, trans, opacity, distance, color, depth_terms, distloss_per_ray, inv_n_out,
                           d_sigmas, (float*)nullptr);
    PERF_LAUNCH_CHECK("perf_train_head_geo");
    return PERF_OK;
}

extern "C" int perf_train_head_app(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                                   const int32_t* packed_info, int64_t n_rays, int64_t sample_rows, const float* bg_color, const float* gt_color,
                                   int64_t global_batch, float color_weight, float loss_scale, float* weights, float* trans,
                                   float* opacity, float* distance, float* color, float* color_terms, float* d_rgbs, void* stream) {
    PERF_REQUIRE(n_rays > 0 && global_batch > 0, "perf_train_head_app: empty batch");
    PERF_REQUIRE(sigmas && rgbs && t_starts && t_ends && packed_info && gt_color && weights && trans && opacity && distance && color &&
                 color_terms && d_rgbs, "NULL pointer");
    perf::HeadLoss hl{gt_color, nullptr, bg_color, nullptr, 1.0f / (float)(global_batch * 3), color_weight, 0.0f, loss_scale, 0};
    if (perf::head_by_teams(n_rays, sample_rows))
        hipLaunchKernelGGL((perf::train_head_kernel<true, true>), perf::team_grid(n_rays), dim3(256), 0, perf::as_stream(stream), sigmas, rgbs, t_starts,
                           t_ends, packed_info, n_rays, hl, weights, trans, opacity, distance, color, color_terms, (float*)nullptr, (float*)nullptr,
                           (float*)nullptr, d_rgbs);
    else
        hipLaunchKernelGGL((perf::train_head_kernel<true, false>), ray_grid(n_rays), dim3(256), 0, perf::as_stream(stream), sigmas, rgbs, t_starts,
                           t_ends, packed_info, n_rays, hl, weights, trans, opacity, distance, color, color_terms, (float*)nullptr, (float*)nullptr,
                           (float*)nullptr, d_rgbs);
    PERF_LAUNCH_CHECK("perf_train_head_app");
    return PERF_OK;
}
