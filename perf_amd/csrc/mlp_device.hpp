// 64-wide bias-free MLP on gfx950 MFMA (tcnn "FullyFusedMLP" semantics).
//
// One wave owns a tile of 32 samples and chains v_mfma_f32_32x32x16_{bf16,f16}:
//     H^T[64 x 32] = W[64 x K] * X^T[K x 32]        (neurons = M rows, samples = N columns)
// so the accumulator of one layer (lane = sample column, registers = neuron rows) IS the B
// operand of the next layer after ReLU + 16-bit packing -- no cross-lane traffic, no LDS.  The
// price is a fixed permutation of the K slots, which is folded into the weight (A) fragments once
// per block when they are staged into LDS:
//   D layout (32x32 tile):  col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
//   operand slot (h=lane>>5, j=0..7) of k-step s=2*m1+t  <->  neuron 32*m1 + 16*t + 8*(j>>2) + 4*h + (j&3)
//   input   slot (h, j) of k-step s                       <->  level 8*s + 2*(j>>1) + h, feature j&1
// Input features arrive LEVEL-MAJOR (feat[l][sample] = packed pair), so each B-fragment dword is
// one coalesced 128-byte read per half-wave.
//
// Backward recomputes the forward in registers (only feat is kept from the forward pass), chains
// dH = W^T dY the same way with transposed weight fragments, and forms the weight gradients
// dW = dH * H^T (contraction over the 32 samples) through a wave-private LDS transpose.
//
// This header holds the kernels and their launchers; they are INSTANTIATED in six translation units (mlp_fwd_{bf16,fp16}.hip,
// mlp_bwd_{bf16,fp16}_nh{1,2}.hip) so that the build compiles them in parallel -- one unit took 65 s --; mlp.hip holds the C-ABI entry points.
#pragma once
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include "common.hpp"
#include "grid_device.hpp"

namespace perf {

struct MlpParams {
    int32_t n_levels;
    int32_t n_out;
    int32_t out_act;
    float exp_shift;
};

constexpr int kTile = 32;      // samples per wave tile
// LDS tiles of the backward pass: [sample][channel position], kPitchT 16-bit elements per sample row (64 channels + pad; 72
// and 80 measured alike: 78.0 / 79.5 us for the colour network).  A lane WRITES the eight channels of a k-step it
// holds for its sample with one 16-byte store; the weight-gradient products READ them channel-major -- eight consecutive
// samples of one channel per lane -- with ds_read_b64_tr_b16 (two per operand).  Round 4 stored the tiles channel-major with
// 2-byte stores: 152 of the ~200 LDS instructions per tile of the colour network, and the LDS pipe was the busiest unit of the
// kernel (SQ_ACTIVE_INST_LDS: 78 % of the density network's backward).
constexpr int kPitchT = 72;

__device__ __forceinline__ int slot_neuron(int s, int h, int j) {
    return 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
}
__device__ __forceinline__ int d_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int NH, int KS>
struct Layout {
    static constexpr int n_in_pad = 16 * KS;
    static constexpr int w1_off = 0;
    static constexpr int w2_off = 64 * n_in_pad;
    static constexpr int wo_off = w2_off + (NH == 2 ? 64 * 64 : 0);
    static constexpr int n_params = wo_off + 16 * 64;
    // forward fragments: A1[m][s] (2*KS), A2[m][s] (8 if NH==2), Ao[s] (4)
    static constexpr int f_a1 = 0;
    static constexpr int f_a2 = 2 * KS;
    static constexpr int f_ao = f_a2 + (NH == 2 ? 8 : 0);
    static constexpr int n_fwd = f_ao + 4;
    // backward (transposed) fragments: AoT[m] (2), A2T[m][s] (8 if NH==2), A1T[mb][s] (4 per block of 32 input features)
    static constexpr int MB = (n_in_pad + 31) / 32;
    static constexpr int f_aot = n_fwd;
    static constexpr int f_a2t = f_aot + 2;
    static constexpr int f_a1t = f_a2t + (NH == 2 ? 8 : 0);
    static constexpr int n_all = f_a1t + 4 * MB;
};

__device__ __forceinline__ u32x4 pack8(const uint16_t v[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (uint32_t)v[2 * i] | ((uint32_t)v[2 * i + 1] << 16);
    return r;
}

// Element of the weight vector that slot j of lane (c, h) of fragment f holds, or -1 for a zero.
template <int NH, int KS>
__device__ __forceinline__ int frag_src(int f, int c, int h, int j) {
    using L = Layout<NH, KS>;
    if (f < L::f_a2) {                           // A1[m][s]: row = neuron 32m+c, slot -> input feature
        const int m = (f - L::f_a1) / KS, s = (f - L::f_a1) % KS;
        return L::w1_off + (32 * m + c) * L::n_in_pad + 2 * (8 * s + 2 * (j >> 1) + h) + (j & 1);
    }
    if (f < L::f_ao) {                           // A2[m][s]
        const int m = (f - L::f_a2) >> 2, s = (f - L::f_a2) & 3;
        return L::w2_off + (32 * m + c) * 64 + slot_neuron(s, h, j);
    }
    if (f < L::n_fwd)                            // Ao[s]: rows 0..15 = output layer, 16..31 zero
        return (c < 16) ? L::wo_off + c * 64 + slot_neuron(f - L::f_ao, h, j) : -1;
    if (f < L::f_a2t)                            // AoT[m]: row = neuron 32m+c, slot (h,j) -> output row d_row(j,h)
        return L::wo_off + d_row(j, h) * 64 + 32 * (f - L::f_aot) + c;
    if (f < L::f_a1t) {                          // A2T[m][s]: row = input neuron 32m+c, slot -> output neuron
        const int m = (f - L::f_a2t) >> 2, s = (f - L::f_a2t) & 3;
        return L::w2_off + slot_neuron(s, h, j) * 64 + 32 * m + c;
    }
    const int mb = (f - L::f_a1t) >> 2, s = (f - L::f_a1t) & 3;      // A1T[mb][s]: row = input feature 32*mb + c, slot -> neuron
    const int in = 32 * mb + c;
    return (in < L::n_in_pad) ? L::w1_off + slot_neuron(s, h, j) * L::n_in_pad + in : -1;
}

// Stage permuted weight fragments into LDS: frag f occupies lds[f*64 + lane] (16 B per lane).  The elements of ALL of a wave's
// fragments are requested before any is packed: one memory round trip per block instead of one per fragment.
template <int NH, int KS, bool BWD>
__device__ __forceinline__ void stage_fragments(const uint16_t* __restrict__ w, u32x4* lds) {
    using L = Layout<NH, KS>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;     // (blocks of four waves)
    const int c = lane & 31, h = lane >> 5;
    constexpr int nf = BWD ? L::n_all : L::n_fwd;
    constexpr int per_wave = (nf + 3) / 4;
    uint16_t v[per_wave][8];
#pragma unroll
    for (int k = 0; k < per_wave; ++k) {
        const int f = wave + 4 * k < nf ? wave + 4 * k : nf - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int src = frag_src<NH, KS>(f, c, h, j);
            v[k][j] = w[src < 0 ? 0 : src];
        }
    }
#pragma unroll
    for (int k = 0; k < per_wave; ++k) {
        const int f = wave + 4 * k;
        if (f < nf) {
            uint16_t z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = frag_src<NH, KS>(f, c, h, j) < 0 ? (uint16_t)0 : v[k][j];
            lds[f * 64 + lane] = pack8(z);
        }
    }
}

// max(x, 0) as ONE integer v_max_i32 on the bit pattern (negative floats are negative ints; fmaxf costs a canonicalising
// v_max_f32 plus the v_max_f32 itself)
__device__ __forceinline__ float relu(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// ReLU masks of the backward pass come from the PACKED activations themselves (what tcnn's backward does too: it masks with the
// stored 16-bit forward activation): a post-ReLU 16-bit value is non-negative, so "active" is "bits != 0", and a pair of them
// turns into a pair of 0xffff / 0 half-word masks with two packed 16-bit instructions (negate, arithmetic shift by 15).  Rounds 1-4 kept the
// masks of the last hidden layer as 32 wave predicates (64 scalar registers: more than the file holds beside everything else --
// the kernel spilled them to vector lanes, ~400 v_readlane / v_writelane in the colour network's loop) and those of the layer
// before as a per-lane bit field (a compare, a select and an or per element to build, an and, a compare and a select to apply).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t nonzero_halves(uint32_t packed) {      // (halves in [0, 0x7fff]: 0 - x is negative iff x != 0)
    const s16x2 t = (s16x2){0, 0} - __builtin_bit_cast(s16x2, packed);
    return __builtin_bit_cast(uint32_t, (s16x2)(t >> (s16x2){15, 15}));
}
template <typename T16>
__device__ __forceinline__ void relu_pack_plain(const f32x16& acc, u32x4& lo, u32x4& hi) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] = T16::pack(relu(acc[2 * i]), relu(acc[2 * i + 1]));
        hi[i] = T16::pack(relu(acc[8 + 2 * i]), relu(acc[8 + 2 * i + 1]));
    }
}
// pack d where the forward activation (same rows, same lanes: hlo / hhi) is active
template <typename T16>
__device__ __forceinline__ void pack_active(const f32x16& d, const u32x4& hlo, const u32x4& hhi, u32x4& lo, u32x4& hi) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] = T16::pack(d[2 * i], d[2 * i + 1]) & nonzero_halves(hlo[i]);
        hi[i] = T16::pack(d[8 + 2 * i], d[8 + 2 * i + 1]) & nonzero_halves(hhi[i]);
    }
}

__device__ __forceinline__ float act_fwd(float y, int act, float shift) {
    if (act == PERF_ACT_SIGMOID) return 1.0f / (1.0f + expf(-y));
    if (act == PERF_ACT_EXP) return expf(y - shift);
    return y;
}
__device__ __forceinline__ float act_bwd(float y, float g, int act, float shift) {
    if (act == PERF_ACT_SIGMOID) { float s = 1.0f / (1.0f + expf(-y)); return g * s * (1.0f - s); }
    if (act == PERF_ACT_EXP) return g * expf(fminf(y - shift, 15.0f));
    return g;
}

// Features and their gradient are level-major: element (level, sample) at level * n + sample.  A lane's level is
// (a compile-time part) + (its half h) * const, so the address is a UNIFORM base per access plus ONE 32-bit lane offset per
// tile -- the saddr + voffset form of global_load / global_store -- instead of a 64-bit multiply-add and a branch per level
// (which was more than half of the kernel's vector instructions).  Holds while the offsets fit 32 bits.
constexpr int64_t kMaxFastStride = (int64_t)1 << 27;

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// FUSED: encode + MLP in one kernel for small batches (no level-group / XCD pinning is at stake below ~64 k samples): the wave
// forms the B operand of the first layer IN REGISTERS -- lane (sample c, half h) of k-step s owns the levels 8s + 2i + h,
// i = 0..3, i.e. it gathers and interpolates eight levels of its sample (encode_pair: the very function the level-major
// encode kernels use, so the packed pairs are bit-identical) -- and the features never travel through memory, unless the
// caller wants them (feat_out: the density pass of the sampler keeps them for the gradient pass).
struct FusedIn {
    GridParams gp;
    const uint32_t* table;
    const float* x01;
    uint32_t* feat_out;
};
struct NoFusedIn {};

template <typename T16, int NH, int KS, bool FUSED>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpParams mp, const uint16_t* __restrict__ w,
                                                      const uint32_t* __restrict__ feat,
                                                      const uint8_t* __restrict__ sel, float* __restrict__ out,
                                                      int64_t n, const int64_t* __restrict__ n_dev,
                                                      std::conditional_t<FUSED, FusedIn, NoFusedIn> fz) {
    using L = Layout<NH, KS>;
    const int64_t n_live = live_count(n, n_dev);        // n stays the stride of the level-major features
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* frag = reinterpret_cast<u32x4*>(smem);
    stage_fragments<NH, KS, false>(w, frag);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n_live + kTile - 1) / kTile;
    struct TileIn { u32x4 b1[KS]; uint8_t sv; };
    // the layers and the store of one tile, given the packed first-layer operand
    auto layers = [&](int64_t si, bool valid, const u32x4 (&b1)[KS], float sv) __attribute__((always_inline)) {
        f32x16 acc[2];
        u32x4 hb[4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            acc[m] = f32x16{0};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc[m] = T16::mfma(frag[(L::f_a1 + m * KS + s) * 64 + lane], b1[s], acc[m]);
            relu_pack_plain<T16>(acc[m], hb[2 * m], hb[2 * m + 1]);
        }
        if constexpr (NH == 2) {
            u32x4 hb2[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m] = f32x16{0};
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[m] = T16::mfma(frag[(L::f_a2 + m * 4 + s) * 64 + lane], hb[s], acc[m]);
                relu_pack_plain<T16>(acc[m], hb2[2 * m], hb2[2 * m + 1]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) hb[s] = hb2[s];
        }
        f32x16 o = f32x16{0};
#pragma unroll
        for (int s = 0; s < 4; ++s) o = T16::mfma(frag[(L::f_ao + s) * 64 + lane], hb[s], o);
        if (valid) {
            // the activation is a kernel argument: branch on it ONCE (scalar), and stop at the last register that can hold
            // a real output row (rows of register r: d_row(r, 0) < d_row(r, 1)) -- otherwise both exponentials are
            // evaluated for all 8 registers and selected afterwards (~250 vector instructions per tile)
            auto emit = [&](auto act) {
                if (mp.n_out == 16 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {   // registers 0..3 / 4..7 are rows 4h..4h+3 / 8+4h..: two 16-byte stores
                    float* row0 = out + si * 16 + 4 * h;
                    *reinterpret_cast<float4*>(row0) = make_float4(act(o[0]) * sv, act(o[1]) * sv, act(o[2]) * sv, act(o[3]) * sv);
                    *reinterpret_cast<float4*>(row0 + 8) = make_float4(act(o[4]) * sv, act(o[5]) * sv, act(o[6]) * sv, act(o[7]) * sv);
                    return;
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (d_row(r, 0) >= mp.n_out) break;
                    const int row = d_row(r, h);
                    if (row < mp.n_out) out[si * mp.n_out + row] = act(o[r]) * sv;
                }
            };
            if (mp.out_act == PERF_ACT_SIGMOID) emit([](float y) { return 1.0f / (1.0f + expf(-y)); });
            else if (mp.out_act == PERF_ACT_EXP) emit([&](float y) { return expf(y - mp.exp_shift); });
            else emit([](float y) { return y; });
        }
    };
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if constexpr (FUSED) {
        const bool smooth = fz.gp.interpolation == PERF_INTERP_SMOOTHSTEP;
        for (; tile < n_tiles; tile += tile_step) {
            const int64_t si = tile * kTile + c;
            const bool valid = si < n_live;
            u32x4 b1[KS];
            float x = 0.5f, y = 0.5f, z = 0.5f;
            if (valid) { x = fz.x01[3 * si]; y = fz.x01[3 * si + 1]; z = fz.x01[3 * si + 2]; }
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int level = 8 * s + 2 * i + h;
                    uint32_t pair = 0u;
                    if (valid && level < mp.n_levels) {
                        pair = encode_pair<T16>(fz.gp, fz.table, level, x, y, z, smooth);
                        if (fz.feat_out) fz.feat_out[(int64_t)level * n + si] = pair;
                    }
                    b1[s][i] = pair;
                }
            layers(si, valid, b1, (valid && sel) ? (float)sel[si] : 1.0f);
        }
    } else if (n <= kMaxFastStride && mp.n_levels == 8 * KS) {      // (uniform) see kMaxFastStride
        // one tile ahead on alternating register sets, as in mlp_bwd_kernel: a request is unconditional, issues a fixed number
        // of loads and computes nothing from them; lanes past the end work on the LAST sample's features (finite values,
        // results discarded) rather than on zeros
        if (tile >= n_tiles) return;
        const int64_t last_tile = n_tiles - 1;
        auto request = [&](int64_t tile_unclamped, TileIn& t) {
            const int64_t tl = tile_unclamped < last_tile ? tile_unclamped : last_tile;
            const int64_t si = tl * kTile + c;
            const int64_t sc = si < n_live ? si : n_live - 1;
            const uint32_t off = 4u * (uint32_t)(sc + (int64_t)h * n);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned char* base = reinterpret_cast<const unsigned char*>(feat) + (int64_t)(8 * s + 2 * i) * n * 4;
                    t.b1[s][i] = *reinterpret_cast<const uint32_t*>(base + off);
                }
            t.sv = *(sel ? sel + sc : reinterpret_cast<const uint8_t*>(feat));
        };
        auto process = [&](int64_t tl, const TileIn& t) __attribute__((always_inline)) {
            const int64_t si = tl * kTile + c;
            layers(si, si < n_live, t.b1, sel ? (float)t.sv : 1.0f);
        };
        TileIn ta, tb;
        request(tile, ta);
        for (;;) {
            request(tile + tile_step, tb);
            process(tile, ta);
            tile += tile_step;
            if (tile >= n_tiles) break;
            request(tile + tile_step, ta);
            process(tile, tb);
            tile += tile_step;
            if (tile >= n_tiles) break;
        }
    } else {
        for (; tile < n_tiles; tile += tile_step) {
            const int64_t si = tile * kTile + c;
            const bool valid = si < n_live;
            u32x4 b1[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int level = 8 * s + 2 * i + h;
                    b1[s][i] = (valid && level < mp.n_levels) ? feat[(int64_t)level * n + si] : 0u;
                }
            layers(si, valid, b1, (valid && sel) ? (float)sel[si] : 1.0f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// a packed B-fragment set (4 k-steps x 4 dwords: the 32 channels this lane owns of a 64-channel tensor, for its sample c) ->
// row c of the [sample][channel position] tile; channel POSITION p = 16 s + 8 h + j holds neuron slot_neuron(s, h, j)
__device__ __forceinline__ void lds_put_hidden(uint16_t* tile, const u32x4 fr[4], int c, int h) {
#pragma unroll
    for (int s = 0; s < 4; ++s) *reinterpret_cast<u32x4*>(tile + c * kPitchT + 16 * s + 8 * h) = fr[s];
}
__device__ __forceinline__ int position_neuron(int p) { return slot_neuron(p >> 4, (p >> 3) & 1, p & 7); }

// ds_read_b64_tr_b16 (lane map measured with tools/exp/tr_probe.hip): within a group of 16 lanes, lane i receives element
// (i & 3) of the 8 bytes addressed by lanes (i >> 2), (i >> 2) + 4, (i >> 2) + 8, (i >> 2) + 12 of the group.  With lane
// 4 k + q of the group addressing &tile[sample s0 + k][channel c0 + 4 q], lane i receives channel c0 + i of samples s0 .. s0 + 3.
typedef short v4s __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 lds_tr8(const uint16_t* p) {       // samples s0 .. s0 + 7 of the lane's channel
    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p));
    const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 4 * kPitchT)));
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}
// Operand of a 32x32x16 product over samples 16 s16 .. + 15: lane (cl = lane & 31, h = lane >> 5) gets samples 16 s16 + 8 h .. + 7
// of channel position 32 m + cl.  off32: the lane's part of the address (tr_offset32), computed once.
__device__ __forceinline__ int tr_offset32(int lane) {
    const int g = lane >> 4, i = lane & 15;
    return (8 * (g >> 1) + (i >> 2)) * kPitchT + 16 * (g & 1) + 4 * (i & 3);
}
__device__ __forceinline__ u32x4 lds_get_frag(const uint16_t* tile, int off32, int m, int s16) {
    return lds_tr8(tile + off32 + 16 * s16 * kPitchT + 32 * m);
}
// Operand of a 16x16x32 product over the tile's 32 samples: lane (r16 = lane & 15, kb = lane >> 4) gets samples 8 kb .. + 7 of
// channel position cbase + r16.
__device__ __forceinline__ int tr_offset16(int lane) {
    const int i = lane & 15;
    return (8 * (lane >> 4) + (i >> 2)) * kPitchT + 4 * (i & 3);
}

// FAST (chosen by the launcher): every level slot of the first layer is a real level (n_levels == 8 * KS) and the level-major
// offsets fit 32 bits -- the addressing above, and a FIXED number of loads per request (below).
template <typename T16, int NH, int KS, bool FAST>
// Two workgroups per CU (256 registers per lane) for KS < 3 -- except <NH = 2, KS = 2, plain>, whose general addressing does not fit 256
// registers (396 B of scratch per lane: 252 us per 1 M samples at L = 12 against 142 us with one workgroup per CU and no scratch).  The
// NH = 2, KS = 1 variants keep their 88-104 B of scratch: 110 us at two workgroups per CU against 121 us spill-free at one
// (tools/exp/mlp_bwd_scratch_ab.py; none of these is PeRF's L = 16, whose <2, 2, FAST> carries 16 B).
__global__ __launch_bounds__(256, (KS < 3 && !(NH == 2 && KS == 2 && !FAST)) ? 2 : 1) void mlp_bwd_kernel(MlpParams mp, const uint16_t* __restrict__ w,
                                                      const uint32_t* __restrict__ feat,
                                                      const int32_t* __restrict__ feat_index, int64_t feat_stride,
                                                      const uint8_t* __restrict__ sel,
                                                      const float* __restrict__ dout, float2* __restrict__ dfeat,
                                                      float* __restrict__ partials, float* __restrict__ level_absmax,
                                                      int64_t n, const int64_t* __restrict__ n_dev) {
    using L = Layout<NH, KS>;
    const int64_t n_live = live_count(n, n_dev);        // n stays the stride of dfeat (and of feat, unless feat_index is given)
    float amax = 0.f;      // running max |dfeat| over the 8 levels this half-wave owns (one register, not eight)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* frag = reinterpret_cast<u32x4*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t* tA = reinterpret_cast<uint16_t*>(smem + L::n_all * 1024) + wave * (2 * kTile * kPitchT);
    uint16_t* tB = tA + kTile * kPitchT;
    const int off32 = tr_offset32(lane), off16 = tr_offset16(lane);
    stage_fragments<NH, KS, true>(w, frag);
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n_live + kTile - 1) / kTile;

    constexpr int MB = L::MB;       // blocks of 32 input features (2 for grids of more than 16 levels)
    // dWo is 16 x 64: four 16x16x32 products (ONE k-step covers the tile's 32 samples) in 16 accumulator registers; as two
    // 32x32x16 tiles half of 32 registers held the zero rows 16..31
    f32x16 gW1[2 * MB], gW2[NH == 2 ? 4 : 1];
    f32x4 gWo[4];
#pragma unroll
    for (int m = 0; m < 2 * MB; ++m) gW1[m] = f32x16{0};
#pragma unroll
    for (int m = 0; m < 4; ++m) gWo[m] = f32x4{0};
#pragma unroll
    for (int m = 0; m < (NH == 2 ? 4 : 1); ++m) gW2[m] = f32x16{0};

    // One tile AHEAD: the inputs of the next tile (features, upstream gradient, selector) are requested before this tile's
    // work and, above all, before this tile's dfeat stores.  Loads and stores retire through ONE in-order counter, so a load
    // issued behind the stores is only known to have landed once they are acknowledged: the chain per tile was
    // store-acknowledge + load + compute; now it is compute.  What makes the compiler's s_waitcnt exact rather than "everything":
    // a request is UNCONDITIONAL and issues the same number of loads on every path (clamped indices instead of predicates, a
    // dummy byte when there is no selector), nothing is computed from a requested value before its tile's turn, and the two
    // register sets alternate (a copy at the end of the iteration would wait for the loads it copies).
    // feat_index: a sample's features are row feat_index[i] of feat_stride rows per level.  The row of the NEXT request travels
    // with the current one (TileIn::row_next: requested a tile earlier, so that the feature addresses of a request never wait
    // for a load of the same request); without an index the same slot carries a dummy word and the row is the sample itself.
    struct TileIn { u32x4 b1[KS]; float g[8]; uint8_t sv; int32_t row_next; };
    const int64_t last_tile = n_tiles - 1;
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    const bool indexed = feat_index != nullptr;         // (uniform)
    const int64_t fstride = indexed ? feat_stride : n;
    const int32_t* index_or_dummy = indexed ? feat_index : reinterpret_cast<const int32_t*>(dout);
    auto sample_of = [&](int64_t tile_unclamped) {      // (in-range sample of this lane in a tile; past the end: the last ones)
        const int64_t tile = tile_unclamped < last_tile ? tile_unclamped : last_tile;
        const int64_t si = tile * kTile + c;
        return si < n_live ? si : n_live - 1;
    };
    auto request = [&](int64_t tile_unclamped, TileIn& t, int32_t row_in) {
        const int64_t tile = tile_unclamped < last_tile ? tile_unclamped : last_tile;     // past the end: re-read the last tile
        const int64_t si = tile * kTile + c;
        const bool valid = si < n_live;
        const int64_t sc = valid ? si : n_live - 1;      // lanes past the end read the LAST sample: finite values, zero dY
        t.row_next = index_or_dummy[indexed ? sample_of(tile_unclamped + tile_step) : 0];
        const int64_t row = indexed ? (int64_t)row_in : sc;
        if constexpr (FAST) {
            const uint32_t off = 4u * (uint32_t)(row + (int64_t)h * fstride);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned char* base = reinterpret_cast<const unsigned char*>(feat) + (int64_t)(8 * s + 2 * i) * fstride * 4;
                    t.b1[s][i] = *reinterpret_cast<const uint32_t*>(base + off);
                }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int level = 8 * s + 2 * i + h;
                    t.b1[s][i] = (valid && level < mp.n_levels) ? feat[(int64_t)level * fstride + row] : 0u;
                }
        }
        const float* drow = dout + sc * mp.n_out;
#pragma unroll
        for (int r = 0; r < 8; ++r) {                    // RAW values: any arithmetic here would wait for the load
            const int row = d_row(r, h);
            t.g[r] = drow[row < mp.n_out ? row : mp.n_out - 1];
        }
        t.sv = *(sel ? sel + sc : reinterpret_cast<const uint8_t*>(drow));
    };
    // FULL (a tag type): all 32 samples of the tile are live -- every tile but possibly the last; the loop below handles FULL tiles
    // only, so that (with FAST) a tile's eight dfeat stores are unconditional and the counter arithmetic above stays exact:
    // the wait for a tile's inputs then lets the stores of the tile before and the next request stay in flight.
    auto process = [&](auto full_tag, int64_t tile, const TileIn& cur) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int64_t si = tile * kTile + c;
        const bool valid = FULL || si < n_live;
        // ---- recompute forward
        u32x4 b1[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) b1[s] = cur.b1[s];
        f32x16 acc[2];
        u32x4 hb1[4], hb2[4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            acc[m] = f32x16{0};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc[m] = T16::mfma(frag[(L::f_a1 + m * KS + s) * 64 + lane], b1[s], acc[m]);
            relu_pack_plain<T16>(acc[m], hb1[2 * m], hb1[2 * m + 1]);
        }
        if constexpr (NH == 2) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m] = f32x16{0};
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[m] = T16::mfma(frag[(L::f_a2 + m * 4 + s) * 64 + lane], hb1[s], acc[m]);
                relu_pack_plain<T16>(acc[m], hb2[2 * m], hb2[2 * m + 1]);
            }
        }
        const u32x4* hlast = (NH == 2) ? hb2 : hb1;
        f32x16 o = f32x16{0};
#pragma unroll
        for (int s = 0; s < 4; ++s) o = T16::mfma(frag[(L::f_ao + s) * 64 + lane], hlast[s], o);
        // ---- output gradient (activation derivative and selector applied here)
        float dy[8];
        const float sv = sel ? (float)cur.sv : 1.0f;
#pragma unroll
        for (int r = 0; r < 8; ++r) dy[r] = 0.f;
        {
            auto emit = [&](auto dact) {         // (scalar branch on the activation, registers without a real output row skipped)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (d_row(r, 0) >= mp.n_out) break;
                    const int row = d_row(r, h);
                    if (valid && row < mp.n_out) dy[r] = dact(o[r], cur.g[r] * sv);
                }
            };
            if (mp.out_act == PERF_ACT_SIGMOID) emit([](float y, float g) { const float s_ = 1.0f / (1.0f + expf(-y)); return g * s_ * (1.0f - s_); });
            else if (mp.out_act == PERF_ACT_EXP) emit([&](float y, float g) { return g * expf(fminf(y - mp.exp_shift, 15.0f)); });
            else emit([](float, float g) { return g; });
        }
        u32x4 dyb;
#pragma unroll
        for (int i = 0; i < 4; ++i) dyb[i] = T16::pack(dy[2 * i], dy[2 * i + 1]);
        // ---- the transposes of the output layer's weight gradient (dWo[16 x 64] += dY * Hlast^T) go to LDS FIRST: the
        //      products below do not need them and cover the round trip
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<u32x4*>(tA + c * kPitchT + 8 * h) = dyb;     // position 8 h + j holds output row d_row(j, h)
        lds_put_hidden(tB, hlast, c, h);
        __builtin_amdgcn_wave_barrier();
        // ---- dH_last = Wo^T dY, masked
        u32x4 dhl[4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x16 d = T16::mfma(frag[(L::f_aot + m) * 64 + lane], dyb, f32x16{0});
            pack_active<T16>(d, hlast[2 * m], hlast[2 * m + 1], dhl[2 * m], dhl[2 * m + 1]);
        }
        __builtin_amdgcn_wave_barrier();
        {   // operands of the 16x16x32 form: lane (row lane & 15, k-block lane >> 4) holds samples 8 * (lane >> 4) .. + 7
            const u32x4 a = lds_tr8(tA + off16);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) gWo[nb] = T16::mfma16(a, lds_tr8(tB + off16 + 16 * nb), gWo[nb]);
        }
        u32x4 dh1[4];
        if constexpr (NH == 2) {
            // ---- dH1 = W2^T dH2, masked by layer-1 activations
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x16 d = f32x16{0};
#pragma unroll
                for (int s = 0; s < 4; ++s) d = T16::mfma(frag[(L::f_a2t + m * 4 + s) * 64 + lane], dhl[s], d);
                pack_active<T16>(d, hb1[2 * m], hb1[2 * m + 1], dh1[2 * m], dh1[2 * m + 1]);
            }
            // ---- dW2[64 x 64] += dH2 * H1^T
            __builtin_amdgcn_wave_barrier();
            lds_put_hidden(tA, dhl, c, h);
            lds_put_hidden(tB, hb1, c, h);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const u32x4 b0 = lds_get_frag(tB, off32, 0, s), b1t = lds_get_frag(tB, off32, 1, s);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const u32x4 a = lds_get_frag(tA, off32, m, s);
                    gW2[2 * m] = T16::mfma(a, b0, gW2[2 * m]);
                    gW2[2 * m + 1] = T16::mfma(a, b1t, gW2[2 * m + 1]);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) dh1[s] = dhl[s];
        }
        // ---- the transposes of dW1[64 x n_in_pad] += dH1 * X^T first, the dX products and stores cover their round trip
        __builtin_amdgcn_wave_barrier();
        lds_put_hidden(tA, dh1, c, h);
#pragma unroll
        for (int s = 0; s < KS; ++s)       // position 16 s + 8 h + j holds input feature 2 (8 s + 2 (j >> 1) + h) + (j & 1)
            *reinterpret_cast<u32x4*>(tB + c * kPitchT + 16 * s + 8 * h) = b1[s];
        __builtin_amdgcn_wave_barrier();
        // ---- dX = W1^T dH1 (rows = input features in natural order 2*level+feat)
        if (FAST || dfeat != nullptr) {         // (the launcher sends a call without dfeat to the kernel without FAST)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                f32x16 dx = f32x16{0};
#pragma unroll
                for (int s = 0; s < 4; ++s) dx = T16::mfma(frag[(L::f_a1t + 4 * mb + s) * 64 + lane], dh1[s], dx);
                if (FAST && 16 * mb + 16 <= 8 * KS && valid) {
                    // register pair (2q,2q+1) -> level 16*mb + d_row(2q,h)/2 = 16*mb + 4*(q>>1) + (q&1) + 2*h: all real levels
                    const uint32_t off = 8u * (uint32_t)(si + (int64_t)(2 * h) * n);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        unsigned char* base = reinterpret_cast<unsigned char*>(dfeat) + (int64_t)(16 * mb + 4 * (q >> 1) + (q & 1)) * n * 8;
                        *reinterpret_cast<float2*>(base + off) = make_float2(dx[2 * q], dx[2 * q + 1]);
                        amax = fmaxf(amax, fmaxf(fabsf(dx[2 * q]), fabsf(dx[2 * q + 1])));
                    }
                } else if (valid) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int level = 16 * mb + (d_row(2 * q, h) >> 1);
                        if (level < mp.n_levels) {
                            dfeat[(int64_t)level * n + si] = make_float2(dx[2 * q], dx[2 * q + 1]);
                            amax = fmaxf(amax, fmaxf(fabsf(dx[2 * q]), fabsf(dx[2 * q + 1])));
                        }
                    }
                }
            }
        }
        // ---- dW1
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int nb = 0; nb < MB; ++nb) {
                u32x4 b = lds_get_frag(tB, off32, nb, s);            // (every lane takes part in the transposing read)
                if (32 * nb + c >= L::n_in_pad) b = u32x4{0, 0, 0, 0};
#pragma unroll
                for (int m = 0; m < 2; ++m) gW1[m * MB + nb] = T16::mfma(lds_get_frag(tA, off32, m, s), b, gW1[m * MB + nb]);
            }
    };
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_full = n_live / kTile;            // tiles without a lane past the end
    TileIn ta, tb;
    auto first_row = [&](int64_t tl) { return indexed ? feat_index[sample_of(tl)] : 0; };
    if (tile < n_full) {
        // (the first tile is peeled so that the loop is entered, like its back edge, with one request and one tile's stores in flight)
        request(tile, ta, first_row(tile));
        request(tile + tile_step, tb, ta.row_next);
        process(std::true_type{}, tile, ta);
        tile += tile_step;
        while (tile < n_full) {
            request(tile + tile_step, ta, tb.row_next);
            process(std::true_type{}, tile, tb);
            tile += tile_step;
            if (tile >= n_full) break;
            request(tile + tile_step, tb, ta.row_next);
            process(std::true_type{}, tile, ta);
            tile += tile_step;
        }
    }
    if (tile == n_full && n_full < n_tiles) {         // the ragged last tile, on the wave whose turn it is
        request(tile, ta, first_row(tile));
        process(std::false_type{}, tile, ta);
    }
    // ---- per-level max |dfeat| (feeds the fixed-point scale of the grid backward): lanes of one half-wave hold the
    //      same 8 levels, non-negative floats order like their bit patterns
    if (level_absmax != nullptr) {      // per-(wave, half) max -> workspace slot; mlp_reduce_kernel folds them (no atomics)
        float v = amax;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
        if (c == 0) level_absmax[((int64_t)blockIdx.x * 4 + wave) * 2 + h] = v;
    }
    // ---- block reduction of the four waves' accumulators through LDS (lane-linear slots, conflict free),
    //      then ONE partial per block -> global (summed by mlp_reduce_kernel)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    constexpr int kW1 = 2 * MB;
    constexpr int kAcc = kW1 + 1 + (NH == 2 ? 4 : 0);        // accumulators per lane, in units of 16 registers
    for (int src = 1; src < 4; ++src) {
        if (wave == src) {
#pragma unroll
            for (int m = 0; m < kW1; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((m) * 16 + r) * 64 + lane] = gW1[m][r];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(kW1 * 16 + 4 * m + r) * 64 + lane] = gWo[m][r];
            if constexpr (NH == 2) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((kW1 + 1 + m) * 16 + r) * 64 + lane] = gW2[m][r];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int m = 0; m < kW1; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) gW1[m][r] += red[((m) * 16 + r) * 64 + lane];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) gWo[m][r] += red[(kW1 * 16 + 4 * m + r) * 64 + lane];
            if constexpr (NH == 2) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) gW2[m][r] += red[((kW1 + 1 + m) * 16 + r) * 64 + lane];
            }
        }
        __syncthreads();
    }
    static_assert(kAcc * 16 * 64 * 4 <= Layout<NH, KS>::n_all * 1024 + 4 * 2 * kTile * kPitchT * 2, "reduction scratch exceeds LDS");
    if (wave != 0) return;
    float* p = partials + (int64_t)blockIdx.x * L::n_params;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nb = 0; nb < MB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {       // rows and columns of the products are channel POSITIONS of the tiles
                const int row = position_neuron(32 * m + d_row(r, h));
                const int pc = 32 * nb + c, in = (pc & ~15) + 4 * ((pc & 7) >> 1) + 2 * ((pc >> 3) & 1) + (pc & 1);
                if (pc < L::n_in_pad) p[L::w1_off + row * L::n_in_pad + in] = gW1[m * MB + nb][r];
            }
    if constexpr (NH == 2) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    p[L::w2_off + position_neuron(32 * m + d_row(r, h)) * 64 + position_neuron(32 * nn + c)] = gW2[2 * m + nn][r];
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)            // 16x16 result: row = 4 * (lane >> 4) + register, column = lane & 15
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int op = 4 * (lane >> 4) + r;                   // position 8 h + j of the dY tile holds output row d_row(j, h)
            p[L::wo_off + d_row(op & 7, op >> 3) * 64 + position_neuron(16 * nb + (lane & 15))] = gWo[nb][r];
        }
}

static inline int mlp_blocks(int64_t n, int per_cu) {
    int64_t tiles = div_up(n, kTile);
    int64_t want = div_up(tiles, 4);
    int64_t cap = (int64_t)kNumCU * per_cu;
    return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}

static int check_mlp(const perf_mlp_desc* m, int* nh, int* ks) {
    PERF_REQUIRE(m != nullptr, "mlp desc is NULL");
    PERF_REQUIRE(m->n_levels >= 1 && m->n_levels <= 24, "mlp n_levels %d out of range", m->n_levels);
    PERF_REQUIRE(m->n_hidden_layers == 1 || m->n_hidden_layers == 2, "n_hidden_layers must be 1 or 2");
    PERF_REQUIRE(m->n_out >= 1 && m->n_out <= 16, "n_out out of range");
    PERF_REQUIRE(m->out_act >= 0 && m->out_act <= 2, "bad out_act");
    *nh = m->n_hidden_layers;
    *ks = m->n_levels > 16 ? 3 : (m->n_levels > 8 ? 2 : 1);
    return PERF_OK;
}

template <int NH, int KS>
static int n_params_of() { return Layout<NH, KS>::n_params; }

static int n_params_rt(int nh, int ks) {
    if (nh == 1) return ks == 1 ? n_params_of<1, 1>() : (ks == 2 ? n_params_of<1, 2>() : n_params_of<1, 3>());
    return ks == 1 ? n_params_of<2, 1>() : (ks == 2 ? n_params_of<2, 2>() : n_params_of<2, 3>());
}

// backward workgroups per CU: the one-hidden-layer kernel fits 2 waves per SIMD (<= 256 registers) for up to 16 levels,
// the two-layer one and the 17..24-level variants (two more accumulator tiles for dW1) 1
static inline int bwd_blocks_per_cu(int nh, int ks) { (void)nh; return ks < 3 ? 2 : 1; }      // (mlp_bwd_kernel's launch bounds)

}  // namespace perf

namespace perf {

template <typename T16, int NH, int KS>
static void launch_fwd(int blocks, hipStream_t st, MlpParams mp, const uint16_t* w, const uint32_t* feat, const uint8_t* sel,
                       float* out, int64_t n, const int64_t* n_dev) {
    constexpr int lds_bytes = Layout<NH, KS>::n_fwd * 1024;
    mlp_fwd_kernel<T16, NH, KS, false><<<dim3(blocks), dim3(256), lds_bytes, st>>>(mp, w, feat, sel, out, n, n_dev, NoFusedIn{});
}

template <typename T16, int NH, int KS>
static void launch_fused(int blocks, hipStream_t st, MlpParams mp, const uint16_t* w, const uint8_t* sel, float* out, int64_t n,
                         const int64_t* n_dev, FusedIn fz) {
    constexpr int lds_bytes = Layout<NH, KS>::n_fwd * 1024;
    mlp_fwd_kernel<T16, NH, KS, true><<<dim3(blocks), dim3(256), lds_bytes, st>>>(mp, w, nullptr, sel, out, n, n_dev, fz);
}

template <typename T16, typename... Args>
static void dispatch_fused(int nh, int ks, Args... a) {
    if (nh == 1 && ks == 1) launch_fused<T16, 1, 1>(a...);
    else if (nh == 1) launch_fused<T16, 1, 2>(a...);
    else if (ks == 1) launch_fused<T16, 2, 1>(a...);
    else launch_fused<T16, 2, 2>(a...);
}

template <typename T16, int NH, int KS>
static void launch_bwd(int blocks, hipStream_t st, MlpParams mp, const uint16_t* w, const uint32_t* feat, const int32_t* feat_index,
                       int64_t feat_stride, const uint8_t* sel, const float* dout, float2* dfeat, float* partials, float* level_absmax,
                       int64_t n, const int64_t* n_dev) {
    constexpr int lds_bytes = Layout<NH, KS>::n_all * 1024 + 4 * 2 * kTile * kPitchT * 2;
    static std::once_flag attr_once;            // (one flag per template instance) safe under concurrent callers
    std::call_once(attr_once, []() {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_kernel<T16, NH, KS, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_kernel<T16, NH, KS, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    });
    if (n <= kMaxFastStride && (feat_index == nullptr || feat_stride <= kMaxFastStride) && mp.n_levels == 8 * KS && dfeat != nullptr)
        mlp_bwd_kernel<T16, NH, KS, true><<<dim3(blocks), dim3(256), lds_bytes, st>>>(mp, w, feat, feat_index, feat_stride, sel, dout, dfeat, partials, level_absmax, n, n_dev);
    else
        mlp_bwd_kernel<T16, NH, KS, false><<<dim3(blocks), dim3(256), lds_bytes, st>>>(mp, w, feat, feat_index, feat_stride, sel, dout, dfeat, partials, level_absmax, n, n_dev);
}

template <typename T16, typename... Args>
static void dispatch_fwd(int nh, int ks, Args... a) {
    if (nh == 1 && ks == 1) launch_fwd<T16, 1, 1>(a...);
    else if (nh == 1 && ks == 2) launch_fwd<T16, 1, 2>(a...);
    else if (nh == 1) launch_fwd<T16, 1, 3>(a...);
    else if (ks == 1) launch_fwd<T16, 2, 1>(a...);
    else if (ks == 2) launch_fwd<T16, 2, 2>(a...);
    else launch_fwd<T16, 2, 3>(a...);
}

template <typename T16, typename... Args>
static void dispatch_bwd(int nh, int ks, Args... a) {
    if (nh == 1 && ks == 1) launch_bwd<T16, 1, 1>(a...);
    else if (nh == 1 && ks == 2) launch_bwd<T16, 1, 2>(a...);
    else if (nh == 1) launch_bwd<T16, 1, 3>(a...);
    else if (ks == 1) launch_bwd<T16, 2, 1>(a...);
    else if (ks == 2) launch_bwd<T16, 2, 2>(a...);
    else launch_bwd<T16, 2, 3>(a...);
}

template <typename T16, int NH, typename... Args>
static void dispatch_bwd_nh(int ks, Args... a) {
    if (ks == 1) launch_bwd<T16, NH, 1>(a...);
    else if (ks == 2) launch_bwd<T16, NH, 2>(a...);
    else launch_bwd<T16, NH, 3>(a...);
}

// ---- defined in the instantiation units ------------------------------------------------------------------------------
#define PERF_MLP_FWD_ARGS int nh, int ks, int blocks, hipStream_t st, MlpParams mp, const uint16_t* w, const uint32_t* feat, const uint8_t* sel, \
                          float* out, int64_t n, const int64_t* n_dev
#define PERF_MLP_FUSED_ARGS int nh, int ks, int blocks, hipStream_t st, MlpParams mp, const uint16_t* w, const uint8_t* sel, float* out, int64_t n, \
                            const int64_t* n_dev, FusedIn fz
#define PERF_MLP_BWD_ARGS int ks, int blocks, hipStream_t st, MlpParams mp, const uint16_t* w, const uint32_t* feat, const int32_t* feat_index, \
                          int64_t feat_stride, const uint8_t* sel, const float* dout, float2* dfeat, float* partials, float* level_absmax,  \
                          int64_t n, const int64_t* n_dev
void mlp_fwd_bf16(PERF_MLP_FWD_ARGS);
void mlp_fwd_fp16(PERF_MLP_FWD_ARGS);
void mlp_fused_bf16(PERF_MLP_FUSED_ARGS);
void mlp_fused_fp16(PERF_MLP_FUSED_ARGS);
void mlp_bwd_bf16_nh1(PERF_MLP_BWD_ARGS);
void mlp_bwd_bf16_nh2(PERF_MLP_BWD_ARGS);
void mlp_bwd_fp16_nh1(PERF_MLP_BWD_ARGS);
void mlp_bwd_fp16_nh2(PERF_MLP_BWD_ARGS);

}  // namespace perf
