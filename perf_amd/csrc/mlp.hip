// C-ABI entry points of the 64-wide MLP (perf_mlp_fwd / perf_mlp_bwd / perf_field_infer).  The kernels live in mlp_device.hpp and are
// instantiated in mlp_fwd_{bf16,fp16}.hip and mlp_bwd_{bf16,fp16}_nh{1,2}.hip.
#include "mlp_device.hpp"
#include "mlp_reduce_device.hpp"

namespace perf {

// dw[i] = sum_k partials[k][i] (fixed order: deterministic), combined through LDS (mlp_reduce_device.hpp)
__global__ __launch_bounds__(256) void mlp_reduce_kernel(MlpReduceJob job) { mlp_reduce_block(job, (int)blockIdx.x); }

}  // namespace perf

using namespace perf;

extern "C" int perf_mlp_fwd(const perf_mlp_desc* mlp, const void* w16, const void* feat16, const uint8_t* sel,
                            float* out, int64_t n, const int64_t* n_dev, int dtype, void* stream) {
    int nh, ks;
    int rc = check_mlp(mlp, &nh, &ks);
    if (rc) return rc;
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(w16 && feat16 && out, "NULL pointer");
    PERF_REQUIRE(dtype == PERF_DTYPE_BF16 || dtype == PERF_DTYPE_FP16, "bad dtype %d", dtype);
    MlpParams mp{mlp->n_levels, mlp->n_out, mlp->out_act, mlp->exp_shift};
    const int blocks = mlp_blocks(n, nh == 1 ? 4 : 3);          // = the waves per SIMD the kernels' registers allow: fragments are staged once per block
    if (dtype == PERF_DTYPE_BF16)
        mlp_fwd_bf16(nh, ks, blocks, as_stream(stream), mp, (const uint16_t*)w16, (const uint32_t*)feat16, sel, out, n, n_dev);
    else
        mlp_fwd_fp16(nh, ks, blocks, as_stream(stream), mp, (const uint16_t*)w16, (const uint32_t*)feat16, sel, out, n, n_dev);
    PERF_LAUNCH_CHECK("perf_mlp_fwd");
    return PERF_OK;
}

extern "C" int64_t perf_mlp_bwd_workspace_bytes(const perf_mlp_desc* mlp, int64_t n) {
    int nh, ks;
    if (check_mlp(mlp, &nh, &ks)) return -1;
    const int blocks = mlp_blocks(n > 0 ? n : 1, bwd_blocks_per_cu(nh, ks));
    return ((int64_t)blocks * n_params_rt(nh, ks) + (int64_t)blocks * 8) * (int64_t)sizeof(float);
}

void perf_internal_launch_mlp_reduce(const MlpReduceJob& job, void* stream) {
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)job.n_blocks), dim3(256), 0, as_stream(stream), job);
}

extern "C" int perf_mlp_bwd(const perf_mlp_desc* mlp, const void* w16, const void* feat16, const int32_t* feat_index, int64_t feat_stride,
                            const uint8_t* sel, const float* dout, float* dfeat, float* dw, float* level_absmax, void* workspace,
                            int64_t workspace_bytes, int64_t n, const int64_t* n_dev, int dtype, void* stream) {
    return perf_internal_mlp_bwd(mlp, w16, feat16, feat_index, feat_stride, sel, dout, dfeat, dw, level_absmax, workspace, workspace_bytes, n, n_dev,
                                 dtype, stream, nullptr);
}

// defer != NULL: the second stage is NOT launched; *defer describes it (n_blocks == 0 when there is nothing to reduce) and the caller
// lets it ride in a later launch of the same stream (perf_field_bwd: the tile-code pre-pass of the grid backward)
int perf_internal_mlp_bwd(const perf_mlp_desc* mlp, const void* w16, const void* feat16, const int32_t* feat_index, int64_t feat_stride,
                          const uint8_t* sel, const float* dout, float* dfeat, float* dw, float* level_absmax, void* workspace,
                          int64_t workspace_bytes, int64_t n, const int64_t* n_dev, int dtype, void* stream, MlpReduceJob* defer) {
    if (defer) defer->n_blocks = 0;
    int nh, ks;
    int rc = check_mlp(mlp, &nh, &ks);
    if (rc) return rc;
    PERF_REQUIRE(w16 && dw && workspace, "NULL pointer");
    PERF_REQUIRE(dtype == PERF_DTYPE_BF16 || dtype == PERF_DTYPE_FP16, "bad dtype %d", dtype);
    const int np = n_params_rt(nh, ks);
    if (n == 0) {
        if (level_absmax) (void)hipMemsetAsync(level_absmax, 0, PERF_MAX_LEVELS * sizeof(float), as_stream(stream));
        hipError_t e = hipMemsetAsync(dw, 0, np * sizeof(float), as_stream(stream));
        if (e != hipSuccess) { set_error("perf_mlp_bwd: memset failed"); return PERF_E_LAUNCH; }
        return PERF_OK;
    }
    PERF_REQUIRE(feat16 && dout, "NULL pointer");
    PERF_REQUIRE(feat_index == nullptr || feat_stride > 0, "perf_mlp_bwd: feat_index needs feat_stride > 0");
    const int64_t need = perf_mlp_bwd_workspace_bytes(mlp, n);
    PERF_REQUIRE(workspace_bytes >= need, "perf_mlp_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    MlpParams mp{mlp->n_levels, mlp->n_out, mlp->out_act, mlp->exp_shift};
    const int blocks = mlp_blocks(n, bwd_blocks_per_cu(nh, ks));
    float* amax_slots = level_absmax ? (float*)workspace + (int64_t)blocks * np : nullptr;
    if (dtype == PERF_DTYPE_BF16)
        (nh == 1 ? mlp_bwd_bf16_nh1 : mlp_bwd_bf16_nh2)(ks, blocks, as_stream(stream), mp, (const uint16_t*)w16, (const uint32_t*)feat16, feat_index, feat_stride, sel,
                           dout, (float2*)dfeat, (float*)workspace, amax_slots, n, n_dev);
    else
        (nh == 1 ? mlp_bwd_fp16_nh1 : mlp_bwd_fp16_nh2)(ks, blocks, as_stream(stream), mp, (const uint16_t*)w16, (const uint32_t*)feat16, feat_index, feat_stride, sel,
                           dout, (float2*)dfeat, (float*)workspace, amax_slots, n, n_dev);
    PERF_LAUNCH_CHECK("perf_mlp_bwd");
    MlpReduceJob job{(const float*)workspace, dw, (const float*)amax_slots, level_absmax, np, blocks, (int32_t)mlp->n_levels, (int32_t)div_up(np, 16) + 1};
    if (defer) { *defer = job; return PERF_OK; }
    perf_internal_launch_mlp_reduce(job, stream);
    PERF_LAUNCH_CHECK("perf_mlp_bwd(reduce)");
    return PERF_OK;
}

// ---- encode + MLP as ONE boundary call (SURVEY.md 8(b) perf_field_infer; NGPNeRF.query_density / query_rgb without
// gradient, modules/fields/ngp_nerf.py:136-162).  Two launches inside: the encode is cut by level group and pinned to XCDs
// (the 13 MB table does not fit one XCD's 4 MiB L2; the pinning is worth 1.9-2.4x on the gathers), while the
// MLP needs all 32 features of a sample in one wave -- a single kernel would give up the pinning to save a 64 B/sample
// round trip of 16-bit features through L2 / Infinity Cache.  `scratch` holds those features: 4 * n_levels * n bytes.
extern "C" int64_t perf_field_infer_scratch_bytes(const perf_grid_desc* grid, int64_t n) {
    if (!grid || grid->n_levels < 1 || grid->n_levels > PERF_MAX_LEVELS || n < 0) return -1;
    return (int64_t)grid->n_levels * n * 4;
}

// Batches of up to this many rows run encode + MLP as ONE kernel (features in registers); larger ones keep the level-group
// kernel pinned to XCDs followed by the MLP kernel.  (PERF_FUSED_MAX_SAMPLES overrides, 0 = never.)  Measured on MI355X
// (reference-faithful training episode, tools/train_episode.py): with the 16,384-row head pass fused the geometry step takes
// 0.2514 ms, unfused 0.2461 ms -- at that size a wave's 64 dependent-address gathers per lane cost more than the second
// launch saves; 512x1024 frames in 32,768-ray batches (65,536-row heads): 491 fused vs 489-500 frames/s unfused.  The
// fused kernel is therefore kept for what it cannot lose on: batches of a few thousand points.
constexpr int64_t kFusedMaxSamples = 4096;

extern "C" int perf_field_infer(const perf_grid_desc* grid, const perf_mlp_desc* mlp, const float* x01, const uint8_t* sel,
                                const void* table16, const void* w16, float* out, int64_t n, const int64_t* n_dev,
                                void* scratch, int64_t scratch_bytes, void* feat_out, int dtype, void* stream) {
    PERF_REQUIRE(grid && mlp, "NULL descriptor");
    PERF_REQUIRE(mlp->n_levels == grid->n_levels, "perf_field_infer: the MLP takes %d levels, the grid has %d", (int)mlp->n_levels, (int)grid->n_levels);
    if (n == 0) return PERF_OK;
    static const int64_t fused_max = getenv("PERF_FUSED_MAX_SAMPLES") ? atoll(getenv("PERF_FUSED_MAX_SAMPLES")) : kFusedMaxSamples;
    if (n <= fused_max && grid->n_levels <= 16 && grid->layout == PERF_LAYOUT_TCNN) {
        int nh, ks;
        int rc = check_mlp(mlp, &nh, &ks);
        if (rc) return rc;
        PERF_REQUIRE(x01 && table16 && w16 && out, "NULL pointer");
        PERF_REQUIRE(dtype == PERF_DTYPE_BF16 || dtype == PERF_DTYPE_FP16, "bad dtype %d", dtype);
        FusedIn fz;
        rc = fill_params(grid, &fz.gp);
        if (rc) return rc;
        fz.table = (const uint32_t*)table16; fz.x01 = x01; fz.feat_out = (uint32_t*)feat_out;
        MlpParams mp{mlp->n_levels, mlp->n_out, mlp->out_act, mlp->exp_shift};
        const int blocks = mlp_blocks(n, nh == 1 ? 4 : 3);
        if (dtype == PERF_DTYPE_BF16) mlp_fused_bf16(nh, ks, blocks, as_stream(stream), mp, (const uint16_t*)w16, sel, out, n, n_dev, fz);
        else mlp_fused_fp16(nh, ks, blocks, as_stream(stream), mp, (const uint16_t*)w16, sel, out, n, n_dev, fz);
        PERF_LAUNCH_CHECK("perf_field_infer(fused)");
        return PERF_OK;
    }
    void* feat = feat_out ? feat_out : scratch;
    PERF_REQUIRE(feat_out || (scratch && scratch_bytes >= perf_field_infer_scratch_bytes(grid, n)), "perf_field_infer: scratch too small");
    int rc = perf_hashgrid_fwd(grid, x01, table16, feat, n, n_dev, dtype, stream);
    if (rc) return rc;
    return perf_mlp_fwd(mlp, w16, feat, sel, out, n, n_dev, dtype, stream);
}

// ---- the whole backward of one field as ONE boundary call (what a tcnn.NetworkWithInputEncoding backward is to its caller:
// modules/fields/ngp_nerf.py:142,158 under loss.backward(), modules/scene/nerf.py:252-253): MLP backward (weight gradient, feature
// gradient, per-level max |dfeat|) -> grid backward into the table part of the same flat gradient -> [the predicated fp32 repair
// launch of a fixed-point call].  Nothing new runs on the device -- the three entry points are chained on the stream -- but a
// binding crosses the boundary once per backward instead of three times (the operator-shim step is host bound:
// profiles/r06_shim_step_host.json).  grad: [n_net | 2 * table entries] fp32, overwritten.  level_absmax != NULL selects the packed
// fixed-point accumulation (then overflow_flag / headroom_state as in perf_hashgrid_bwd); redo != 0 appends the repair launch.
extern "C" int64_t perf_field_bwd_workspace_bytes(const perf_grid_desc* grid, const perf_mlp_desc* mlp, int64_t n, int64_t* mlp_ws_bytes,
                                                 int64_t* grid_ws_bytes, int64_t* dfeat_bytes) {
    if (!grid || !mlp || n < 0) return -1;
    const int64_t a = perf_mlp_bwd_workspace_bytes(mlp, n), b = perf_hashgrid_bwd_workspace_bytes(grid, n);
    if (a < 0 || b < 0) return -1;
    const int64_t a16 = (a + 15) & ~(int64_t)15, b16 = (b + 31) & ~(int64_t)15, c = (int64_t)grid->n_levels * n * 8;
    if (mlp_ws_bytes) *mlp_ws_bytes = a16;
    if (grid_ws_bytes) *grid_ws_bytes = b16;
    if (dfeat_bytes) *dfeat_bytes = c;
    return a16 + b16 + ((c + 15) & ~(int64_t)15) + PERF_MAX_LEVELS * (int64_t)sizeof(float);
}

static int field_bwd_chain(const perf_grid_desc* grid, const perf_mlp_desc* mlp, const float* x01, const void* w16_net,
                           const void* feat16, const int32_t* feat_index, int64_t feat_stride, const uint8_t* sel, const float* dout,
                           float* grad, int32_t fixed, int32_t redo, int32_t* overflow_flag, int32_t* headroom_state,
                           void* workspace, int64_t workspace_bytes, int64_t n, const int64_t* n_dev, int dtype, const perf_step_book* book,
                           void* stream);

extern "C" int perf_field_bwd(const perf_grid_desc* grid, const perf_mlp_desc* mlp, const float* x01, const void* w16_net,
                              const void* feat16, const int32_t* feat_index, int64_t feat_stride, const uint8_t* sel, const float* dout,
                              float* grad, int32_t fixed, int32_t redo, int32_t* overflow_flag, int32_t* headroom_state,
                              void* workspace, int64_t workspace_bytes, int64_t n, const int64_t* n_dev, int dtype, void* stream) {
    return field_bwd_chain(grid, mlp, x01, w16_net, feat16, feat_index, feat_stride, sel, dout, grad, fixed, redo, overflow_flag, headroom_state,
                           workspace, workspace_bytes, n, n_dev, dtype, nullptr, stream);
}

// ... and the step's bookkeeping in one thread of the repair launch (one launch per training step fewer; the flag is left to
// perf_adam_step_dev's clear_flag)
extern "C" int perf_field_bwd_book(const perf_grid_desc* grid, const perf_mlp_desc* mlp, const float* x01, const void* w16_net,
                                   const void* feat16, const int32_t* feat_index, int64_t feat_stride, const uint8_t* sel, const float* dout,
                                   float* grad, int32_t* overflow_flag, int32_t* headroom_state, void* workspace, int64_t workspace_bytes,
                                   int64_t n, const int64_t* n_dev, int dtype, const perf_step_book* book, void* stream) {
    PERF_REQUIRE(book && overflow_flag && headroom_state, "perf_field_bwd_book: NULL pointer");
    PERF_REQUIRE(book->overflow_flag == overflow_flag, "perf_field_bwd_book: book->overflow_flag must be the call's overflow_flag");
    PERF_REQUIRE(n > 0, "perf_field_bwd_book: n == 0");
    return field_bwd_chain(grid, mlp, x01, w16_net, feat16, feat_index, feat_stride, sel, dout, grad, 1, 1, overflow_flag, headroom_state,
                           workspace, workspace_bytes, n, n_dev, dtype, book, stream);
}

static int field_bwd_chain(const perf_grid_desc* grid, const perf_mlp_desc* mlp, const float* x01, const void* w16_net,
                           const void* feat16, const int32_t* feat_index, int64_t feat_stride, const uint8_t* sel, const float* dout,
                           float* grad, int32_t fixed, int32_t redo, int32_t* overflow_flag, int32_t* headroom_state,
                           void* workspace, int64_t workspace_bytes, int64_t n, const int64_t* n_dev, int dtype, const perf_step_book* book,
                           void* stream) {
    PERF_REQUIRE(grid && mlp && grad && workspace, "NULL pointer");
    PERF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "perf_field_bwd: the workspace must be 16-byte aligned");
    int nh, ks;
    int rc = check_mlp(mlp, &nh, &ks);
    if (rc) return rc;
    int64_t a16 = 0, b16 = 0, c = 0;
    const int64_t need = perf_field_bwd_workspace_bytes(grid, mlp, n, &a16, &b16, &c);
    PERF_REQUIRE(need >= 0 && workspace_bytes >= need, "perf_field_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)need);
    char* ws = reinterpret_cast<char*>(workspace);
    void* mlp_ws = ws;
    void* grid_ws = ws + a16;
    float* dfeat = reinterpret_cast<float*>(ws + a16 + b16);
    float* amax = reinterpret_cast<float*>(ws + a16 + b16 + ((c + 15) & ~(int64_t)15));
    const int n_net = n_params_rt(nh, ks);
    // (the MLP backward's second stage -- weight-gradient partials, per-level maxima -- rides in the grid backward's tile-code launch)
    MlpReduceJob job;
    rc = perf_internal_mlp_bwd(mlp, w16_net, feat16, feat_index, feat_stride, sel, dout, dfeat, grad, fixed ? amax : nullptr, mlp_ws, a16, n, n_dev, dtype,
                               stream, &job);
    if (rc) return rc;
    rc = perf_internal_hashgrid_bwd(grid, x01, dfeat, grad + n_net, n, n_dev, 0, fixed ? amax : nullptr, fixed ? overflow_flag : nullptr,
                                    fixed ? headroom_state : nullptr, nullptr, 0, nullptr, grid_ws, b16, stream, &job);
    if (rc) return rc;
    if (fixed && redo)
        rc = perf_internal_hashgrid_bwd(grid, x01, dfeat, grad + n_net, n, n_dev, 0, nullptr, nullptr, headroom_state, nullptr, 0, overflow_flag, nullptr, 0,
                                        stream, nullptr, book);
    return rc;
}
