// Streaming helpers: parameter cast, fused Adam, sample positions, panorama ray generation,
// occupancy bit packing and the occupancy pre-grid splat.  All are one-pass HBM-bound kernels with
// fully coalesced accesses.
#include <stdarg.h>
#include <stdio.h>
#include "common.hpp"
#include "step_book_device.hpp"

namespace perf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

template <typename T16>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 v = *reinterpret_cast<const float4*>(src + i);
        uint2 o = make_uint2(T16::pack(v.x, v.y), T16::pack(v.z, v.w));
        *reinterpret_cast<uint2*>(dst + i) = o;
    } else {
        for (; i < n; ++i) dst[i] = T16::one(src[i]);
    }
}

// torch.optim.Adam (foreach form): m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
template <typename T16, bool W16>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ g, uint16_t* __restrict__ w16, int64_t n,
                                                   float one_minus_b1, float b2, float one_minus_b2, float step_size,
                                                   float inv_bc2_sqrt, float eps, int zero_grad,
                                                   const int32_t* __restrict__ step_dev, const float* __restrict__ lr_dev,
                                                   const int64_t* __restrict__ gate_dev, int32_t* __restrict__ clear_flag) {
    __shared__ float sc[2];
    if (clear_flag && blockIdx.x == 0 && threadIdx.x == 0) clear_flag[0] = 0;      // (taken or not: perf_adam_step_dev)
    if (gate_dev && gate_dev[0] <= 0) return;       // batch without samples: the reference skips the step (nerf.py:204-206)
    if (step_dev) {       // step count and learning rate live on the device (hipGraph replay): derive the scalars here
        if (threadIdx.x == 0) {
            const float t = (float)step_dev[0];
            const float bc1 = 1.0f - powf(1.0f - one_minus_b1, t), bc2 = 1.0f - powf(b2, t);
            sc[0] = lr_dev[0] / bc1;
            sc[1] = 1.0f / sqrtf(bc2);
        }
        __syncthreads();
        step_size = sc[0];
        inv_bc2_sqrt = sc[1];
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float mi = m[i], vi = v[i], pi = p[i];
    mi = mi + one_minus_b1 * (gi - mi);
    vi = vi * b2 + one_minus_b2 * gi * gi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    m[i] = mi; v[i] = vi; p[i] = pi;
    if (W16) w16[i] = T16::one(pi);
    if (zero_grad) g[i] = 0.f;
}

// The same step, four consecutive elements per thread (16-byte loads and stores) when the arrays allow it, and with the loads
// issued BEFORE the scalar prologue (gate, step count -> bias corrections, learning rate: a dependent chain of device reads and
// two powf that every workgroup repeats): the 199 MB of a 3.3 M-entry table's step stream while it runs.  Element for element
// the arithmetic of adam_kernel.
template <typename T16, bool W16>
__global__ __launch_bounds__(256) void adam4_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                    float* __restrict__ g, uint16_t* __restrict__ w16, int64_t n,
                                                    float one_minus_b1, float b2, float one_minus_b2, float step_size,
                                                    float inv_bc2_sqrt, float eps, int zero_grad,
                                                    const int32_t* __restrict__ step_dev, const float* __restrict__ lr_dev,
                                                    const int64_t* __restrict__ gate_dev, int32_t* __restrict__ clear_flag) {
    __shared__ float sc[2];
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int cnt = i + 3 < n ? 4 : (i < n ? (int)(n - i) : 0);
    float gi[4] = {0.f, 0.f, 0.f, 0.f}, mi[4] = {0.f, 0.f, 0.f, 0.f}, vi[4] = {0.f, 0.f, 0.f, 0.f}, pi[4] = {0.f, 0.f, 0.f, 0.f};
    if (cnt == 4) {
        const float4 a = *reinterpret_cast<const float4*>(g + i), b = *reinterpret_cast<const float4*>(m + i),
                     c = *reinterpret_cast<const float4*>(v + i), d = *reinterpret_cast<const float4*>(p + i);
        gi[0] = a.x; gi[1] = a.y; gi[2] = a.z; gi[3] = a.w; mi[0] = b.x; mi[1] = b.y; mi[2] = b.z; mi[3] = b.w;
        vi[0] = c.x; vi[1] = c.y; vi[2] = c.z; vi[3] = c.w; pi[0] = d.x; pi[1] = d.y; pi[2] = d.z; pi[3] = d.w;
    } else {
        for (int k = 0; k < cnt; ++k) { gi[k] = g[i + k]; mi[k] = m[i + k]; vi[k] = v[i + k]; pi[k] = p[i + k]; }
    }
    if (clear_flag && blockIdx.x == 0 && threadIdx.x == 0) clear_flag[0] = 0;      // (taken or not: perf_adam_step_dev)
    if (gate_dev && gate_dev[0] <= 0) return;       // batch without samples: the reference skips the step (nerf.py:204-206)
    if (step_dev) {
        if (threadIdx.x == 0) {
            const float t = (float)step_dev[0];
            const float bc1 = 1.0f - powf(1.0f - one_minus_b1, t), bc2 = 1.0f - powf(b2, t);
            sc[0] = lr_dev[0] / bc1;
            sc[1] = 1.0f / sqrtf(bc2);
        }
        __syncthreads();
        step_size = sc[0];
        inv_bc2_sqrt = sc[1];
    }
    if (cnt == 0) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mi[k] = mi[k] + one_minus_b1 * (gi[k] - mi[k]);
        vi[k] = vi[k] * b2 + one_minus_b2 * gi[k] * gi[k];
        const float denom = sqrtf(vi[k]) * inv_bc2_sqrt + eps;
        pi[k] = pi[k] - step_size * (mi[k] / denom);
    }
    if (cnt == 4) {
        *reinterpret_cast<float4*>(m + i) = make_float4(mi[0], mi[1], mi[2], mi[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vi[0], vi[1], vi[2], vi[3]);
        *reinterpret_cast<float4*>(p + i) = make_float4(pi[0], pi[1], pi[2], pi[3]);
        if (W16) *reinterpret_cast<uint2*>(w16 + i) = make_uint2(T16::pack(pi[0], pi[1]), T16::pack(pi[2], pi[3]));
        if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int k = 0; k < cnt; ++k) {
            m[i + k] = mi[k]; v[i + k] = vi[k]; p[i + k] = pi[k];
            if (W16) w16[i + k] = T16::one(pi[k]);
            if (zero_grad) g[i + k] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void points_from_rays_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                               const int64_t* __restrict__ ri, const float* __restrict__ ts,
                                                               const float* __restrict__ te, Aabb bb, float* __restrict__ x01,
                                                               uint8_t* __restrict__ sel, int64_t n,
                                                               const int64_t* __restrict__ n_dev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    const int64_t r = ri[i];
    sample_point_store(o + 3 * r, d + 3 * r, ts[i], te[i], bb, x01, sel, i);
}

__global__ __launch_bounds__(256) void points_normalize_kernel(const float* __restrict__ x, Aabb bb, float* __restrict__ x01,
                                                               uint8_t* __restrict__ sel, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    normalize_store(x[3 * i], x[3 * i + 1], x[3 * i + 2], bb, x01, sel, i);
}

struct RayGen {
    float pose[16];
    float i_start, i_end, i_step, j_start, j_end, j_step;
    int32_t height, width, row0, nrows;
};

// torch.linspace(start, end, n)[idx] in fp32: symmetric evaluation around the midpoint
__device__ __forceinline__ float linspace_at(float start, float end, float step, int n, int idx) {
    return (idx < n / 2) ? start + step * (float)idx : end - step * (float)(n - 1 - idx);
}

__global__ __launch_bounds__(256) void pano_raygen_kernel(RayGen rg, const float* __restrict__ pose_dev, float* __restrict__ ro,
                                                          float* __restrict__ rd) {
    if (pose_dev) {                 // pose read from device memory (a captured hipGraph is replayed with new poses)
#pragma unroll
        for (int k = 0; k < 12; ++k) rg.pose[k] = pose_dev[k];
    }
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)rg.nrows * rg.width;
    if (t >= total) return;
    const int i = rg.row0 + (int)(t / rg.width), j = (int)(t % rg.width);
    const float y = linspace_at(rg.i_start, rg.i_end, rg.i_step, rg.height, i);
    const float x = linspace_at(rg.j_start, rg.j_end, rg.j_step, rg.width, j);
    const float kPi = 3.14159274101257324f;
    const float beta = -(y - 0.5f) * kPi;
    const float alpha = (-(x - 0.5f) * 2.0f) * kPi;
    const float cb = cosf(beta), sb = sinf(beta), ca = cosf(alpha), sa = sinf(alpha);
    const float dx = ca * cb, dy = sa * cb, dz = sb;
    const float* P = rg.pose;
    rd[3 * t] = P[0] * dx + P[1] * dy + P[2] * dz;
    rd[3 * t + 1] = P[4] * dx + P[5] * dy + P[6] * dz;
    rd[3 * t + 2] = P[8] * dx + P[9] * dy + P[10] * dz;
    ro[3 * t] = P[3]; ro[3 * t + 1] = P[7]; ro[3 * t + 2] = P[11];
}

__global__ __launch_bounds__(256) void occ_pack_kernel(const uint8_t* __restrict__ b, uint32_t* __restrict__ bits, int64_t n_cells) {
    const int64_t wi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (wi * 32 >= n_cells) return;
    uint32_t word = 0;
    if (wi * 32 + 32 <= n_cells && (reinterpret_cast<uintptr_t>(b) & 15) == 0) {      // 32 cells = two 16-byte loads
        const uint4 lo = *reinterpret_cast<const uint4*>(b + wi * 32), hi = *reinterpret_cast<const uint4*>(b + wi * 32 + 16);
        const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) word |= ((q[k] >> (8 * j)) & 0xffu) ? 1u << (4 * k + j) : 0u;
    } else {
        for (int k = 0; k < 32; ++k) {
            const int64_t c = wi * 32 + k;
            if (c < n_cells && b[c]) word |= 1u << k;
        }
    }
    bits[wi] = word;
}

// SupInfoPool.gen_occ_grid: 27 shifted copies of p = o + d*dist, index of
// int64((clip(p+shift, +-.999)*.5+.5)*res), x-major.
__global__ __launch_bounds__(256) void occ_splat_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                        const float* __restrict__ dist, int64_t n, int32_t res, float shift,
                                                        uint8_t* __restrict__ occ) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = dist[i];
    float p[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = add_rn(o[3 * i + a], mul_rn(d[3 * i + a], t));
    const float sh[3] = {-shift, 0.f, shift};
    const float rf = (float)res;
    int64_t c[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = add_rn(sh[k], p[a]);
            v = fminf(fmaxf(v, -0.999f), 0.999f);
            v = mul_rn(add_rn(mul_rn(v, 0.5f), 0.5f), rf);
            c[a][k] = (int64_t)v;
        }
    const int64_t r = res;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kz = 0; kz < 3; ++kz) occ[c[0][kx] * r * r + c[1][ky] * r + c[2][kz]] = 1;
}

}  // namespace perf

using namespace perf;

extern "C" int perf_version(void) { return PERF_ABI_VERSION; }
extern "C" int64_t perf_sizeof_grid_desc(void) { return (int64_t)sizeof(perf_grid_desc); }
extern "C" int64_t perf_sizeof_mlp_desc(void) { return (int64_t)sizeof(perf_mlp_desc); }
extern "C" const char* perf_last_error(void) { return g_err; }

extern "C" int perf_cast_params(const float* src, void* dst16, int64_t n, int dtype, void* stream) {
    PERF_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(src && dst16, "NULL pointer");
    PERF_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst16) & 7) == 0,
                 "perf_cast_params: src must be 16-byte and dst 8-byte aligned");
    dim3 g((unsigned)div_up(div_up(n, 4), 256)), b(256);
    if (dtype == PERF_DTYPE_BF16) hipLaunchKernelGGL(cast_kernel<BF16>, g, b, 0, as_stream(stream), src, (uint16_t*)dst16, n);
    else if (dtype == PERF_DTYPE_FP16) hipLaunchKernelGGL(cast_kernel<FP16>, g, b, 0, as_stream(stream), src, (uint16_t*)dst16, n);
    else { set_error("perf_cast_params: bad dtype %d", dtype); return PERF_E_INVALID; }
    PERF_LAUNCH_CHECK("perf_cast_params");
    return PERF_OK;
}

static int adam_launch(float* p, float* m, float* v, float* g, void* w16, int64_t n, int dtype, int32_t step,
                       float lr, float beta1, float beta2, float eps, int zero_grad, const int32_t* step_dev,
                       const float* lr_dev, const int64_t* gate_dev, int32_t* clear_flag, void* stream);

extern "C" int perf_adam_step(float* p, float* m, float* v, float* g, void* w16, int64_t n, int dtype, int32_t step,
                              float lr, float beta1, float beta2, float eps, int zero_grad, void* stream) {
    PERF_REQUIRE(step >= 1, "perf_adam_step: step < 1");
    return adam_launch(p, m, v, g, w16, n, dtype, step, lr, beta1, beta2, eps, zero_grad, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int perf_adam_step_dev(float* p, float* m, float* v, float* g, void* w16, int64_t n, int dtype,
                                  const int32_t* step_dev, const float* lr_dev, const int64_t* gate_dev, float beta1,
                                  float beta2, float eps, int zero_grad, int32_t* clear_flag, void* stream) {
    PERF_REQUIRE(step_dev && lr_dev, "perf_adam_step_dev: NULL scalar pointers");
    PERF_REQUIRE(!clear_flag || n > 0, "perf_adam_step_dev: clear_flag needs a launch (n > 0)");
    return adam_launch(p, m, v, g, w16, n, dtype, 1, 0.f, beta1, beta2, eps, zero_grad, step_dev, lr_dev, gate_dev, clear_flag, stream);
}

static int adam_launch(float* p, float* m, float* v, float* g, void* w16, int64_t n, int dtype, int32_t step,
                       float lr, float beta1, float beta2, float eps, int zero_grad, const int32_t* step_dev,
                       const float* lr_dev, const int64_t* gate_dev, int32_t* clear_flag, void* stream) {
    PERF_REQUIRE(n >= 0, "perf_adam_step: n < 0");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(p && m && v && g, "NULL pointer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    dim3 gr((unsigned)div_up(n, 256)), b(256);
    const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    const uintptr_t align = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                            reinterpret_cast<uintptr_t>(g) | (reinterpret_cast<uintptr_t>(w16) << 1);
    if ((align & 15) == 0 && n >= 1024) {       // four elements per thread
        dim3 g4((unsigned)div_up(div_up(n, 4), 256));
        if (!w16) hipLaunchKernelGGL((adam4_kernel<BF16, false>), g4, b, 0, as_stream(stream), p, m, v, g, (uint16_t*)nullptr, n, omb1, beta2, omb2, step_size, inv_bc2_sqrt, eps, zero_grad, step_dev, lr_dev, gate_dev, clear_flag);
        else if (dtype == PERF_DTYPE_BF16) hipLaunchKernelGGL((adam4_kernel<BF16, true>), g4, b, 0, as_stream(stream), p, m, v, g, (uint16_t*)w16, n, omb1, beta2, omb2, step_size, inv_bc2_sqrt, eps, zero_grad, step_dev, lr_dev, gate_dev, clear_flag);
        else if (dtype == PERF_DTYPE_FP16) hipLaunchKernelGGL((adam4_kernel<FP16, true>), g4, b, 0, as_stream(stream), p, m, v, g, (uint16_t*)w16, n, omb1, beta2, omb2, step_size, inv_bc2_sqrt, eps, zero_grad, step_dev, lr_dev, gate_dev, clear_flag);
        else { set_error("perf_adam_step: bad dtype %d", dtype); return PERF_E_INVALID; }
        PERF_LAUNCH_CHECK("perf_adam_step");
        return PERF_OK;
    }
    if (!w16) hipLaunchKernelGGL((adam_kernel<BF16, false>), gr, b, 0, as_stream(stream), p, m, v, g, (uint16_t*)nullptr, n, omb1, beta2, omb2, step_size, inv_bc2_sqrt, eps, zero_grad, step_dev, lr_dev, gate_dev, clear_flag);
    else if (dtype == PERF_DTYPE_BF16) hipLaunchKernelGGL((adam_kernel<BF16, true>), gr, b, 0, as_stream(stream), p, m, v, g, (uint16_t*)w16, n, omb1, beta2, omb2, step_size, inv_bc2_sqrt, eps, zero_grad, step_dev, lr_dev, gate_dev, clear_flag);
    else if (dtype == PERF_DTYPE_FP16) hipLaunchKernelGGL((adam_kernel<FP16, true>), gr, b, 0, as_stream(stream), p, m, v, g, (uint16_t*)w16, n, omb1, beta2, omb2, step_size, inv_bc2_sqrt, eps, zero_grad, step_dev, lr_dev, gate_dev, clear_flag);
    else { set_error("perf_adam_step: bad dtype %d", dtype); return PERF_E_INVALID; }
    PERF_LAUNCH_CHECK("perf_adam_step");
    return PERF_OK;
}

static Aabb make_aabb(const float* a) {
    Aabb bb;
    for (int i = 0; i < 3; ++i) { bb.lo[i] = a[i]; bb.hi[i] = a[3 + i]; }
    return bb;
}

extern "C" int perf_points_from_rays(const float* rays_o, const float* rays_d, const int64_t* ray_indices,
                                     const float* t_starts, const float* t_ends, const float* aabb, float* x01,
                                     uint8_t* sel, int64_t n, const int64_t* n_dev, void* stream) {
    PERF_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(rays_o && rays_d && ray_indices && t_starts && t_ends && aabb && x01, "NULL pointer");
    hipLaunchKernelGGL(points_from_rays_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, as_stream(stream), rays_o,
                       rays_d, ray_indices, t_starts, t_ends, make_aabb(aabb), x01, sel, n, n_dev);
    PERF_LAUNCH_CHECK("perf_points_from_rays");
    return PERF_OK;
}

extern "C" int perf_points_normalize(const float* x, const float* aabb, float* x01, uint8_t* sel, int64_t n, void* stream) {
    PERF_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x && aabb && x01, "NULL pointer");
    hipLaunchKernelGGL(points_normalize_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, as_stream(stream), x,
                       make_aabb(aabb), x01, sel, n);
    PERF_LAUNCH_CHECK("perf_points_normalize");
    return PERF_OK;
}

static int raygen_launch(const float* pose, const float* pose_dev, int32_t height, int32_t width, int32_t row0, int32_t nrows,
                         float* rays_o, float* rays_d, void* stream);

extern "C" int perf_pano_raygen(const float* pose, int32_t height, int32_t width, int32_t row0, int32_t nrows,
                                float* rays_o, float* rays_d, void* stream) {
    PERF_REQUIRE(pose, "NULL pointer");
    return raygen_launch(pose, nullptr, height, width, row0, nrows, rays_o, rays_d, stream);
}

extern "C" int perf_pano_raygen_dev(const float* pose_dev, int32_t height, int32_t width, int32_t row0, int32_t nrows,
                                    float* rays_o, float* rays_d, void* stream) {
    PERF_REQUIRE(pose_dev, "NULL pointer");
    return raygen_launch(nullptr, pose_dev, height, width, row0, nrows, rays_o, rays_d, stream);
}

static int raygen_launch(const float* pose, const float* pose_dev, int32_t height, int32_t width, int32_t row0, int32_t nrows,
                         float* rays_o, float* rays_d, void* stream) {
    PERF_REQUIRE(rays_o && rays_d, "NULL pointer");
    PERF_REQUIRE(height >= 2 && width >= 2 && row0 >= 0 && nrows >= 0 && row0 + nrows <= height, "bad panorama shape");
    if (nrows == 0) return PERF_OK;
    RayGen rg;
    for (int i = 0; i < 16; ++i) rg.pose[i] = pose ? pose[i] : 0.f;
    rg.i_start = (float)(.5 / height); rg.i_end = (float)(1. - .5 / height);
    rg.j_start = (float)(.5 / width); rg.j_end = (float)(1. - .5 / width);
    rg.i_step = (rg.i_end - rg.i_start) / (float)(height - 1);
    rg.j_step = (rg.j_end - rg.j_start) / (float)(width - 1);
    rg.height = height; rg.width = width; rg.row0 = row0; rg.nrows = nrows;
    const int64_t total = (int64_t)nrows * width;
    hipLaunchKernelGGL(pano_raygen_kernel, dim3((unsigned)div_up(total, 256)), dim3(256), 0, as_stream(stream), rg, pose_dev, rays_o, rays_d);
    PERF_LAUNCH_CHECK("perf_pano_raygen");
    return PERF_OK;
}

// one thread: the bookkeeping of a sync-free training step (see perf_step_bookkeeping in the header; step_book_device.hpp)
namespace perf {
__global__ void step_bookkeeping_kernel(perf_step_book b) { step_bookkeeping_thread(b, true); }
}  // namespace perf

extern "C" int64_t perf_sizeof_step_book(void) { return (int64_t)sizeof(perf_step_book); }

extern "C" int perf_step_bookkeeping(int32_t* step_dev, const int64_t* gate_dev, int64_t* counters,
                                     const int64_t* n_marched_dev, const int64_t* n_kept_dev, int64_t capacity,
                                     int32_t* overflow_flag, const float* remote_flags, int32_t overflow_redone, int64_t* eff_gate_out,
                                     const float* schedule, int32_t n_schedule, int32_t* iter_dev, float* lr_out, float* ratio_out,
                                     void* stream) {
    perf_step_book b{step_dev, gate_dev, counters, n_marched_dev, n_kept_dev, capacity, overflow_flag, remote_flags, eff_gate_out, schedule, iter_dev,
                     lr_out, ratio_out, n_schedule, overflow_redone};
    hipLaunchKernelGGL(perf::step_bookkeeping_kernel, dim3(1), dim3(1), 0, as_stream(stream), b);
    PERF_LAUNCH_CHECK("perf_step_bookkeeping");
    return PERF_OK;
}

extern "C" int perf_occ_pack_bits(const uint8_t* binaries, uint32_t* bits, int64_t n_cells, void* stream) {
    PERF_REQUIRE(binaries && bits && n_cells > 0, "bad arguments");
    hipLaunchKernelGGL(occ_pack_kernel, dim3((unsigned)div_up(div_up(n_cells, 32), 256)), dim3(256), 0, as_stream(stream),
                       binaries, bits, n_cells);
    PERF_LAUNCH_CHECK("perf_occ_pack_bits");
    return PERF_OK;
}

extern "C" int perf_occ_splat(const float* rays_o, const float* rays_d, const float* dist, int64_t n, int32_t res,
                              uint8_t* occ, void* stream) {
    PERF_REQUIRE(n >= 0 && res > 0, "bad arguments");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(rays_o && rays_d && dist && occ, "NULL pointer");
    const float shift = (float)(1.0 / res);
    hipLaunchKernelGGL(occ_splat_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, as_stream(stream), rays_o, rays_d,
                       dist, n, res, shift, occ);
    PERF_LAUNCH_CHECK("perf_occ_splat");
    return PERF_OK;
}

// ---- supervision batch ------------------------------------------------------------------------------------------
// rand_ray_color_data (modules/dataset/sup_info.py:236-259) gathers origins, directions, colours, distances and normals of
// the drawn pixels with five indexing kernels; here one launch gathers whatever is asked for (NULL = skip).
namespace perf {
// The rows of pixel j -> slot i of whatever outputs are asked for.  ALL the loads are issued before the first store: written as
// "if (o) o[..] = o_all[..]" per element, every load is waited for on its own before the store that follows it -- thirteen
// dependent round trips to random rows for what is one (11 us -> the latency of one gather, measured on the draw kernel).
__device__ __forceinline__ void gather_rows(int64_t i, int64_t j, const float* __restrict__ o_all, const float* __restrict__ d_all,
                                            const float* __restrict__ c_all, const float* __restrict__ t_all,
                                            const float* __restrict__ n_all, float* __restrict__ o, float* __restrict__ d,
                                            float* __restrict__ c, float* __restrict__ t, float* __restrict__ nrm) {
    float vo[3] = {0.f, 0.f, 0.f}, vd[3] = {0.f, 0.f, 0.f}, vc[3] = {0.f, 0.f, 0.f}, vn[3] = {0.f, 0.f, 0.f}, vt = 0.f;
    if (o) { vo[0] = o_all[3 * j]; vo[1] = o_all[3 * j + 1]; vo[2] = o_all[3 * j + 2]; }
    if (d) { vd[0] = d_all[3 * j]; vd[1] = d_all[3 * j + 1]; vd[2] = d_all[3 * j + 2]; }
    if (c) { vc[0] = c_all[3 * j]; vc[1] = c_all[3 * j + 1]; vc[2] = c_all[3 * j + 2]; }
    if (nrm) { vn[0] = n_all[3 * j]; vn[1] = n_all[3 * j + 1]; vn[2] = n_all[3 * j + 2]; }
    if (t) vt = t_all[j];
    if (o) { o[3 * i] = vo[0]; o[3 * i + 1] = vo[1]; o[3 * i + 2] = vo[2]; }
    if (d) { d[3 * i] = vd[0]; d[3 * i + 1] = vd[1]; d[3 * i + 2] = vd[2]; }
    if (c) { c[3 * i] = vc[0]; c[3 * i + 1] = vc[1]; c[3 * i + 2] = vc[2]; }
    if (nrm) { nrm[3 * i] = vn[0]; nrm[3 * i + 1] = vn[1]; nrm[3 * i + 2] = vn[2]; }
    if (t) t[i] = vt;
}

__global__ __launch_bounds__(256) void gather_supervision_kernel(const int64_t* __restrict__ idx, int64_t n,
                                                                 const float* __restrict__ o_all, const float* __restrict__ d_all,
                                                                 const float* __restrict__ c_all, const float* __restrict__ t_all,
                                                                 const float* __restrict__ n_all, float* __restrict__ o,
                                                                 float* __restrict__ d, float* __restrict__ c, float* __restrict__ t,
                                                                 float* __restrict__ nrm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    gather_rows(i, idx[i], o_all, d_all, c_all, t_all, n_all, o, d, c, t, nrm);
}
}  // namespace perf

extern "C" int perf_gather_supervision(const int64_t* indices, int64_t n, const float* o_all, const float* d_all,
                                       const float* color_all, const float* dist_all, const float* normal_all, float* o,
                                       float* d, float* color, float* dist, float* normal, void* stream) {
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(indices, "NULL pointer");
    PERF_REQUIRE((!o || o_all) && (!d || d_all) && (!color || color_all) && (!dist || dist_all) && (!normal || normal_all),
                 "perf_gather_supervision: output without source");
    hipLaunchKernelGGL(perf::gather_supervision_kernel, dim3((unsigned)perf::div_up(n, 256)), dim3(256), 0, perf::as_stream(stream),
                       indices, n, o_all, d_all, color_all, dist_all, normal_all, o, d, color, dist, normal);
    PERF_LAUNCH_CHECK("perf_gather_supervision");
    return PERF_OK;
}

// ---- training batch draw: index stream + per-ray uniforms + supervision gather in one launch --------------------------------
namespace perf {
// Philox4x32-10 (Salmon et al., SC'11; the counter-based generator family torch.rand uses on the GPU): 128-bit counter,
// 64-bit key -> four 32-bit words.
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x & 0xffffffu) * (1.0f / 16777216.0f); }   // 24 bits, [0, 1)

__global__ __launch_bounds__(256) void draw_train_batch_kernel(uint32_t seed_lo, uint32_t seed_hi, int64_t* __restrict__ counter,
                                                               int32_t* __restrict__ ticket, int64_t pool_lo, int64_t pool_n,
                                                               int64_t n_local, int64_t first_global,
                                                               const float* __restrict__ o_all, const float* __restrict__ d_all,
                                                               const float* __restrict__ c_all, const float* __restrict__ t_all,
                                                               const float* __restrict__ n_all, float* __restrict__ o,
                                                               float* __restrict__ d, float* __restrict__ c, float* __restrict__ t,
                                                               float* __restrict__ nrm, int64_t* __restrict__ idx_out,
                                                               float* __restrict__ jitter, float* __restrict__ noise,
                                                               float* __restrict__ bg) {
    const uint64_t draw = (uint64_t)counter[0];              // read by every workgroup BEFORE any of them can advance it
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_local) {
        const uint64_t g = (uint64_t)(first_global + i);     // position in the GLOBAL batch: ranks draw disjoint slices of one stream
        uint32_t r[4] = {(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)draw, (uint32_t)(draw >> 32) & 0x7fffffffu};
        philox4x32_10(r, seed_lo, seed_hi);
        const int64_t j = pool_lo + (int64_t)(((uint64_t)r[0] * (uint64_t)pool_n) >> 32);
        gather_rows(i, j, o_all, d_all, c_all, t_all, n_all, o, d, c, t, nrm);     // (first: its loads fly during the second Philox)
        if (idx_out) idx_out[i] = j;
        if (jitter) jitter[i] = u01(r[1]);
        if (noise) noise[i] = u01(r[2]);
        if (bg) {
            uint32_t q[4] = {(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)draw, ((uint32_t)(draw >> 32) & 0x7fffffffu) | 0x80000000u};
            philox4x32_10(q, seed_lo, seed_hi);
            bg[3 * i] = u01(q[0]); bg[3 * i + 1] = u01(q[1]); bg[3 * i + 2] = u01(q[2]);
        }
    }
    // the last workgroup to finish advances the draw counter (every workgroup has read it by then) and rearms the ticket
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
        counter[0] = (int64_t)(draw + 1);
        *ticket = 0;
    }
}
}  // namespace perf

extern "C" int perf_draw_train_batch(uint64_t seed, int64_t* counter_dev, int32_t* ticket, int64_t pool_lo, int64_t pool_hi,
                                     int64_t n_local, int64_t first_global, const float* o_all, const float* d_all,
                                     const float* color_all, const float* dist_all, const float* normal_all, float* o, float* d,
                                     float* color, float* dist, float* normal, int64_t* indices_out, float* jitter, float* noise,
                                     float* bg, void* stream) {
    PERF_REQUIRE(counter_dev && ticket, "perf_draw_train_batch: NULL counter / ticket");
    PERF_REQUIRE(pool_hi > pool_lo && pool_lo >= 0 && pool_hi - pool_lo < ((int64_t)1 << 32), "perf_draw_train_batch: bad pool range");
    PERF_REQUIRE(n_local > 0 && first_global >= 0, "perf_draw_train_batch: bad batch");
    PERF_REQUIRE((!o || o_all) && (!d || d_all) && (!color || color_all) && (!dist || dist_all) && (!normal || normal_all),
                 "perf_draw_train_batch: output without source");
    hipLaunchKernelGGL(perf::draw_train_batch_kernel, dim3((unsigned)perf::div_up(n_local, 256)), dim3(256), 0, perf::as_stream(stream),
                       (uint32_t)seed, (uint32_t)(seed >> 32), counter_dev, ticket, pool_lo, pool_hi - pool_lo, n_local, first_global, o_all,
                       d_all, color_all, dist_all, normal_all, o, d, color, dist, normal, indices_out, jitter, noise, bg);
    PERF_LAUNCH_CHECK("perf_draw_train_batch");
    return PERF_OK;
}


// ---- occupancy-grid update (nerfacc OccGridEstimator.update_every_n_steps; PeRF calls it 256 times per episode with a
//      look-up closure, modules/scene/nerf.py:147-168) -------------------------------------------------------------------------------
// The estimator evaluates every cell at a jittered point, folds the result into an exponential moving maximum and thresholds.
// The closure in the middle is the caller's Python; what surrounds it are two launches instead of ~20 torch element-wise
// passes over 16.7 M x 3 floats: (1) the jittered points of a range of cells, (2) occs = max(occs * decay, occ) with the
// range's sum, (3) -- once per update -- the threshold min(mean(occs), occ_thre) applied to all cells, as bool bytes.
namespace perf {
__global__ __launch_bounds__(256) void occ_jitter_points_kernel(uint32_t seed_lo, uint32_t seed_hi, uint64_t call, int64_t cell_lo,
                                                                int64_t n, int32_t res, Aabb bb, float* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t c = cell_lo + i;
    uint32_t r[4] = {(uint32_t)(c & 0xffffffffll), (uint32_t)(c >> 32), (uint32_t)(call & 0xffffffffull), (uint32_t)(call >> 32) ^ 0x0cc5eedu};
    philox4x32_10(r, seed_lo, seed_hi);
    const int32_t cz = (int32_t)(c % res), cy = (int32_t)((c / res) % res), cx = (int32_t)(c / ((int64_t)res * res));
    const float fr = (float)res;        // (coord + U[0,1)) / res, IEEE division like torch
    const float u[3] = {__fdiv_rn((float)cx + u01(r[0]), fr), __fdiv_rn((float)cy + u01(r[1]), fr), __fdiv_rn((float)cz + u01(r[2]), fr)};
#pragma unroll
    for (int a = 0; a < 3; ++a) x[3 * i + a] = bb.lo[a] + u[a] * (bb.hi[a] - bb.lo[a]);
}

__global__ __launch_bounds__(256) void occ_ema_kernel(float* __restrict__ occs, const float* __restrict__ occ, int64_t n, float decay,
                                                      double* __restrict__ sum_out) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = fmaxf(occs[i] * decay, occ[i]);
        occs[i] = v;
        acc += v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum_out, (double)((part[0] + part[1]) + (part[2] + part[3])));
}

__global__ __launch_bounds__(256) void occ_threshold_kernel(const float* __restrict__ occs, int64_t n, const double* __restrict__ sum,
                                                            float occ_thre, uint8_t* __restrict__ binaries) {
    const float thre = fminf((float)(sum[0] / (double)n), occ_thre);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) binaries[i] = occs[i] > thre ? 1 : 0;
}
}  // namespace perf

extern "C" int perf_occ_jitter_points(uint64_t seed, uint64_t call, int64_t cell_lo, int64_t n, int32_t res, const float* aabb,
                                      float* x, void* stream) {
    PERF_REQUIRE(res > 0 && cell_lo >= 0 && n >= 0 && aabb, "perf_occ_jitter_points: bad arguments");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(x, "NULL pointer");
    perf::Aabb bb;
    for (int a = 0; a < 3; ++a) { bb.lo[a] = aabb[a]; bb.hi[a] = aabb[3 + a]; }
    hipLaunchKernelGGL(perf::occ_jitter_points_kernel, dim3((unsigned)perf::div_up(n, 256)), dim3(256), 0, perf::as_stream(stream),
                       (uint32_t)(seed & 0xffffffffull), (uint32_t)(seed >> 32), call, cell_lo, n, res, bb, x);
    PERF_LAUNCH_CHECK("perf_occ_jitter_points");
    return PERF_OK;
}

extern "C" int perf_occ_ema_update(float* occs, const float* occ, int64_t n, float ema_decay, double* sum_out, void* stream) {
    PERF_REQUIRE(n >= 0 && sum_out, "perf_occ_ema_update: bad arguments");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(occs && occ, "NULL pointer");
    int64_t blocks = perf::div_up(n, 256 * 8);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(perf::occ_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, perf::as_stream(stream), occs, occ, n, ema_decay, sum_out);
    PERF_LAUNCH_CHECK("perf_occ_ema_update");
    return PERF_OK;
}

extern "C" int perf_occ_threshold(const float* occs, int64_t n, const double* sum, float occ_thre, uint8_t* binaries, void* stream) {
    PERF_REQUIRE(n > 0 && occs && sum && binaries, "perf_occ_threshold: bad arguments");
    hipLaunchKernelGGL(perf::occ_threshold_kernel, dim3((unsigned)perf::div_up(n, 256)), dim3(256), 0, perf::as_stream(stream), occs, n, sum,
                       occ_thre, binaries);
    PERF_LAUNCH_CHECK("perf_occ_threshold");
    return PERF_OK;
}
