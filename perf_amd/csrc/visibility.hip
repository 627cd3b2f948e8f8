// Reprojection visibility tests around the renderer: NeRFScene.get_pano_visibility_mask
// (modules/scene/nerf.py:321-358) and SupInfoPool.geo_check (modules/dataset/sup_info.py:261-302).  Per pixel of a rendered
// panorama: back-projected point -> direction and distance in a registered panorama's frame -> equirect image coordinate
// (utils/camera_utils.py:134-151) -> bilinear look-up of that panorama's distance map with grid_sample's border padding /
// align_corners=False arithmetic -> depth test, folded over the panoramas with max (visibility) or min (consistency).
// Then binary morphology with OpenCV's elliptical structuring elements.  One thread per pixel, fully coalesced; the
// reference runs ~15 torch kernels per registered panorama and two kornia convolutions.
#include "common.hpp"

namespace perf {

struct PanoFrame { float rt[9]; float t[3]; };      // rt = R^T (row major), t = camera centre

__global__ __launch_bounds__(256) void pano_reproject_kernel(const float* __restrict__ pts, int64_t n, PanoFrame pf,
                                                             const float* __restrict__ dmap, int32_t h, int32_t w, int32_t mode,
                                                             float eps, float* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = sub_rn(pts[3 * i], pf.t[0]), py = sub_rn(pts[3 * i + 1], pf.t[1]), pz = sub_rn(pts[3 * i + 2], pf.t[2]);
    // apply_rot(p - t, R^T): matmul row by row (utils/camera_utils.py:44-46)
    float l[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) l[a] = add_rn(add_rn(mul_rn(pf.rt[3 * a], px), mul_rn(pf.rt[3 * a + 1], py)), mul_rn(pf.rt[3 * a + 2], pz));
    const float dist = sqrtf(add_rn(add_rn(mul_rn(l[0], l[0]), mul_rn(l[1], l[1])), mul_rn(l[2], l[2])));
    float d[3] = {__fdiv_rn(l[0], dist), __fdiv_rn(l[1], dist), __fdiv_rn(l[2], dist)};
    // direction_to_img_coord normalises once more
    const float nn = sqrtf(add_rn(add_rn(mul_rn(d[0], d[0]), mul_rn(d[1], d[1])), mul_rn(d[2], d[2])));
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = __fdiv_rn(d[a], nn);
    const float kPi = 3.14159265358979323846f;
    const float beta = asinf(d[2]), alpha = atan2f(d[1], d[0]);
    const float row = add_rn(__fdiv_rn(-beta, kPi), 0.5f);
    const float col = add_rn(-__fdiv_rn(alpha, mul_rn(2.0f, kPi)), 0.5f);
    // img_coord_to_sample_coord: x = col*2-1, y = row*2-1; grid_sample (align_corners=False): ix = ((x+1)*W - 1)/2, border = clamp
    const float gx = sub_rn(mul_rn(col, 2.0f), 1.0f), gy = sub_rn(mul_rn(row, 2.0f), 1.0f);
    float ix = __fdiv_rn(sub_rn(mul_rn(add_rn(gx, 1.0f), (float)w), 1.0f), 2.0f);
    float iy = __fdiv_rn(sub_rn(mul_rn(add_rn(gy, 1.0f), (float)h), 1.0f), 2.0f);
    ix = fminf(fmaxf(ix, 0.0f), (float)(w - 1));
    iy = fminf(fmaxf(iy, 0.0f), (float)(h - 1));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = sub_rn(ix, fx0), wy1 = sub_rn(iy, fy0), wx0 = sub_rn(1.0f, wx1), wy0 = sub_rn(1.0f, wy1);
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? dmap[(int64_t)yy * w + xx] : 0.0f; };
    // torch's bilinear accumulation order: nw, ne, sw, se
    float proj = mul_rn(at(y0, x0), mul_rn(wx0, wy0));
    proj = add_rn(proj, mul_rn(at(y0, x1), mul_rn(wx1, wy0)));
    proj = add_rn(proj, mul_rn(at(y1, x0), mul_rn(wx0, wy1)));
    proj = add_rn(proj, mul_rn(at(y1, x1), mul_rn(wx1, wy1)));
    if (mode == 0) {            // visibility: seen by this panorama if not behind what it stored (+ 1/256)
        const float v = (dist < add_rn(proj, eps)) ? 1.0f : 0.0f;
        mask[i] = fmaxf(mask[i], v);
    } else {                    // consistency: the point must lie behind what this panorama stored
        const float v = (proj < dist) ? 1.0f : 0.0f;
        mask[i] = fminf(mask[i], v);
    }
}

struct Ellipse { uint32_t row_bits[16]; int32_t rows, cols; };

// op 0: dilation (outside the image = background), op 1: erosion (outside = foreground), both with the element's reflection
// handled by the caller's row masks; in/out are 0/1 floats
__global__ __launch_bounds__(256) void morph_kernel(const float* __restrict__ in, float* __restrict__ out, int32_t h, int32_t w,
                                                    Ellipse el, int32_t op) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)h * w) return;
    const int y = (int)(i / w), x = (int)(i % w);
    const int ry = el.rows / 2, rx = el.cols / 2;
    bool any = false, all = true;
    for (int dy = 0; dy < el.rows; ++dy)
        for (int dx = 0; dx < el.cols; ++dx) {
            if (!((el.row_bits[dy] >> dx) & 1u)) continue;
            const int yy = y + dy - ry, xx = x + dx - rx;
            const bool inside = yy >= 0 && yy < h && xx >= 0 && xx < w;
            const bool v = inside ? (in[(int64_t)yy * w + xx] > 0.5f) : (op == 1);
            any |= v; all &= v;
        }
    out[i] = (op == 0 ? any : all) ? 1.0f : 0.0f;
}

}  // namespace perf

using namespace perf;

extern "C" int perf_pano_reproject(const float* pts, int64_t n, const float* pose, const float* distance_map, int32_t height,
                                   int32_t width, int32_t mode, float eps, float* mask, void* stream) {
    PERF_REQUIRE(n >= 0 && height >= 1 && width >= 1 && (mode == 0 || mode == 1), "perf_pano_reproject: bad arguments");
    if (n == 0) return PERF_OK;
    PERF_REQUIRE(pts && pose && distance_map && mask, "NULL pointer");
    PanoFrame pf;
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) pf.rt[3 * a + b] = pose[4 * b + a];     // R^T
        pf.t[a] = pose[4 * a + 3];
    }
    hipLaunchKernelGGL(pano_reproject_kernel, dim3((unsigned)div_up(n, 256)), dim3(256), 0, as_stream(stream), pts, n, pf,
                       distance_map, height, width, mode, eps, mask);
    PERF_LAUNCH_CHECK("perf_pano_reproject");
    return PERF_OK;
}

extern "C" int perf_morph_binary(const float* in, float* out, int32_t height, int32_t width, const uint32_t* row_bits,
                                 int32_t rows, int32_t cols, int32_t op, void* stream) {
    PERF_REQUIRE(height >= 1 && width >= 1 && rows >= 1 && rows <= 16 && cols >= 1 && cols <= 32 && (op == 0 || op == 1),
                 "perf_morph_binary: bad arguments");
    PERF_REQUIRE(in && out && row_bits && in != out, "perf_morph_binary: NULL or aliased buffers");
    Ellipse el;
    for (int r = 0; r < 16; ++r) el.row_bits[r] = r < rows ? row_bits[r] : 0u;
    el.rows = rows; el.cols = cols;
    hipLaunchKernelGGL(morph_kernel, dim3((unsigned)div_up((int64_t)height * width, 256)), dim3(256), 0, as_stream(stream), in, out,
                       height, width, el, op);
    PERF_LAUNCH_CHECK("perf_morph_binary");
    return PERF_OK;
}
