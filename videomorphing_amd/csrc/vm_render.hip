// vm_render.hip -- compositor and result-delivery kernels for gfx950.
//
// k_render:  kernel_render_halfway_image, Algorithm/render.cu:16-60.  The
//   reference samples float4 textures that RenderStage2 re-uploads every frame
//   (UI/RenderWidget.cpp:229-266); here the Poisson-extended canvases stay
//   resident as RGBA8 (uchar -> float is exact, so the taps see the same
//   values) and v/u as pitched float2: 8 B + 8 B + 3 B of compulsory traffic
//   per pixel plus gathered canvas taps served by L2.
// k_upscale: CMatchingThread::update_result + Resize,
//   Algorithm/MatchingThread.cpp:22-136.
#include "vm_internal.h"

namespace {

__device__ __forceinline__ float2 tap2(const float2 *__restrict__ img, int w, int h, int rs,
                                       float x, float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = xb - fi, b = yb - fj;
    fi = fminf(fmaxf(fi, -1.0f), (float)w);
    fj = fminf(fmaxf(fj, -1.0f), (float)h);
    int i0 = (int)fi, j0 = (int)fj;
    int i1 = min(max(i0 + 1, 0), w - 1), j1 = min(max(j0 + 1, 0), h - 1);
    i0 = min(max(i0, 0), w - 1);
    j0 = min(max(j0, 0), h - 1);
    float2 t00 = img[j0 * rs + i0], t10 = img[j0 * rs + i1];
    float2 t01 = img[j1 * rs + i0], t11 = img[j1 * rs + i1];
    float2 r;
    r.x = (1 - a) * (1 - b) * t00.x + a * (1 - b) * t10.x + (1 - a) * b * t01.x + a * b * t11.x;
    r.y = (1 - a) * (1 - b) * t00.y + a * (1 - b) * t10.y + (1 - a) * b * t01.y + a * b * t11.y;
    return r;
}

__device__ __forceinline__ float3 tap_rgb(const uchar4 *__restrict__ img, int w, int h, float x,
                                          float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = xb - fi, b = yb - fj;
    fi = fminf(fmaxf(fi, -1.0f), (float)w);
    fj = fminf(fmaxf(fj, -1.0f), (float)h);
    int i0 = (int)fi, j0 = (int)fj;
    int i1 = min(max(i0 + 1, 0), w - 1), j1 = min(max(j0 + 1, 0), h - 1);
    i0 = min(max(i0, 0), w - 1);
    j0 = min(max(j0, 0), h - 1);
    uchar4 t00 = img[(size_t)j0 * w + i0], t10 = img[(size_t)j0 * w + i1];
    uchar4 t01 = img[(size_t)j1 * w + i0], t11 = img[(size_t)j1 * w + i1];
    float w00 = (1 - a) * (1 - b), w10 = a * (1 - b), w01 = (1 - a) * b, w11 = a * b;
    float3 r;
    r.x = w00 * (float)t00.x + w10 * (float)t10.x + w01 * (float)t01.x + w11 * (float)t11.x;
    r.y = w00 * (float)t00.y + w10 * (float)t10.y + w01 * (float)t01.y + w11 * (float)t11.y;
    r.z = w00 * (float)t00.z + w10 * (float)t10.z + w01 * (float)t01.z + w11 * (float)t11.z;
    return r;
}

__global__ __launch_bounds__(256) void k_render(uint8_t *__restrict__ out, int out_pitch, int w,
                                                int h, int rs, int ex, float color_fa,
                                                float geo_fa, int color_from,
                                                const uchar4 *__restrict__ ext0,
                                                const uchar4 *__restrict__ ext1,
                                                const float2 *__restrict__ vf,
                                                const float2 *__restrict__ uf)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    const float alpha = 0.8f;
    const float s1 = 2 * geo_fa - 1;
    const float s2 = 4 * geo_fa - 4 * geo_fa * geo_fa;
    const float qx = (float)x, qy = (float)y;
    float px = qx, py = qy;
    float2 v = tap2(vf, w, h, rs, px + 0.5f, py + 0.5f);
    float2 u = uf ? tap2(uf, w, h, rs, px + 0.5f, py + 0.5f) : make_float2(0.0f, 0.0f);
    for (int i = 0; i < 20; ++i) {
        px = qx - s1 * v.x - s2 * u.x;
        py = qy - s1 * v.y - s2 * u.y;
        float2 t = tap2(vf, w, h, rs, px + 0.5f, py + 0.5f);
        v.x = alpha * t.x + (1 - alpha) * v.x;
        v.y = alpha * t.y + (1 - alpha) * v.y;
        if (uf) {
            t = tap2(uf, w, h, rs, px + 0.5f, py + 0.5f);
            u.x = alpha * t.x + (1 - alpha) * u.x;
            u.y = alpha * t.y + (1 - alpha) * u.y;
        } else {
            // a zero path stays zero: alpha*0 + (1-alpha)*0
        }
    }
    float3 c0 = tap_rgb(ext0, cw, ch, px - v.x + ex + 0.5f, py - v.y + ex + 0.5f);
    float3 c1 = tap_rgb(ext1, cw, ch, px + v.x + ex + 0.5f, py + v.y + ex + 0.5f);
    double r, g, b;
    if (color_from == 0) {
        r = c0.x + 0.5; g = c0.y + 0.5; b = c0.z + 0.5;
    } else if (color_from == 1) {
        r = c0.x * (1 - color_fa) + c1.x * color_fa + 0.5;
        g = c0.y * (1 - color_fa) + c1.y * color_fa + 0.5;
        b = c0.z * (1 - color_fa) + c1.z * color_fa + 0.5;
    } else {
        r = c1.x + 0.5; g = c1.y + 0.5; b = c1.z + 0.5;
    }
    uint8_t *o = out + (size_t)y * out_pitch + 3 * x;
    o[0] = (uint8_t)r; // make_uchar3: truncation (render.cu:49-56)
    o[1] = (uint8_t)g;
    o[2] = (uint8_t)b;
}

// BiLinear of MatchingThread.cpp:103-136 on the level's v scaled by (rx, ry)
__global__ __launch_bounds__(256) void k_upscale(float2 *__restrict__ dst, int w0, int h0,
                                                 int dpitch, const float2 *__restrict__ v, int w,
                                                 int h, int rs)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w0 || y >= h0)
        return;
    const float rx = (float)w0 / (float)w, ry = (float)h0 / (float)h;
    if (w == w0 && h == h0) {
        float2 s = v[y * rs + x];
        dst[(size_t)y * dpitch + x] = s; // ratio 1: copied unscaled (MatchingThread.cpp:42,55-58)
        return;
    }
    const float fy = (float)((y + 0.5) / h0 * h - 0.5);
    const float fx = (float)((x + 0.5) / w0 * w - 0.5);
    int xi[2] = {(int)floorf(fx), (int)ceilf(fx)};
    int yi[2] = {(int)floorf(fy), (int)ceilf(fy)};
    const float uu = fx - xi[0], vv = fy - yi[0];
    float2 val[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int tx = min(max(xi[i], 0), w - 1), ty = min(max(yi[j], 0), h - 1);
            float2 s = v[ty * rs + tx];
            val[i][j] = make_float2(s.x * rx, s.y * ry);
        }
    float2 r;
    r.x = val[0][0].x * (1 - uu) * (1 - vv) + val[0][1].x * (1 - uu) * vv +
          val[1][0].x * uu * (1 - vv) + val[1][1].x * uu * vv;
    r.y = val[0][0].y * (1 - uu) * (1 - vv) + val[0][1].y * (1 - uu) * vv +
          val[1][0].y * uu * (1 - vv) + val[1][1].y * uu * vv;
    dst[(size_t)y * dpitch + x] = r;
}

// the frames the temporal pyramid skipped: _vector[beg] * (1 - fa) + _vector[end] * fa
// (MatchingThread.cpp:62-78; a cv::Mat expression = addWeighted in float: two products, one sum)
__global__ __launch_bounds__(256) void k_blend_v(float2 *__restrict__ dst, int dpitch, const float2 *__restrict__ a,
                                                 const float2 *__restrict__ b, int spitch, int w0, int h0, float alpha,
                                                 float beta)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w0 || y >= h0)
        return;
    const float2 p = a[(size_t)y * spitch + x], q = b[(size_t)y * spitch + x];
    float2 r;
    r.x = p.x * alpha + q.x * beta;
    r.y = p.y * alpha + q.y * beta;
    dst[(size_t)y * dpitch + x] = r;
}

} // namespace

void vm_launch_blend_v(float2 *dst, int dpitch, const float2 *a, const float2 *b, int spitch, int w0, int h0,
                       float alpha, float beta, hipStream_t s)
{
    dim3 blk(64, 4), g((w0 + 63) / 64, (h0 + 3) / 4);
    hipLaunchKernelGGL(k_blend_v, g, blk, 0, s, dst, dpitch, a, b, spitch, w0, h0, alpha, beta);
}

void vm_launch_upscale(float2 *dst, int w0, int h0, int dpitch, const float2 *v, int w, int h,
                       int rs, hipStream_t s)
{
    dim3 b(64, 4), g((w0 + 63) / 64, (h0 + 3) / 4);
    hipLaunchKernelGGL(k_upscale, g, b, 0, s, dst, w0, h0, dpitch, v, w, h, rs);
}

void vm_launch_render(uint8_t *out, int out_pitch, int w, int h, int rs, int ex, float color_fa,
                      float geo_fa, int color_from, const uchar4 *ext0, const uchar4 *ext1,
                      const float2 *v, const float2 *u, hipStream_t s)
{
    dim3 b(64, 4), g((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_render, g, b, 0, s, out, out_pitch, w, h, rs, ex, color_fa, geo_fa,
                       color_from, ext0, ext1, v, u);
}
