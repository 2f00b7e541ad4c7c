// vm_temporal.hip -- the temporal coherence path of the halfway optimizer for gfx950:
// temp_ref, interpolate_temp_ref, smooth, fill_zeros_x and kernel_initialize_temp of
// Algorithm/upsample.cu:28-211, and the flow half of Pyramid::build
// (Algorithm/pyramid.cu:284-321, 375-442).  The energy term these fields feed
// (energy_change with flag == true, morph.cu:752-759) is in vm_sweep_kernels.hip.
//
// One arithmetic mode (IEEE, no contraction): none of this is hot -- a few streaming
// passes per page per level -- and both optimizer modes share it.
//
// temp_ref is a forward splat: every pixel of the neighbouring page scatters its
// advected vector into <= 4 pixels.  The reference does that with float atomicAdd in an
// unspecified order (upsample.cu:57-58), so its own result varies from run to run in the
// last bits.  Here every contribution is formed exactly as the reference writes it (incl.
// its double-precision weight expression) and accumulated by 64-bit INTEGER atomics in
// fixed point (x 2^32, round to nearest even): order-independent, deterministic and
// bit-identical to the oracle, at HBM atomic rate (3 atomics x <= 4 targets per pixel).
#include "vm_internal.h"
#include "vm_temporal.h"

namespace {

// tex2D(float2 texture, linear, clamp, unnormalised), upsample.cu:227-233
__device__ __forceinline__ float2 tapf2(const float2 *__restrict__ img, int w, int h, int rs, float x, float y)
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

__device__ __forceinline__ long long to_fixed(float c) { return __double2ll_rn((double)c * 4294967296.0); }
__device__ __forceinline__ float from_fixed(long long a) { return (float)((double)a / 4294967296.0); }

// temp_ref, upsample.cu:28-62.  acc: 3 int64 per pixel (v.x, v.y, weight), pitch rs
__global__ __launch_bounds__(256) void k_temp_splat(int w, int h, int rs, const float2 *__restrict__ v_prev,
                                                    const float2 *__restrict__ f0, const float2 *__restrict__ f1,
                                                    const float *__restrict__ ssim, long long *__restrict__ acc)
{
    const int px = blockIdx.x * 64 + threadIdx.x, py = blockIdx.y * 4 + threadIdx.y;
    if (px >= w || py >= h)
        return;
    const int idx = py * rs + px;
    const float p_x = (float)px, p_y = (float)py;
    const float2 v = v_prev[idx];
    const float2 a = tapf2(f0, w, h, rs, p_x - v.x + 0.5f, p_y - v.y + 0.5f);
    const float2 b = tapf2(f1, w, h, rs, p_x + v.x + 0.5f, p_y + v.y + 0.5f);
    const float prx = p_x + 0.5f * (a.x + b.x), pry = p_y + 0.5f * (a.y + b.y);
    const float vrx = v.x + 0.5f * (b.x - a.x), vry = v.y + 0.5f * (b.y - a.y);
    const int xx = (int)floorf(prx), yy = (int)floorf(pry);
    const float ssim_fa = ssim ? ssim[idx] : 1.0f;
    for (int y = yy; y <= yy + 1; ++y)
        for (int x = xx; x <= xx + 1; ++x) {
            if (x < 0 || x >= w || y < 0 || y >= h)
                continue;
            // ssim_fa*(1.0-abs((float)x-p_ref.x))*(1.0-abs((float)y-p_ref.y)): double literals
            const float fa = (float)((double)ssim_fa * (1.0 - (double)fabsf((float)x - prx)) *
                                     (1.0 - (double)fabsf((float)y - pry)));
            unsigned long long *d = (unsigned long long *)(acc + 3 * (size_t)(y * rs + x));
            atomicAdd(d + 0, (unsigned long long)to_fixed(vrx * fa));
            atomicAdd(d + 1, (unsigned long long)to_fixed(vry * fa));
            atomicAdd(d + 2, (unsigned long long)to_fixed(fa));
        }
}

// fixed point -> float, interpolate_temp_ref (upsample.cu:64-77), and then either
//  - kernel_initialize_temp (upsample.cu:190-211): temp.ref / temp.mask of the page, or
//  - v_cur / weight for the in-between page of upsample() (:322-328)
__global__ __launch_bounds__(256) void k_temp_finish(int w, int h, int rs, const long long *__restrict__ acc,
                                                     float2 *__restrict__ ref_out, float *__restrict__ mask_out,
                                                     int init_temp)
{
    const int px = blockIdx.x * 64 + threadIdx.x, py = blockIdx.y * 4 + threadIdx.y;
    if (px >= w || py >= h)
        return;
    const size_t idx = (size_t)py * rs + px;
    float x = from_fixed(acc[3 * idx]), y = from_fixed(acc[3 * idx + 1]);
    const float wt = from_fixed(acc[3 * idx + 2]);
    if (wt > 0) {
        x /= wt;
        y /= wt;
    }
    if (init_temp) {
        if (wt > 0) {
            ref_out[idx] = make_float2(x, y);
            mask_out[idx] = wt;
        } else {
            mask_out[idx] = 0.0f;
        }
    } else {
        ref_out[idx] = make_float2(x, y);
        mask_out[idx] = wt;
    }
}

// smooth, upsample.cu:80-111 (v_out pre-filled with zeros)
__global__ __launch_bounds__(256) void k_smooth(int w, int h, int rs, float2 *__restrict__ v_out,
                                                const float2 *__restrict__ v_cur, const float *__restrict__ weight)
{
    const int px = blockIdx.x * 64 + threadIdx.x, py = blockIdx.y * 4 + threadIdx.y;
    if (px >= w || py >= h)
        return;
    float ww = 0.0f, sx = 0, sy = 0;
    for (int y = py - 1; y <= py + 1; ++y)
        for (int x = px - 1; x <= px + 1; ++x) {
            if (x < 0 || x >= w || y < 0 || y >= h)
                continue;
            const int idx = y * rs + x;
            if (weight[idx] > 0) {
                ww += 1;
                const float2 c = v_cur[idx];
                sx += c.x;
                sy += c.y;
            }
        }
    if (ww > 0)
        v_out[py * rs + px] = make_float2(sx / ww, sy / ww);
}

// fill_zeros_x, upsample.cu:115-151 (incl. its un-weighted numerator).  Writes pixels
// without weight only, reads pixels with weight only: no ordering issue.  fill_zeros_y
// (:153-189) only sets entries of the weight array that is freed right after: not built.
__global__ __launch_bounds__(256) void k_fill_zeros_x(int w, int h, int rs, float2 *__restrict__ v_out,
                                                      const float *__restrict__ weight)
{
    const int px = blockIdx.x * 64 + threadIdx.x, py = blockIdx.y * 4 + threadIdx.y;
    if (px >= w || py >= h)
        return;
    const int row = py * rs;
    if (weight[row + px] > 0)
        return;
    float ww = 0.0f, sx = 0, sy = 0;
    for (int x = px; x >= 0; --x)
        if (weight[row + x] > 0) {
            ww = (float)((double)ww + 1.0 / (px - x));
            const float2 c = v_out[row + x];
            sx += c.x;
            sy += c.y;
            break;
        }
    for (int x = px; x < w; ++x)
        if (weight[row + x] > 0) {
            ww = (float)((double)ww + 1.0 / (x - px));
            const float2 c = v_out[row + x];
            sx += c.x;
            sy += c.y;
            break;
        }
    if (ww > 0)
        v_out[row + px] = make_float2(sx / ww, sy / ww);
}

// ---- flow half of Pyramid::build ----
__device__ __forceinline__ float srgbcurve(float f) // include/resample/color.h:7-17
{
    const float a = 0.055f;
    return f <= 0.0031308f ? 12.92f * f : (1.f + a) * powf(f, 1.f / 2.4f) - a;
}
__device__ __forceinline__ float srgbuncurve(float f) // color.h:26-35
{
    const float a = 0.055f;
    return f <= 0.04045f ? f / 12.92f : powf((f + a) / (1.f + a), 2.4f);
}

// image::load(rgba, data, w, h, rowstride, -50, 50), image.cpp:33-54: flow (pitched float2)
// -> r, g planes through srgbuncurve, b plane = 1
__global__ __launch_bounds__(256) void k_flow_load(const float2 *__restrict__ flow, int pitch, float *__restrict__ img,
                                                   int w, int h)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    const float mn = -50.f, mx = 50.f;
    const float tof = 1.f / (mx - mn);
    const size_t n = (size_t)w * h, p = (size_t)y * w + x;
    const float2 f = flow[(size_t)y * pitch + x];
    img[p] = srgbuncurve((f.x - mn) * tof);
    img[n + p] = srgbuncurve((f.y - mn) * tof);
    img[2 * n + p] = 1.0f;
}

// image::store(data, rgba, rowstride, -50, 50), image.cpp:72-85, then x (ratiox, ratioy)
// when the size shrank (pyramid.cu:306-321, 389-404)
__global__ __launch_bounds__(256) void k_flow_store(const float *__restrict__ img, float2 *__restrict__ flow, int pitch,
                                                    int w, int h, float ratiox, float ratioy)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    const float mn = -50.f, mx = 50.f;
    const size_t n = (size_t)w * h, p = (size_t)y * w + x;
    float fx = srgbcurve(fminf(fmaxf(img[p], 0.f), 1.f)) * (mx - mn) + mn;
    float fy = srgbcurve(fminf(fmaxf(img[n + p], 0.f), 1.f)) * (mx - mn) + mn;
    if (ratiox < 1 || ratioy < 1) {
        fx *= ratiox;
        fy *= ratioy;
    }
    flow[(size_t)y * pitch + x] = make_float2(fx, fy);
}

// temporal concatenation, pyramid.cu:406-442 with Pyramid::BiLinear (:486-522):
// out(p) = f(p) + BiLinear(f_next, p + f(p)); out may alias f (each pixel reads its own f only)
__global__ __launch_bounds__(256) void k_flow_concat(float2 *f, const float2 *__restrict__ f_next, int pitch, int w, int h)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    const float2 c = f[(size_t)y * pitch + x];
    const float px = (float)x + c.x, py = (float)y + c.y;
    int xs[2], ys[2];
    xs[0] = (int)floorf(px);
    ys[0] = (int)floorf(py);
    xs[1] = (int)ceilf(px);
    ys[1] = (int)ceilf(py);
    const float u = px - xs[0], v = py - ys[0];
    float2 val[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tx = min(max(xs[i], 0), w - 1), ty = min(max(ys[j], 0), h - 1);
            val[i][j] = f_next[(size_t)ty * pitch + tx];
        }
    float2 o;
    o.x = val[0][0].x * (1 - u) * (1 - v) + val[0][1].x * (1 - u) * v + val[1][0].x * u * (1 - v) + val[1][1].x * u * v;
    o.y = val[0][0].y * (1 - u) * (1 - v) + val[0][1].y * (1 - u) * v + val[1][0].y * u * (1 - v) + val[1][1].y * u * v;
    f[(size_t)y * pitch + x] = make_float2(c.x + o.x, c.y + o.y);
}

inline dim3 g2(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }
const dim3 B2(64, 4);

} // namespace

void vm_temp_launch_splat(int w, int h, int rs, const float2 *v_prev, const float2 *f0, const float2 *f1,
                          const float *ssim, long long *acc, hipStream_t s)
{
    hipLaunchKernelGGL(k_temp_splat, g2(w, h), B2, 0, s, w, h, rs, v_prev, f0, f1, ssim, acc);
}
void vm_temp_launch_finish(int w, int h, int rs, const long long *acc, float2 *ref_out, float *mask_out, int init_temp,
                           hipStream_t s)
{
    hipLaunchKernelGGL(k_temp_finish, g2(w, h), B2, 0, s, w, h, rs, acc, ref_out, mask_out, init_temp);
}
void vm_temp_launch_smooth(int w, int h, int rs, float2 *v_out, const float2 *v_cur, const float *weight, hipStream_t s)
{
    hipLaunchKernelGGL(k_smooth, g2(w, h), B2, 0, s, w, h, rs, v_out, v_cur, weight);
}
void vm_temp_launch_fill_zeros_x(int w, int h, int rs, float2 *v_out, const float *weight, hipStream_t s)
{
    hipLaunchKernelGGL(k_fill_zeros_x, g2(w, h), B2, 0, s, w, h, rs, v_out, weight);
}
void vm_flow_launch_load(const float2 *flow, int pitch, float *img, int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_flow_load, g2(w, h), B2, 0, s, flow, pitch, img, w, h);
}
void vm_flow_launch_store(const float *img, float2 *flow, int pitch, int w, int h, float ratiox, float ratioy,
                          hipStream_t s)
{
    hipLaunchKernelGGL(k_flow_store, g2(w, h), B2, 0, s, img, flow, pitch, w, h, ratiox, ratioy);
}
void vm_flow_launch_concat(float2 *f, const float2 *f_next, int pitch, int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_flow_concat, g2(w, h), B2, 0, s, f, f_next, pitch, w, h);
}
