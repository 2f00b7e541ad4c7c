// vm_pyramid.hip -- device-side luma pyramid builder for gfx950.
//
// What: the stage-2 image chain of Pyramid::build (Algorithm/pyramid.cu:203-211,
// 268-279, 355-364): load (x/255, sRGB -> linear), per level scale() with the
// cardinal cubic B-spline generalized-sampling kernel of include/resample
// (scale.cpp:9-272, dlti.cpp:66-270: weighted B-spline gather, then the inverse of
// the sampled B-spline [1/6 4/6 1/6] with mirror boundary along the same axis),
// store_gray (clamp, linear -> sRGB, x255, .299 R + .587 G + .114 B).
//
// How: the reference runs this on the CPU (OpenMP) and uploads each level; here the
// frame is uploaded once as RGB8 and everything stays in HBM.  Planar f32 linear
// light, one thread per output sample for the gathers, one thread per image line
// for the tridiagonal solve (its LU factors depend only on the line length and are
// computed on the host, vm_pyramid_api.cpp).  HBM-bound streaming kernels except the
// per-line recursions, which are latency-bound (a few hundred lines of 10^3 samples).
#include "vm_internal.h"
#include "vm_pyramid.h"

namespace {

__device__ __forceinline__ float srgbcurve(float f) // color.h:7-17
{
    const float a = 0.055f;
    return f <= 0.0031308f ? 12.92f * f : (1.f + a) * powf(f, 1.f / 2.4f) - a;
}
__device__ __forceinline__ float srgbuncurve(float f) // color.h:26-35
{
    const float a = 0.055f;
    return f <= 0.04045f ? f / 12.92f : powf((f + a) / (1.f + a), 2.4f);
}
__device__ __forceinline__ float bspline3(float r) // generating.h:220-232
{
    r = fabsf(r);
    if (r < 1.f) return (4.f + r * r * (-6.f + 3.f * r)) / 6.f;
    if (r < 2.f) return (8.f + r * (-12.f + (6.f - r) * r)) / 6.f;
    return 0.f;
}
__device__ __forceinline__ int ext_mirror(int i, int n) // extension.h:43-66
{
    const int m = 2 * n;
    i = i >= 0 ? i % m : (m - 1) - ((-i - 1) % m);
    return i >= n ? m - i - 1 : i;
}

// image::load, image.cpp:10-31: RGB8 (pitched) -> 3 linear-light planes
__global__ __launch_bounds__(256) void k_load(const uint8_t *__restrict__ rgb, int pitch, float *__restrict__ img,
                                              int w, int h)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    const float tof = 1.f / 255.f;
    const size_t n = (size_t)w * h, p = (size_t)y * w + x;
    const uint8_t *s = rgb + (size_t)y * pitch + 3 * x;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        img[k * n + p] = srgbuncurve((float)s[k] * tof);
}

__global__ __launch_bounds__(256) void k_curve(float *img, size_t n, int to_gamma)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        img[i] = to_gamma ? srgbcurve(img[i]) : srgbuncurve(img[i]);
}

// downsample_rows / downsample_columns gather, scale.cpp:125-223 (before the inverse filter)
__global__ __launch_bounds__(256) void k_down(const float *__restrict__ src, float *__restrict__ dst, int win,
                                              int hin, int nout, int axis)
{
    const int wout = axis == 0 ? nout : win, hout = axis == 0 ? hin : nout;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, c = blockIdx.z;
    if (x >= wout || y >= hout)
        return;
    const int nin = axis == 0 ? win : hin, o = axis == 0 ? x : y;
    const float inv_nin = 1.f / (float)nin;
    const float inv_sw = (float)nout * inv_nin;
    const float sw = 1.f / inv_sw;
    const float s = 4.f;
    int lo = (int)ceilf(.5f * sw * (2.f * o + 1.f - s) - .5f);
    int hi = (int)floorf(.5f * sw * (2.f * o + 1.f + s) - .5f);
    if (lo > hi)
        lo = hi = (int)(.5f * sw * (2.f * o + 1.f));
    const float *plane = src + (size_t)c * win * hin;
    float sum = 0.f, sum_w = 0.f;
    for (int i = lo; i <= hi; ++i) {
        const float kj = (float)(0.5 + o - (i + 0.5f) * inv_sw);
        const float wgt = bspline3(kj);
        const int q = min(max(ext_mirror(i, nin), 0), nin - 1);
        sum += (axis == 0 ? plane[(size_t)y * win + q] : plane[(size_t)q * win + x]) * wgt;
        sum_w += wgt;
    }
    dst[(size_t)c * wout * hout + (size_t)y * wout + x] = sum / sum_w;
}

// upsample_rows / upsample_columns reconstruction, scale.cpp:9-123 (input already prefiltered)
__global__ __launch_bounds__(256) void k_up(const float *__restrict__ src, float *__restrict__ dst, int win,
                                            int hin, int nout, int axis)
{
    const int wout = axis == 0 ? nout : win, hout = axis == 0 ? hin : nout;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, c = blockIdx.z;
    if (x >= wout || y >= hout)
        return;
    const int nin = axis == 0 ? win : hin, o = axis == 0 ? x : y;
    const float inv_nout = 1.f / (float)nout;
    const float inv_sw = (float)nin * inv_nout;
    const float f = ((float)o + .5f) * inv_sw - .5f;
    const int ci = (int)floorf(f);
    const float d = f - ci;
    const float *plane = src + (size_t)c * win * hin;
    float sum = 0.f;
#pragma unroll
    for (int j = -1; j <= 2; ++j) {
        const int q = min(max(ext_mirror(ci + j, nin), 0), nin - 1);
        sum += (axis == 0 ? plane[(size_t)y * win + q] : plane[(size_t)q * win + x]) * bspline3(d - j);
    }
    dst[(size_t)c * wout * hout + (size_t)y * wout + x] = sum;
}

// solve_rows / solve_columns, dlti.cpp:97-168: one thread per line.  A: the factored
// band, A[(i-j+1)*n + j] (lower, inverse pivot, upper).
// The two recursions are chains of dependent multiply-subtracts; what made them slow was a
// memory round trip per element (load, update, store through one pointer).  A line is walked in
// chunks of VM_TRI_CH elements held in registers: the loads of a chunk are issued together, the
// next chunk's while the current one is computed -- same operations in the same order, 3-4x less
// waiting (k_tri_solve was 85 % of the flow pyramid of a video).
#define VM_TRI_CH 32
__global__ __launch_bounds__(64) void k_tri_solve(float *img, const float *__restrict__ A, int w, int h, int axis)
{
    const int n = axis == 0 ? w : h, lines = axis == 0 ? h : w;
    const int l = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y;
    if (l >= lines)
        return;
    float *x = img + (size_t)c * w * h + (axis == 0 ? (size_t)l * w : (size_t)l);
    const size_t st = axis == 0 ? 1 : (size_t)w;
    const float *lower = A + 2 * (size_t)n, *diag = A + (size_t)n, *upper = A;
    // forward: x[j] -= A(j, j-1) * x[j-1]
    {
        float prev = x[0];
        int j = 1;
        float cur[VM_TRI_CH], nxt[VM_TRI_CH];
        if (j + VM_TRI_CH <= n) {
#pragma unroll
            for (int k = 0; k < VM_TRI_CH; ++k) cur[k] = x[(size_t)(j + k) * st];
        }
        for (; j + VM_TRI_CH <= n; j += VM_TRI_CH) {
            const bool more = j + 2 * VM_TRI_CH <= n;
            if (more) {
#pragma unroll
                for (int k = 0; k < VM_TRI_CH; ++k) nxt[k] = x[(size_t)(j + VM_TRI_CH + k) * st];
            }
            float lo[VM_TRI_CH];
#pragma unroll
            for (int k = 0; k < VM_TRI_CH; ++k) lo[k] = lower[j + k - 1];
#pragma unroll
            for (int k = 0; k < VM_TRI_CH; ++k) {
                const float v = cur[k] - lo[k] * prev;
                cur[k] = v;
                prev = v;
            }
#pragma unroll
            for (int k = 0; k < VM_TRI_CH; ++k) x[(size_t)(j + k) * st] = cur[k];
            if (more) {
#pragma unroll
                for (int k = 0; k < VM_TRI_CH; ++k) cur[k] = nxt[k];
            }
        }
        for (; j < n; ++j) {
            const float v = x[(size_t)j * st] - lower[j - 1] * prev;
            x[(size_t)j * st] = v;
            prev = v;
        }
    }
    // backward: x[j] = (x[j] - A(j, j+1) * x[j+1]) * inverse pivot
    {
        float next = 0.f;
        int j = n - 1;
        // the last element has no upper neighbour
        {
            float v = x[(size_t)j * st];
            v *= diag[j];
            x[(size_t)j * st] = v;
            next = v;
            --j;
        }
        float cur[VM_TRI_CH], nxt[VM_TRI_CH];
        if (j - VM_TRI_CH + 1 >= 0) {
#pragma unroll
            for (int k = 0; k < VM_TRI_CH; ++k) cur[k] = x[(size_t)(j - k) * st];
        }
        for (; j - VM_TRI_CH + 1 >= 0; j -= VM_TRI_CH) {
            const bool more = j - 2 * VM_TRI_CH + 1 >= 0;
            if (more) {
#pragma unroll
                for (int k = 0; k < VM_TRI_CH; ++k) nxt[k] = x[(size_t)(j - VM_TRI_CH - k) * st];
            }
            float up[VM_TRI_CH], dg[VM_TRI_CH];
#pragma unroll
            for (int k = 0; k < VM_TRI_CH; ++k) {
                up[k] = upper[j - k + 1];
                dg[k] = diag[j - k];
            }
#pragma unroll
            for (int k = 0; k < VM_TRI_CH; ++k) {
                float v = cur[k];
                v -= up[k] * next;
                v *= dg[k];
                cur[k] = v;
                next = v;
            }
#pragma unroll
            for (int k = 0; k < VM_TRI_CH; ++k) x[(size_t)(j - k) * st] = cur[k];
            if (more) {
#pragma unroll
                for (int k = 0; k < VM_TRI_CH; ++k) cur[k] = nxt[k];
            }
        }
        for (; j >= 0; --j) {
            float v = x[(size_t)j * st];
            v -= upper[j + 1] * next;
            v *= diag[j];
            x[(size_t)j * st] = v;
            next = v;
        }
    }
}

// image::store_gray, image.cpp:87-103, into the level's pitched luma array
__global__ __launch_bounds__(256) void k_store_gray(const float *__restrict__ img, float *__restrict__ luma, int w,
                                                    int h, int rs)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    const size_t n = (size_t)w * h, p = (size_t)y * w + x;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = fminf(fmaxf(img[k * n + p], 0.f), 1.f);
        c[k] = srgbcurve(v) * 255;
    }
    luma[(size_t)y * rs + x] = (float)(c[0] * 0.299 + c[1] * 0.587 + c[2] * 0.114);
}

inline dim3 g2(int w, int h, int z = 1) { return dim3((w + 63) / 64, (h + 3) / 4, z); }
const dim3 B2(64, 4);

} // namespace

void vm_pyr_launch_load(const uint8_t *rgb, int pitch, float *img, int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_load, g2(w, h), B2, 0, s, rgb, pitch, img, w, h);
}
void vm_pyr_launch_curve(float *img, size_t n, int to_gamma, hipStream_t s)
{
    hipLaunchKernelGGL(k_curve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, n, to_gamma);
}
void vm_pyr_launch_down(const float *src, float *dst, int win, int hin, int nout, int axis, hipStream_t s)
{
    const int wout = axis == 0 ? nout : win, hout = axis == 0 ? hin : nout;
    hipLaunchKernelGGL(k_down, g2(wout, hout, 3), B2, 0, s, src, dst, win, hin, nout, axis);
}
void vm_pyr_launch_up(const float *src, float *dst, int win, int hin, int nout, int axis, hipStream_t s)
{
    const int wout = axis == 0 ? nout : win, hout = axis == 0 ? hin : nout;
    hipLaunchKernelGGL(k_up, g2(wout, hout, 3), B2, 0, s, src, dst, win, hin, nout, axis);
}
void vm_pyr_launch_tri_solve(float *img, const float *A, int w, int h, int axis, hipStream_t s)
{
    const int lines = axis == 0 ? h : w;
    hipLaunchKernelGGL(k_tri_solve, dim3((lines + 63) / 64, 3), dim3(64), 0, s, img, A, w, h, axis);
}
void vm_pyr_launch_store_gray(const float *img, float *luma, int w, int h, int rs, hipStream_t s)
{
    hipLaunchKernelGGL(k_store_gray, g2(w, h), B2, 0, s, img, luma, w, h, rs);
}
