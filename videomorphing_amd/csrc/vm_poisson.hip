// vm_poisson.hip -- Poisson boundary extension on the device (gfx950).
//
// What: CPoissonExt::prepare + poissonExtend, Algorithm/PoissonExt.cpp:49-362,
// for one side of one frame: classify the extended canvas, fill the outside
// ring by following the halfway field into the other image, and solve the
// screened 5-point Poisson system that PoissonExt.cpp:214-312 assembles.
//
// How: the reference builds a CSR matrix on the host and factorises it with
// Intel MKL DSS (PoissonExt.cpp:321-329).  Here nothing is assembled: this file
// holds classification, fill, right-hand side / initial guess and paste; the
// system itself is solved matrix-free by the batched multigrid-preconditioned
// CG of vm_mgb.hip (three colour channels per 12-byte vector entry).  The
// quadratic motion path's set-up kernels (QuadraticPath.cpp:24-223) live here too.
#include "vm_internal.h"
#include "vm_poisson.h"
#include "vm_mgb.h"
#include <algorithm>
static_assert(VM_QP_SLOTS <= VM_MGB_SLOTS, "the quadratic path parks its sums in the batched solver's bb lines");

namespace {

__device__ __forceinline__ bool is_marker(uchar4 c)
{
    return c.x == 255 && c.y == 0 && c.z == 255 && c.w == 0;
}

// classification, PoissonExt.cpp:59-101
__global__ __launch_bounds__(256) void k_classify(const uchar4 *__restrict__ ext, uint8_t *type,
                                                  int cw, int ch)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cw || y >= ch)
        return;
    const size_t ii = (size_t)y * cw + x;
    uint8_t t = 0;
    if (ext[ii].w > 0)
        t = 2;
    else if ((y > 0 && ext[ii - cw].w > 0) || (y < ch - 1 && ext[ii + cw].w > 0) ||
             (x > 0 && ext[ii - 1].w > 0) || (x < cw - 1 && ext[ii + 1].w > 0))
        t = 1;
    type[ii] = t;
}

// BilineaGetColor_clamp<Vec2f,Vec2f>, PoissonExt.cpp:367-397
__device__ __forceinline__ float2 bil_v(const float2 *__restrict__ v, int w, int h, int rs,
                                        float px, float py)
{
    const int x0 = (int)floorf(px), y0 = (int)floorf(py);
    const int x1 = (int)ceilf(px), y1 = (int)ceilf(py);
    const float a = px - x0, b = py - y0;
    const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
    const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
    const float2 v00 = v[cy0 * rs + cx0], v01 = v[cy1 * rs + cx0];
    const float2 v10 = v[cy0 * rs + cx1], v11 = v[cy1 * rs + cx1];
    float2 r;
    r.x = v00.x * (1 - a) * (1 - b) + v01.x * (1 - a) * b + v10.x * a * (1 - b) + v11.x * a * b;
    r.y = v00.y * (1 - a) * (1 - b) + v01.y * (1 - a) * b + v10.y * a * (1 - b) + v11.y * a * b;
    return r;
}

__device__ __forceinline__ uint8_t sat_u8(float f)
{
    // cv::saturate_cast<uchar>(float): round half to even, clamp
    float r = rintf(f);
    return (uint8_t)fminf(fmaxf(r, 0.0f), 255.0f);
}

// outside-pixel fill, PoissonExt.cpp:104-137
__global__ __launch_bounds__(256) void k_fill(uchar4 *ext, const uint8_t *__restrict__ type,
                                              const uchar4 *__restrict__ other,
                                              const float2 *__restrict__ vf, int w, int h, int rs,
                                              int ex, int sign)
{
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cw || y >= ch)
        return;
    const size_t ii = (size_t)y * cw + x;
    if (type[ii] != 2)
        return;
    const float sg = (float)sign;
    float qx = (float)(x - ex), qy = (float)(y - ex);
    float px = qx, py = qy;
    float2 v = bil_v(vf, w, h, rs, px, py);
    const float al = 0.8f;
    for (int i = 0; i < 20; ++i) {
        px = qx + v.x * sg;
        py = qy + v.y * sg;
        float2 t = bil_v(vf, w, h, rs, px, py);
        v.x = al * t.x + (1 - al) * v.x;
        v.y = al * t.y + (1 - al) * v.y;
    }
    qx = px + v.x * sg;
    qy = py + v.y * sg;
    uchar4 o = make_uchar4(255, 0, 255, 0);
    if (qx >= 0 && qy >= 0 && qx < w && qy < h) {
        const int x0 = (int)floorf(qx), y0 = (int)floorf(qy);
        const int x1 = (int)ceilf(qx), y1 = (int)ceilf(qy);
        const float a = qx - x0, b = qy - y0;
        const int cx0 = min(max(x0, 0), w - 1), cx1 = min(max(x1, 0), w - 1);
        const int cy0 = min(max(y0, 0), h - 1), cy1 = min(max(y1, 0), h - 1);
        const uchar4 c00 = other[(size_t)cy0 * w + cx0], c01 = other[(size_t)cy1 * w + cx0];
        const uchar4 c10 = other[(size_t)cy0 * w + cx1], c11 = other[(size_t)cy1 * w + cx1];
#define BL(f) ((float)c00.f * (1 - a) * (1 - b) + (float)c01.f * (1 - a) * b + \
               (float)c10.f * a * (1 - b) + (float)c11.f * a * b)
        uchar4 c = make_uchar4(sat_u8(BL(x)), sat_u8(BL(y)), sat_u8(BL(z)), sat_u8(BL(w)));
#undef BL
        if (c.w == 0)
            o = c;
    }
    ext[ii] = o;
}

__device__ __forceinline__ float4 grad(const uchar4 *__restrict__ ext, const uint8_t *__restrict__ type,
                                       size_t a, size_t b)
{
    // gx/gy of PoissonExt.cpp:146-183: colour(a) - colour(b) when both are
    // outside pixels carrying a real colour, else 0
    if (type[a] <= 1 || type[b] <= 1)
        return make_float4(0, 0, 0, 0);
    const uchar4 ca = ext[a], cb = ext[b];
    if (is_marker(ca) || is_marker(cb))
        return make_float4(0, 0, 0, 0);
    return make_float4((float)ca.x - (float)cb.x, (float)ca.y - (float)cb.y, (float)ca.z - (float)cb.z, 0);
}

// how a solver stores a vector entry: float4 (the Jacobi solver: the diagonal travels in .w) or 12-byte VmV3
__device__ __forceinline__ void put_vec(float4 *a, size_t i, float4 v) { a[i] = v; }
__device__ __forceinline__ void put_vec(VmV3 *a, size_t i, float4 v) { a[i] = VmV3{v.x, v.y, v.z}; }
__device__ __forceinline__ float4 get_vec(const float4 *a, size_t i) { return a[i]; }
__device__ __forceinline__ float4 get_vec(const VmV3 *a, size_t i) { const VmV3 v = a[i]; return make_float4(v.x, v.y, v.z, 0); }

// right-hand side and diagonal, PoissonExt.cpp:214-270; also the initial guess
template <class V>
__global__ __launch_bounds__(256) void k_setup(const uchar4 *__restrict__ ext,
                                               const uint8_t *__restrict__ type, V *B,
                                               V *X, int cw, int ch, int init_x)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cw || y >= ch)
        return;
    const size_t ii = (size_t)y * cw + x;
    const uint8_t t = type[ii];
    float4 b = make_float4(0, 0, 0, 0);
    float diag = 0;
    if (t == 1) {
        const uchar4 c = ext[ii];
        diag += 1.0f;
        b.x += (float)c.x; b.y += (float)c.y; b.z += (float)c.z;
    }
    if (t > 0) {
        if (y - 1 >= 0 && type[ii - cw] > 0) { float4 g = grad(ext, type, ii, ii - cw); diag += 1; b.x += g.x; b.y += g.y; b.z += g.z; }
        if (x - 1 >= 0 && type[ii - 1] > 0) { float4 g = grad(ext, type, ii, ii - 1); diag += 1; b.x += g.x; b.y += g.y; b.z += g.z; }
        if (x + 1 < cw && type[ii + 1] > 0) { float4 g = grad(ext, type, ii + 1, ii); diag += 1; b.x -= g.x; b.y -= g.y; b.z -= g.z; }
        if (y + 1 < ch && type[ii + cw] > 0) { float4 g = grad(ext, type, ii + cw, ii); diag += 1; b.x -= g.x; b.y -= g.y; b.z -= g.z; }
    }
    b.w = diag; // the diagonal travels in the spare lane (float4 form)
    put_vec(B, ii, b);
    if (!init_x)
        return; // X already holds the prolongated coarse solution
    // initial guess: the colour already there (ring and filled pixels), mid grey on holes
    float4 x0 = make_float4(0, 0, 0, 0);
    if (t > 0) {
        const uchar4 c = ext[ii];
        x0 = is_marker(c) ? make_float4(128.f, 128.f, 128.f, 0) : make_float4((float)c.x, (float)c.y, (float)c.z, 0);
    }
    put_vec(X, ii, x0);
}

// block reduction of three doubles, then one double atomic per block and channel
__device__ __forceinline__ void block_sum3(double a, double b, double c, double *dst)
{
    __shared__ double sh[3][4];
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o);
        b += __shfl_down(b, o);
        c += __shfl_down(c, o);
    }
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) { sh[0][wave] = a; sh[1][wave] = b; sh[2][wave] = c; }
    __syncthreads();
    if (tid < 3) {
        double s = sh[tid][0] + sh[tid][1] + sh[tid][2] + sh[tid][3];
        if (s != 0) atomicAdd(&dst[tid], s);
    }
    __syncthreads();
}

// paste, PoissonExt.cpp:333-346
template <class V>
__global__ __launch_bounds__(256) void k_paste(uchar4 *ext, const uint8_t *__restrict__ type,
                                               const V *__restrict__ X, int cw, int ch)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cw || y >= ch)
        return;
    const size_t ii = (size_t)y * cw + x;
    if (type[ii] == 0)
        return;
    const float4 v = get_vec(X, ii);
    ext[ii] = make_uchar4((uint8_t)(int)fminf(fmaxf(v.x, 0.0f), 255.0f),
                          (uint8_t)(int)fminf(fmaxf(v.y, 0.0f), 255.0f),
                          (uint8_t)(int)fminf(fmaxf(v.z, 0.0f), 255.0f), 0);
}

__global__ __launch_bounds__(256) void k_crop(uchar4 *dst, const uchar4 *__restrict__ ext, int w, int h,
                                              int ex)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    dst[(size_t)y * w + x] = ext[(size_t)(y + ex) * (w + 2 * ex) + x + ex];
}

// Pyramid::build's extended canvas, pyramid.cu:186-200 (mixChannels of the frame with a zero alpha plane, pasted into a
// canvas of (255, 255, 255, 255)): one thread per canvas pixel; the frame's RGBA copy goes to `crop` on the way
__global__ __launch_bounds__(256) void k_canvas(uchar4 *ext, uchar4 *crop, const uint8_t *__restrict__ rgb, int w, int h, int ex)
{
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cw || y >= ch)
        return;
    const int fx = x - ex, fy = y - ex;
    uchar4 o = make_uchar4(255, 255, 255, 255);
    if (fx >= 0 && fx < w && fy >= 0 && fy < h) {
        const uint8_t *p = rgb + ((size_t)fy * w + fx) * 3;
        o = make_uchar4(p[0], p[1], p[2], 0);
        crop[(size_t)fy * w + fx] = o;
    }
    ext[(size_t)y * cw + x] = o;
}

inline dim3 grid2(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }
const dim3 B2(64, 4);


// ---------------------------------------------------------------------------
// quadratic motion path, CQuadraticPath::optimize (QuadraticPath.cpp:24-223)

// optimal Jacobian of pixel (x, y), :37-109: J0 = I - grad v, J1 = I + grad v (backward
// differences, forward on the first row / column); per column: average the directions,
// geometric mean of the lengths.  jo = (j00, j01, j10, j11) in the reference's index order.
__device__ __forceinline__ void qp_jopt(const float2 *__restrict__ v, int rs, int x, int y, float *jo)
{
    const float2 c = v[(size_t)y * rs + x];
    float2 dx, dy;
    if (x == 0) { const float2 n = v[(size_t)y * rs + x + 1]; dx = make_float2(n.x - c.x, n.y - c.y); }
    else { const float2 n = v[(size_t)y * rs + x - 1]; dx = make_float2(c.x - n.x, c.y - n.y); }
    if (y == 0) { const float2 n = v[(size_t)(y + 1) * rs + x]; dy = make_float2(n.x - c.x, n.y - c.y); }
    else { const float2 n = v[(size_t)(y - 1) * rs + x]; dy = make_float2(c.x - n.x, c.y - n.y); }
    float j0[4], j1[4];
    j0[0] = 1.0f - dx.x; j0[2] = -dx.y; j1[0] = 1.0f + dx.x; j1[2] = dx.y;
    j0[1] = -dy.x; j0[3] = 1.0f - dy.y; j1[1] = dy.x; j1[3] = 1.0f + dy.y;
    const float la0 = sqrtf(j0[0] * j0[0] + j0[2] * j0[2]), lb0 = sqrtf(j0[1] * j0[1] + j0[3] * j0[3]);
    const float la1 = sqrtf(j1[0] * j1[0] + j1[2] * j1[2]), lb1 = sqrtf(j1[1] * j1[1] + j1[3] * j1[3]);
    float nj[4];
    nj[0] = j0[0] / la0 + j1[0] / la1;
    nj[2] = j0[2] / la0 + j1[2] / la1;
    nj[1] = j0[1] / lb0 + j1[1] / lb1;
    nj[3] = j0[3] / lb0 + j1[3] / lb1;
    float la = sqrtf(nj[0] * nj[0] + nj[2] * nj[2]), lb = sqrtf(nj[1] * nj[1] + nj[3] * nj[3]);
    nj[0] /= la; nj[2] /= la; nj[1] /= lb; nj[3] /= lb;
    la = sqrtf(la0 * la1);
    lb = sqrtf(lb0 * lb1);
    jo[0] = nj[0] * la; jo[2] = nj[2] * la; jo[1] = nj[1] * lb; jo[3] = nj[3] * lb;
}

// right-hand sides, :137-170, both channels in one vector (bx, by, 0[, 0]); X = 0
template <class V>
__global__ __launch_bounds__(256) void k_qp_rhs(const float2 *__restrict__ v, int rs, int w, int h, V *B, V *X)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    float jc[4], je[4], js[4];
    qp_jopt(v, rs, x, y, jc);
    float bx = 0, by = 0;
    if (y - 1 >= 0) { bx += jc[1]; by += jc[3] - 1.0f; }
    if (x - 1 >= 0) { bx += jc[0] - 1.0f; by += jc[2]; }
    if (x + 1 < w) { qp_jopt(v, rs, x + 1, y, je); bx -= je[0] - 1.0f; by -= je[2]; }
    if (y + 1 < h) { qp_jopt(v, rs, x, y + 1, js); bx -= js[1]; by -= js[3] - 1.0f; }
    const size_t ii = (size_t)y * w + x;
    put_vec(B, ii, make_float4(bx, by, 0, 0));
    put_vec(X, ii, make_float4(0, 0, 0, 0));
}

// dst[slot * 16 + {0, 1}] += column sums of X.  Grid-stride over rows of 64-cell segments with a bounded number of
// workgroups, the atomics spread over `nslots` lines (one workgroup per 64x4 cells adding to ONE address took 99 us
// on a 1080p field: 8100 same-address double atomics in a row)
template <class V>
__global__ __launch_bounds__(256) void k_qp_sum(const V *__restrict__ X, int w, int h, double *dst, int nslots)
{
    const int gx = (w + 63) / 64, gy = (h + 3) / 4, nb = gx * gy;
    double a = 0, b = 0;
    for (int blk = blockIdx.x; blk < nb; blk += gridDim.x) {
        const int x = (blk % gx) * 64 + threadIdx.x, y = (blk / gx) * 4 + threadIdx.y;
        if (x < w && y < h) {
            const float4 v = get_vec(X, (size_t)y * w + x);
            a += v.x;
            b += v.y;
        }
    }
    block_sum3(a, b, 0.0, dst + (size_t)(blockIdx.x % nslots) * 16);
}

// B -= mean(B) (the float sums leave the right-hand side a hair off the range of the singular
// operator), or u = X - mean(X) (CG from zero converges to the zero-mean solution)
template <class V>
__global__ __launch_bounds__(256) void k_qp_shift(V *X, int w, int h, const double *__restrict__ sums, int nslots, float2 *u,
                                                  int rs)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    double s0 = 0, s1 = 0;
    for (int k = 0; k < nslots; ++k) {
        s0 += sums[k * 16];
        s1 += sums[k * 16 + 1];
    }
    const double n = (double)w * h;
    const size_t ii = (size_t)y * w + x;
    float4 v = get_vec(X, ii);
    v.x = (float)((double)v.x - s0 / n);
    v.y = (float)((double)v.y - s1 / n);
    if (u)
        u[(size_t)y * rs + x] = make_float2(v.x, v.y);
    else
        put_vec(X, ii, v);
}

} // namespace

void vm_poisson_launch_crop(uchar4 *dst, const uchar4 *ext, int w, int h, int ex, hipStream_t s)
{
    hipLaunchKernelGGL(k_crop, grid2(w, h), B2, 0, s, dst, ext, w, h, ex);
}

void vm_poisson_launch_canvas(uchar4 *ext, uchar4 *crop, const uint8_t *rgb, int w, int h, int ex, hipStream_t s)
{
    hipLaunchKernelGGL(k_canvas, grid2(w + 2 * ex, h + 2 * ex), B2, 0, s, ext, crop, rgb, w, h, ex);
}

void vm_poisson_launch_prepare(uchar4 *ext, uint8_t *type, const uchar4 *other, const float2 *v,
                               int w, int h, int rs, int ex, int sign, hipStream_t s)
{
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    hipLaunchKernelGGL(k_classify, grid2(cw, ch), B2, 0, s, ext, type, cw, ch);
    hipLaunchKernelGGL(k_fill, grid2(cw, ch), B2, 0, s, ext, type, other, v, w, h, rs, ex, sign);
}

void vm_poisson_launch_setup3(const uchar4 *ext, const uint8_t *type, VmV3 *B, VmV3 *X, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_setup<VmV3>, grid2(cw, ch), B2, 0, s, ext, type, B, X, cw, ch, 1);
}

void vm_poisson_launch_paste3(uchar4 *ext, const uint8_t *type, const VmV3 *X, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_paste<VmV3>, grid2(cw, ch), B2, 0, s, ext, type, X, cw, ch);
}

void vm_qpath_launch_rhs3(const float2 *v, int rs, int w, int h, VmV3 *B, VmV3 *X, hipStream_t s)
{
    hipLaunchKernelGGL(k_qp_rhs<VmV3>, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, s, v, rs, w, h, B, X);
}

// sums[0..1] += column sums of X (sums must be zeroed by the caller)
static inline dim3 sum_grid(int w, int h)
{
    return dim3(std::min(((w + 63) / 64) * ((h + 3) / 4), 1024));
}

// the atomics spread over VM_QP_SLOTS lines of 16 doubles (the solver's scalar block has room for 8)
void vm_qpath_launch_sum3(const VmV3 *X, int w, int h, double *sums, hipStream_t s)
{
    hipLaunchKernelGGL(k_qp_sum<VmV3>, sum_grid(w, h), dim3(64, 4), 0, s, X, w, h, sums, VM_QP_SLOTS);
}

// u == nullptr: X -= mean in place; else u = X - mean
void vm_qpath_launch_shift3(VmV3 *X, int w, int h, const double *sums, float2 *u, int rs, hipStream_t s)
{
    hipLaunchKernelGGL(k_qp_shift<VmV3>, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, s, X, w, h, sums, VM_QP_SLOTS, u, rs);
}
