// vm_poisson.hip -- Poisson boundary extension on the device (gfx950).
//
// What: CPoissonExt::prepare + poissonExtend, Algorithm/PoissonExt.cpp:49-362,
// for one side of one frame: classify the extended canvas, fill the outside
// ring by following the halfway field into the other image, and solve the
// screened 5-point Poisson system that PoissonExt.cpp:214-312 assembles.
//
// How: the reference builds a CSR matrix on the host and factorises it with
// Intel MKL DSS (PoissonExt.cpp:321-329).  Here nothing is assembled: the
// operator is applied matrix-free from the 1-byte type map, the three colour
// channels ride together in one float4 per pixel (16-byte coalesced accesses),
// and the system is solved by Jacobi-preconditioned conjugate gradients whose
// scalars (alpha, beta, residual norms) stay in device memory, so an iteration
// is three kernel launches with no host round trip.  Dot products accumulate
// in double.  HBM-bound: ~5 float4 vectors touched per unknown per iteration.
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

__device__ __forceinline__ float4 apply_A(const float4 *__restrict__ P, const uint8_t *__restrict__ type,
                                          float diag, size_t ii, int x, int y, int cw, int ch)
{
    float4 c = P[ii];
    float4 s = make_float4(diag * c.x, diag * c.y, diag * c.z, 0);
    if (y - 1 >= 0 && type[ii - cw] > 0) { float4 n = P[ii - cw]; s.x -= n.x; s.y -= n.y; s.z -= n.z; }
    if (x - 1 >= 0 && type[ii - 1] > 0) { float4 n = P[ii - 1]; s.x -= n.x; s.y -= n.y; s.z -= n.z; }
    if (x + 1 < cw && type[ii + 1] > 0) { float4 n = P[ii + 1]; s.x -= n.x; s.y -= n.y; s.z -= n.z; }
    if (y + 1 < ch && type[ii + cw] > 0) { float4 n = P[ii + cw]; s.x -= n.x; s.y -= n.y; s.z -= n.z; }
    return s;
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

// r = b - A x, z = r / diag, p = z;  scal.rz[c] = r.z, scal.bb[c] = b.b, scal.rr[c] = r.r
__global__ __launch_bounds__(256) void k_cg_init(const float4 *__restrict__ B, const float4 *__restrict__ X,
                                                 float4 *R, float4 *P, const uint8_t *__restrict__ type,
                                                 VmCgScalars *sc, int cw, int ch)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    double rz[3] = {0, 0, 0}, bb[3] = {0, 0, 0}, rr[3] = {0, 0, 0};
    if (x < cw && y < ch) {
        const size_t ii = (size_t)y * cw + x;
        if (type[ii] > 0) {
            const float4 b = B[ii];
            const float4 ax = apply_A(X, type, b.w, ii, x, y, cw, ch);
            const float4 r = make_float4(b.x - ax.x, b.y - ax.y, b.z - ax.z, 0);
            const float inv = 1.0f / b.w;
            R[ii] = r;
            P[ii] = make_float4(r.x * inv, r.y * inv, r.z * inv, 0);
            rz[0] = (double)r.x * r.x * inv; rz[1] = (double)r.y * r.y * inv; rz[2] = (double)r.z * r.z * inv;
            bb[0] = (double)b.x * b.x; bb[1] = (double)b.y * b.y; bb[2] = (double)b.z * b.z;
            rr[0] = (double)r.x * r.x; rr[1] = (double)r.y * r.y; rr[2] = (double)r.z * r.z;
        } else {
            R[ii] = make_float4(0, 0, 0, 0);
            P[ii] = make_float4(0, 0, 0, 0);
        }
    }
    block_sum3(rz[0], rz[1], rz[2], sc->rz);
    block_sum3(bb[0], bb[1], bb[2], sc->bb);
    block_sum3(rr[0], rr[1], rr[2], sc->rr);
}

// q = A p;  pq[c] += p.q
__global__ __launch_bounds__(256) void k_cg_spmv(const float4 *__restrict__ P, float4 *Q,
                                                 const float4 *__restrict__ B,
                                                 const uint8_t *__restrict__ type, VmCgScalars *sc,
                                                 int cw, int ch)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    double pq[3] = {0, 0, 0};
    if (x < cw && y < ch) {
        const size_t ii = (size_t)y * cw + x;
        if (type[ii] > 0) {
            const float4 q = apply_A(P, type, B[ii].w, ii, x, y, cw, ch);
            const float4 p = P[ii];
            Q[ii] = q;
            pq[0] = (double)p.x * q.x; pq[1] = (double)p.y * q.y; pq[2] = (double)p.z * q.z;
        }
    }
    block_sum3(pq[0], pq[1], pq[2], sc->pq);
}

// alpha = rz/pq;  x += alpha p;  r -= alpha q;  rz_new += r.(r/diag);  rr_new += r.r
__global__ __launch_bounds__(256) void k_cg_update(float4 *X, float4 *R, const float4 *__restrict__ P,
                                                   const float4 *__restrict__ Q,
                                                   const float4 *__restrict__ B,
                                                   const uint8_t *__restrict__ type, VmCgScalars *sc,
                                                   int cw, int ch)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    double rz[3] = {0, 0, 0}, rr[3] = {0, 0, 0};
    if (x < cw && y < ch) {
        const size_t ii = (size_t)y * cw + x;
        if (type[ii] > 0) {
            float al[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                al[c] = sc->pq[c] > 0 ? (float)(sc->rz[c] / sc->pq[c]) : 0.0f;
            const float4 p = P[ii], q = Q[ii];
            float4 xx = X[ii], r = R[ii];
            xx.x += al[0] * p.x; xx.y += al[1] * p.y; xx.z += al[2] * p.z;
            r.x -= al[0] * q.x; r.y -= al[1] * q.y; r.z -= al[2] * q.z;
            X[ii] = xx;
            R[ii] = r;
            const float inv = 1.0f / B[ii].w;
            rz[0] = (double)r.x * r.x * inv; rz[1] = (double)r.y * r.y * inv; rz[2] = (double)r.z * r.z * inv;
            rr[0] = (double)r.x * r.x; rr[1] = (double)r.y * r.y; rr[2] = (double)r.z * r.z;
        }
    }
    block_sum3(rz[0], rz[1], rz[2], sc->rz_new);
    block_sum3(rr[0], rr[1], rr[2], sc->rr_new);
}

// beta = rz_new/rz;  p = r/diag + beta p
__global__ __launch_bounds__(256) void k_cg_dir(float4 *P, const float4 *__restrict__ R,
                                                const float4 *__restrict__ B,
                                                const uint8_t *__restrict__ type,
                                                const VmCgScalars *sc, int cw, int ch)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cw || y >= ch)
        return;
    const size_t ii = (size_t)y * cw + x;
    if (type[ii] == 0)
        return;
    float be[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        be[c] = sc->rz[c] > 0 ? (float)(sc->rz_new[c] / sc->rz[c]) : 0.0f;
    const float inv = 1.0f / B[ii].w;
    const float4 r = R[ii];
    float4 p = P[ii];
    p.x = r.x * inv + be[0] * p.x;
    p.y = r.y * inv + be[1] * p.y;
    p.z = r.z * inv + be[2] * p.z;
    P[ii] = p;
}

// rotate the scalars for the next iteration; record the residual history
__global__ void k_cg_rotate(VmCgScalars *sc)
{
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        sc->rz[c] = sc->rz_new[c];
        sc->rr[c] = sc->rr_new[c];
        sc->rz_new[c] = 0;
        sc->rr_new[c] = 0;
        sc->pq[c] = 0;
    }
    if (threadIdx.x == 0)
        sc->iters += 1;
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

// nested iteration: the same problem on a 4x coarser canvas gives the fine solve its
// low frequencies (the outside band is ~0.1 max(W,H) pixels wide, which plain CG has to
// cross one pixel per iteration).  A coarse pixel is outside (2) if its 4x4 block holds
// only outside pixels, interior (0) if only interior ones, else an anchor (1) carrying
// the mean colour of the block's non-outside pixels; outside blocks carry the mean of
// their real colours (marker if none).
__global__ __launch_bounds__(256) void k_coarsen(const uchar4 *__restrict__ ext, const uint8_t *__restrict__ type,
                                                 uchar4 *ext_c, uint8_t *type_c, int cw, int ch, int cw2, int ch2)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cw2 || y >= ch2)
        return;
    int n_out = 0, n_ring = 0, n_col = 0, n_anchor = 0;
    float3 col = make_float3(0, 0, 0), anc = make_float3(0, 0, 0);
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            const int fx = 4 * x + i, fy = 4 * y + j;
            if (fx >= cw || fy >= ch)
                continue;
            const size_t ii = (size_t)fy * cw + fx;
            const uint8_t t = type[ii];
            const uchar4 c = ext[ii];
            if (t == 2) {
                ++n_out;
                if (!is_marker(c)) { ++n_col; col.x += c.x; col.y += c.y; col.z += c.z; }
            } else {
                if (t == 1) ++n_ring;
                ++n_anchor; anc.x += c.x; anc.y += c.y; anc.z += c.z;
            }
        }
    uint8_t t = 0;
    uchar4 o = make_uchar4(255, 0, 255, 0);
    if (n_out > 0 && n_anchor == 0) {
        t = 2;
        if (n_col > 0)
            o = make_uchar4((uint8_t)(col.x / n_col + 0.5f), (uint8_t)(col.y / n_col + 0.5f), (uint8_t)(col.z / n_col + 0.5f), 0);
    } else if (n_out > 0 || n_ring > 0) {
        t = 1;
        o = make_uchar4((uint8_t)(anc.x / n_anchor + 0.5f), (uint8_t)(anc.y / n_anchor + 0.5f), (uint8_t)(anc.z / n_anchor + 0.5f), 0);
    } else if (n_anchor > 0) {
        o = make_uchar4((uint8_t)(anc.x / n_anchor + 0.5f), (uint8_t)(anc.y / n_anchor + 0.5f), (uint8_t)(anc.z / n_anchor + 0.5f), 0);
    }
    const size_t k = (size_t)y * cw2 + x;
    type_c[k] = t;
    ext_c[k] = o;
}

// bilinear prolongation of the coarse solution as the fine initial guess
__global__ __launch_bounds__(256) void k_prolong(const float4 *__restrict__ Xc, const uint8_t *__restrict__ type_c,
                                                 float4 *X, const uint8_t *__restrict__ type, int cw, int ch,
                                                 int cw2, int ch2)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cw || y >= ch)
        return;
    const size_t ii = (size_t)y * cw + x;
    if (type[ii] == 0) { X[ii] = make_float4(0, 0, 0, 0); return; }
    const float fx = (x + 0.5f) * 0.25f - 0.5f, fy = (y + 0.5f) * 0.25f - 0.5f;
    const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    const float a = fx - x0, b = fy - y0;
    float4 acc = make_float4(0, 0, 0, 0);
    float wsum = 0;
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 2; ++i) {
            const int qx = min(max(x0 + i, 0), cw2 - 1), qy = min(max(y0 + j, 0), ch2 - 1);
            const size_t k = (size_t)qy * cw2 + qx;
            if (type_c[k] == 0)
                continue;
            const float wgt = (i ? a : 1 - a) * (j ? b : 1 - b);
            const float4 c = Xc[k];
            acc.x += wgt * c.x; acc.y += wgt * c.y; acc.z += wgt * c.z;
            wsum += wgt;
        }
    if (wsum > 1e-6f)
        X[ii] = make_float4(acc.x / wsum, acc.y / wsum, acc.z / wsum, 0);
    // else: keep the default guess written by k_setup
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

void vm_poisson_launch_prepare(uchar4 *ext, uint8_t *type, const uchar4 *other, const float2 *v,
                               int w, int h, int rs, int ex, int sign, hipStream_t s)
{
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    hipLaunchKernelGGL(k_classify, grid2(cw, ch), B2, 0, s, ext, type, cw, ch);
    hipLaunchKernelGGL(k_fill, grid2(cw, ch), B2, 0, s, ext, type, other, v, w, h, rs, ex, sign);
}

void vm_poisson_launch_setup(const uchar4 *ext, const uint8_t *type, float4 *B, float4 *X, int cw, int ch,
                             hipStream_t s)
{
    hipLaunchKernelGGL(k_setup<float4>, grid2(cw, ch), B2, 0, s, ext, type, B, X, cw, ch, 1);
}

void vm_poisson_launch_setup3(const uchar4 *ext, const uint8_t *type, VmV3 *B, VmV3 *X, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_setup<VmV3>, grid2(cw, ch), B2, 0, s, ext, type, B, X, cw, ch, 1);
}

void vm_poisson_launch_cg_init(const float4 *B, const float4 *X, float4 *R, float4 *P, const uint8_t *type,
                               VmCgScalars *sc, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_cg_init, grid2(cw, ch), B2, 0, s, B, X, R, P, type, sc, cw, ch);
}

void vm_poisson_launch_coarsen(const uchar4 *ext, const uint8_t *type, uchar4 *ext_c, uint8_t *type_c, int cw,
                               int ch, int cw2, int ch2, hipStream_t s)
{
    hipLaunchKernelGGL(k_coarsen, grid2(cw2, ch2), B2, 0, s, ext, type, ext_c, type_c, cw, ch, cw2, ch2);
}

void vm_poisson_launch_prolong(const float4 *Xc, const uint8_t *type_c, float4 *X, const uint8_t *type, int cw,
                               int ch, int cw2, int ch2, hipStream_t s)
{
    hipLaunchKernelGGL(k_prolong, grid2(cw, ch), B2, 0, s, Xc, type_c, X, type, cw, ch, cw2, ch2);
}

void vm_poisson_launch_iter(float4 *X, float4 *R, float4 *P, float4 *Q, const float4 *B,
                            const uint8_t *type, VmCgScalars *sc, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_cg_spmv, grid2(cw, ch), B2, 0, s, P, Q, B, type, sc, cw, ch);
    hipLaunchKernelGGL(k_cg_update, grid2(cw, ch), B2, 0, s, X, R, P, Q, B, type, sc, cw, ch);
    hipLaunchKernelGGL(k_cg_dir, grid2(cw, ch), B2, 0, s, P, R, B, type, sc, cw, ch);
    hipLaunchKernelGGL(k_cg_rotate, dim3(1), dim3(64), 0, s, sc);
}

void vm_poisson_launch_paste(uchar4 *ext, const uint8_t *type, const float4 *X, int cw, int ch,
                             hipStream_t s)
{
    hipLaunchKernelGGL(k_paste<float4>, grid2(cw, ch), B2, 0, s, ext, type, X, cw, ch);
}

void vm_poisson_launch_paste3(uchar4 *ext, const uint8_t *type, const VmV3 *X, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_paste<VmV3>, grid2(cw, ch), B2, 0, s, ext, type, X, cw, ch);
}

void vm_qpath_launch_rhs(const float2 *v, int rs, int w, int h, float4 *B, float4 *X, hipStream_t s)
{
    hipLaunchKernelGGL(k_qp_rhs<float4>, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, s, v, rs, w, h, B, X);
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

void vm_qpath_launch_sum(const float4 *X, int w, int h, double *sums, hipStream_t s)
{
    hipLaunchKernelGGL(k_qp_sum<float4>, sum_grid(w, h), dim3(64, 4), 0, s, X, w, h, sums, 1);
}

// ... spread over VM_QP_SLOTS lines of 16 doubles (the batched solver's scalar block has room for 8)
void vm_qpath_launch_sum3(const VmV3 *X, int w, int h, double *sums, hipStream_t s)
{
    hipLaunchKernelGGL(k_qp_sum<VmV3>, sum_grid(w, h), dim3(64, 4), 0, s, X, w, h, sums, VM_QP_SLOTS);
}

// u == nullptr: X -= mean in place; else u = X - mean
void vm_qpath_launch_shift(float4 *X, int w, int h, const double *sums, float2 *u, int rs, hipStream_t s)
{
    hipLaunchKernelGGL(k_qp_shift<float4>, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, s, X, w, h, sums, 1, u, rs);
}

void vm_qpath_launch_shift3(VmV3 *X, int w, int h, const double *sums, float2 *u, int rs, hipStream_t s)
{
    hipLaunchKernelGGL(k_qp_shift<VmV3>, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, s, X, w, h, sums, VM_QP_SLOTS, u, rs);
}
