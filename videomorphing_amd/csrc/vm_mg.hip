// vm_mg.hip -- multigrid-preconditioned CG for the compositor's 5-point systems (gfx950).
//
// The reference solves these systems with a CPU sparse direct solver (MKL DSS,
// PoissonExt.cpp:321-329) resp. 10 000 unpreconditioned CG iterations through cuSPARSE
// (QuadraticPath.cpp:226-305).  Here: conjugate gradients preconditioned by one V(1,1) cycle
// of an aggregation multigrid -- 2x2 blocks, piecewise-constant interpolation P, Galerkin
// coarse operators P^T A P with the edge weights rescaled by 1/2, damped Jacobi (omega 0.8)
// before and after, 40 Jacobi sweeps on the <= 1024-cell coarsest grid.  Every piece is a
// symmetric operation, so the cycle is an SPD preconditioner.  Measured: 14 / 18 / 21
// iterations to a relative residual of 1e-4 / 1e-5 / 1e-6 on the 2304x1464 canvas of a
// 1080p frame, against 192 / 832 / - with the Jacobi preconditioner + nested iteration.
// All kernels are HBM-bound streams over the canvas (one thread per pixel, 64x4 blocks).
#include "vm_mg.h"

namespace {

__device__ __forceinline__ float4 f4_axpy(float a, float4 x, float4 y) // a x + y
{
    return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), 0);
}

// (A u)(x, y) for an unknown cell with diagonal dg
__device__ __forceinline__ float4 mg_apply(const VmMgLevel &L, const float4 *__restrict__ u, int x, int y, size_t ii,
                                           float dg)
{
    const float4 c = u[ii];
    float4 s = make_float4(dg * c.x, dg * c.y, dg * c.z, 0);
    if (x + 1 < L.w) {
        const float wgt = L.we[ii];
        if (wgt != 0.0f) s = f4_axpy(-wgt, u[ii + 1], s);
    }
    if (x > 0) {
        const float wgt = L.we[ii - 1];
        if (wgt != 0.0f) s = f4_axpy(-wgt, u[ii - 1], s);
    }
    if (y + 1 < L.h) {
        const float wgt = L.ws[ii];
        if (wgt != 0.0f) s = f4_axpy(-wgt, u[ii + L.w], s);
    }
    if (y > 0) {
        const float wgt = L.ws[ii - L.w];
        if (wgt != 0.0f) s = f4_axpy(-wgt, u[ii - L.w], s);
    }
    return s;
}

__global__ __launch_bounds__(256) void k_level0_type(const uint8_t *__restrict__ type, VmMgLevel L)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= L.w || y >= L.h)
        return;
    const size_t ii = (size_t)y * L.w + x;
    const uint8_t t = type[ii];
    float we = 0, ws = 0, dg = 0;
    if (t > 0) {
        dg = t == 1 ? 1.0f : 0.0f; // PoissonExt.cpp:222-228: ring pixels are tied to their colour
        if (x + 1 < L.w && type[ii + 1] > 0) { we = 1; dg += 1; }
        if (y + 1 < L.h && type[ii + L.w] > 0) { ws = 1; dg += 1; }
        if (x > 0 && type[ii - 1] > 0) dg += 1;
        if (y > 0 && type[ii - L.w] > 0) dg += 1;
    }
    L.we[ii] = we;
    L.ws[ii] = ws;
    L.dg[ii] = dg;
}

__global__ __launch_bounds__(256) void k_level0_full(VmMgLevel L)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= L.w || y >= L.h)
        return;
    const size_t ii = (size_t)y * L.w + x;
    L.we[ii] = x + 1 < L.w ? 1.0f : 0.0f;
    L.ws[ii] = y + 1 < L.h ? 1.0f : 0.0f;
    L.dg[ii] = (float)((x + 1 < L.w) + (x > 0) + (y + 1 < L.h) + (y > 0));
}

// coarse weights and, in C.dg for the moment, the coarse screening
__global__ __launch_bounds__(256) void k_coarsen(VmMgLevel F, VmMgLevel C)
{
    const int X = blockIdx.x * 64 + threadIdx.x, Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= C.w || Y >= C.h)
        return;
    float we = 0, ws = 0, sc = 0;
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
            const int x = 2 * X + a, y = 2 * Y + b;
            if (x >= F.w || y >= F.h)
                continue;
            const size_t ii = (size_t)y * F.w + x;
            const float e = F.we[ii], s = F.ws[ii];
            float inc = e + s;
            if (x > 0) inc += F.we[ii - 1];
            if (y > 0) inc += F.ws[ii - F.w];
            sc += F.dg[ii] - inc; // screening = diagonal - incident weights
            if (a == 1) we += e;  // edges leaving the block to the east / south
            if (b == 1) ws += s;
        }
    const size_t k = (size_t)Y * C.w + X;
    C.we[k] = 0.5f * we;
    C.ws[k] = 0.5f * ws;
    C.dg[k] = fmaxf(sc, 0.0f);
}

__global__ __launch_bounds__(256) void k_diag(VmMgLevel C, float *__restrict__ out)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= C.w || y >= C.h)
        return;
    const size_t ii = (size_t)y * C.w + x;
    float d = C.dg[ii] + C.we[ii] + C.ws[ii];
    if (x > 0) d += C.we[ii - 1];
    if (y > 0) d += C.ws[ii - C.w];
    out[ii] = d;
}

__global__ __launch_bounds__(256) void k_jacobi0(VmMgLevel L, float omega)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= L.w || y >= L.h)
        return;
    const size_t ii = (size_t)y * L.w + x;
    const float dg = L.dg[ii];
    float4 o = make_float4(0, 0, 0, 0);
    if (dg > 0) {
        const float k = omega / dg;
        const float4 b = L.b[ii];
        o = make_float4(k * b.x, k * b.y, k * b.z, 0);
    }
    L.x[ii] = o;
}

__global__ __launch_bounds__(256) void k_resid_restrict(VmMgLevel F, VmMgLevel C)
{
    const int X = blockIdx.x * 64 + threadIdx.x, Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= C.w || Y >= C.h)
        return;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) {
            const int x = 2 * X + a, y = 2 * Y + b;
            if (x >= F.w || y >= F.h)
                continue;
            const size_t ii = (size_t)y * F.w + x;
            const float dg = F.dg[ii];
            if (!(dg > 0))
                continue;
            const float4 ax = mg_apply(F, F.x, x, y, ii, dg), bb = F.b[ii];
            acc.x += bb.x - ax.x;
            acc.y += bb.y - ax.y;
            acc.z += bb.z - ax.z;
        }
    C.b[(size_t)Y * C.w + X] = acc;
}

__global__ __launch_bounds__(256) void k_prolong_smooth(VmMgLevel F, VmMgLevel C, float omega)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= F.w || y >= F.h)
        return;
    const size_t ii = (size_t)y * F.w + x;
    const float dg = F.dg[ii];
    if (!(dg > 0)) {
        F.t[ii] = make_float4(0, 0, 0, 0);
        return;
    }
    // x1 = x + P xc at the cell and its neighbours (a neighbour without an edge is never used)
#define X1(QX, QY) ({                                                                \
        const float4 f_ = F.x[(size_t)(QY) * F.w + (QX)];                             \
        const float4 c_ = C.x[(size_t)((QY) >> 1) * C.w + ((QX) >> 1)];              \
        make_float4(f_.x + c_.x, f_.y + c_.y, f_.z + c_.z, 0); })
    const float4 c = X1(x, y);
    float4 s = make_float4(dg * c.x, dg * c.y, dg * c.z, 0);
    if (x + 1 < F.w) {
        const float wgt = F.we[ii];
        if (wgt != 0.0f) s = f4_axpy(-wgt, X1(x + 1, y), s);
    }
    if (x > 0) {
        const float wgt = F.we[ii - 1];
        if (wgt != 0.0f) s = f4_axpy(-wgt, X1(x - 1, y), s);
    }
    if (y + 1 < F.h) {
        const float wgt = F.ws[ii];
        if (wgt != 0.0f) s = f4_axpy(-wgt, X1(x, y + 1), s);
    }
    if (y > 0) {
        const float wgt = F.ws[ii - F.w];
        if (wgt != 0.0f) s = f4_axpy(-wgt, X1(x, y - 1), s);
    }
#undef X1
    const float4 b = F.b[ii];
    const float k = omega / dg;
    F.t[ii] = make_float4(c.x + k * (b.x - s.x), c.y + k * (b.y - s.y), c.z + k * (b.z - s.z), 0);
}

__global__ __launch_bounds__(1024) void k_coarsest(VmMgLevel L, float omega, int sweeps)
{
    __shared__ float4 xa[1024], xb[1024];
    const int t = threadIdx.x, n = L.w * L.h;
    const int x = t % L.w, y = t / L.w;
    float dg = 0, wE = 0, wW = 0, wS = 0, wN = 0;
    float4 b = make_float4(0, 0, 0, 0);
    if (t < n) {
        dg = L.dg[t];
        b = L.b[t];
        if (x + 1 < L.w) wE = L.we[t];
        if (x > 0) wW = L.we[t - 1];
        if (y + 1 < L.h) wS = L.ws[t];
        if (y > 0) wN = L.ws[t - L.w];
    }
    const float k = dg > 0 ? omega / dg : 0.0f;
    float4 cur = make_float4(k * b.x, k * b.y, k * b.z, 0);
    float4 *src = xa, *dst = xb;
    src[t] = cur;
    __syncthreads();
    for (int it = 1; it < sweeps; ++it) {
        if (t < n && dg > 0) {
            float4 s = make_float4(dg * cur.x, dg * cur.y, dg * cur.z, 0);
            if (wE != 0.0f) s = f4_axpy(-wE, src[t + 1], s);
            if (wW != 0.0f) s = f4_axpy(-wW, src[t - 1], s);
            if (wS != 0.0f) s = f4_axpy(-wS, src[t + L.w], s);
            if (wN != 0.0f) s = f4_axpy(-wN, src[t - L.w], s);
            cur = make_float4(cur.x + k * (b.x - s.x), cur.y + k * (b.y - s.y), cur.z + k * (b.z - s.z), 0);
        }
        dst[t] = cur;
        __syncthreads();
        float4 *tmp = src;
        src = dst;
        dst = tmp;
    }
    if (t < n)
        L.x[t] = cur;
}

// The two coarsest grids of the cycle in ONE workgroup (F: at most 4096 cells, C: the coarsest,
// at most 1024): pre-smoothing of F, residual restriction, the Jacobi sweeps on C, coarse
// correction + post-smoothing of F -- what k_jacobi0, k_resid_restrict, k_coarsest and
// k_prolong_smooth do in four launches of a few workgroups each (on the 2304x1464 canvas: grids
// 72x46 and 36x23, launch-bound).  F's iterate stays in LDS throughout.  Same arithmetic.
__global__ __launch_bounds__(1024) void k_coarse_tail(VmMgLevel F, VmMgLevel C, float omega, int sweeps)
{
    __shared__ float4 xf[4096], xa[1024], xb[1024];
    const int t = threadIdx.x, nF = F.w * F.h, nC = C.w * C.h;
    for (int i = t; i < nF; i += 1024) { // x = omega b / dg
        const float dg = F.dg[i];
        float4 o = make_float4(0, 0, 0, 0);
        if (dg > 0) {
            const float k = omega / dg;
            const float4 b = F.b[i];
            o = make_float4(k * b.x, k * b.y, k * b.z, 0);
        }
        xf[i] = o;
    }
    __syncthreads();
    // C.b = P^T (F.b - A x), then `sweeps` damped-Jacobi sweeps on C from zero
    const int X = t % C.w, Y = t / C.w;
    float dg = 0, wE = 0, wW = 0, wS = 0, wN = 0;
    float4 b = make_float4(0, 0, 0, 0);
    if (t < nC) {
        for (int bb = 0; bb < 2; ++bb)
            for (int a = 0; a < 2; ++a) {
                const int x = 2 * X + a, y = 2 * Y + bb;
                if (x >= F.w || y >= F.h)
                    continue;
                const size_t ii = (size_t)y * F.w + x;
                const float fdg = F.dg[ii];
                if (!(fdg > 0))
                    continue;
                const float4 ax = mg_apply(F, xf, x, y, ii, fdg), fb = F.b[ii];
                b.x += fb.x - ax.x;
                b.y += fb.y - ax.y;
                b.z += fb.z - ax.z;
            }
        dg = C.dg[t];
        if (X + 1 < C.w) wE = C.we[t];
        if (X > 0) wW = C.we[t - 1];
        if (Y + 1 < C.h) wS = C.ws[t];
        if (Y > 0) wN = C.ws[t - C.w];
    }
    const float k = dg > 0 ? omega / dg : 0.0f;
    float4 cur = make_float4(k * b.x, k * b.y, k * b.z, 0);
    float4 *src = xa, *dst = xb;
    src[t] = cur;
    __syncthreads();
    for (int it = 1; it < sweeps; ++it) {
        if (t < nC && dg > 0) {
            float4 s = make_float4(dg * cur.x, dg * cur.y, dg * cur.z, 0);
            if (wE != 0.0f) s = f4_axpy(-wE, src[t + 1], s);
            if (wW != 0.0f) s = f4_axpy(-wW, src[t - 1], s);
            if (wS != 0.0f) s = f4_axpy(-wS, src[t + C.w], s);
            if (wN != 0.0f) s = f4_axpy(-wN, src[t - C.w], s);
            cur = make_float4(cur.x + k * (b.x - s.x), cur.y + k * (b.y - s.y), cur.z + k * (b.z - s.z), 0);
        }
        dst[t] = cur;
        __syncthreads();
        float4 *tmp = src;
        src = dst;
        dst = tmp;
    }
    // x1 = x + P xc (in place), then F.t = x1 + omega (F.b - A x1) / dg
    for (int i = t; i < nF; i += 1024) {
        const int x = i % F.w, y = i / F.w;
        const float4 c = src[(y >> 1) * C.w + (x >> 1)], f = xf[i];
        xf[i] = make_float4(f.x + c.x, f.y + c.y, f.z + c.z, 0);
    }
    __syncthreads();
    for (int i = t; i < nF; i += 1024) {
        const int x = i % F.w, y = i / F.w;
        const float fdg = F.dg[i];
        float4 o = make_float4(0, 0, 0, 0);
        if (fdg > 0) {
            const float4 c = xf[i], s = mg_apply(F, xf, x, y, (size_t)i, fdg), fb = F.b[i];
            const float kf = omega / fdg;
            o = make_float4(c.x + kf * (fb.x - s.x), c.y + kf * (fb.y - s.y), c.z + kf * (fb.z - s.z), 0);
        }
        F.t[i] = o;
    }
}

// ---------------------------------------------------------------------------
// PCG on level 0.  The kernels with a dot product give each thread VM_PCG_ROWS rows of the
// 64x4 footprint: every block ends in three same-address double atomics, which the L2
// serialises at ~5 ns each -- with one footprint per block (13 k blocks on the 1080p canvas)
// that alone took 65 us per kernel, more than the streaming.
#define VM_PCG_ROWS 16

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
        const double s = sh[tid][0] + sh[tid][1] + sh[tid][2] + sh[tid][3];
        if (s != 0) atomicAdd(&dst[tid], s);
    }
    __syncthreads();
}

// r = b - A x;  bb = b.b, rr = r.r
__global__ __launch_bounds__(256) void k_pcg_init(VmMgLevel L, const float4 *__restrict__ B,
                                                  const float4 *__restrict__ X, float4 *R, VmPcgScalars *sc)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    double bb[3] = {0, 0, 0}, rr[3] = {0, 0, 0};
    for (int k = 0; k < VM_PCG_ROWS; ++k) {
        const int y = (blockIdx.y * VM_PCG_ROWS + k) * 4 + threadIdx.y;
        if (x < L.w && y < L.h) {
            const size_t ii = (size_t)y * L.w + x;
            const float dg = L.dg[ii];
            float4 r = make_float4(0, 0, 0, 0);
            if (dg > 0) {
                const float4 b = B[ii], ax = mg_apply(L, X, x, y, ii, dg);
                r = make_float4(b.x - ax.x, b.y - ax.y, b.z - ax.z, 0);
                bb[0] += (double)b.x * b.x; bb[1] += (double)b.y * b.y; bb[2] += (double)b.z * b.z;
                rr[0] += (double)r.x * r.x; rr[1] += (double)r.y * r.y; rr[2] += (double)r.z * r.z;
            }
            R[ii] = r;
        }
    }
    block_sum3(bb[0], bb[1], bb[2], sc->bb);
    block_sum3(rr[0], rr[1], rr[2], sc->rr);
}

// q = A p;  pq += p.q
__global__ __launch_bounds__(256) void k_pcg_spmv(VmMgLevel L, const float4 *__restrict__ P, float4 *Q,
                                                  VmPcgScalars *sc)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    double pq[3] = {0, 0, 0};
    for (int k = 0; k < VM_PCG_ROWS; ++k) {
        const int y = (blockIdx.y * VM_PCG_ROWS + k) * 4 + threadIdx.y;
        if (x < L.w && y < L.h) {
            const size_t ii = (size_t)y * L.w + x;
            const float dg = L.dg[ii];
            if (dg > 0) {
                const float4 q = mg_apply(L, P, x, y, ii, dg), p = P[ii];
                Q[ii] = q;
                pq[0] += (double)p.x * q.x; pq[1] += (double)p.y * q.y; pq[2] += (double)p.z * q.z;
            }
        }
    }
    block_sum3(pq[0], pq[1], pq[2], sc->pq);
}

// alpha = rz / pq;  x += alpha p;  r -= alpha q;  rr = r.r  (rr was zeroed by k_pcg_dir)
__global__ __launch_bounds__(256) void k_pcg_update(VmMgLevel L, float4 *X, float4 *R, const float4 *__restrict__ P,
                                                    const float4 *__restrict__ Q, VmPcgScalars *sc)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    double rr[3] = {0, 0, 0};
    float al[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        al[c] = sc->pq[c] > 0 ? (float)(sc->rz[c] / sc->pq[c]) : 0.0f;
    for (int k = 0; k < VM_PCG_ROWS; ++k) {
        const int y = (blockIdx.y * VM_PCG_ROWS + k) * 4 + threadIdx.y;
        if (x < L.w && y < L.h) {
            const size_t ii = (size_t)y * L.w + x;
            if (L.dg[ii] > 0) {
                const float4 p = P[ii], q = Q[ii];
                float4 xx = X[ii], r = R[ii];
                xx.x += al[0] * p.x; xx.y += al[1] * p.y; xx.z += al[2] * p.z;
                r.x -= al[0] * q.x; r.y -= al[1] * q.y; r.z -= al[2] * q.z;
                X[ii] = xx;
                R[ii] = r;
                rr[0] += (double)r.x * r.x; rr[1] += (double)r.y * r.y; rr[2] += (double)r.z * r.z;
            }
        }
    }
    block_sum3(rr[0], rr[1], rr[2], sc->rr);
}

// rz_new = r.z
__global__ __launch_bounds__(256) void k_pcg_dot(VmMgLevel L, const float4 *__restrict__ R,
                                                 const float4 *__restrict__ Z, VmPcgScalars *sc)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    double rz[3] = {0, 0, 0};
    for (int k = 0; k < VM_PCG_ROWS; ++k) {
        const int y = (blockIdx.y * VM_PCG_ROWS + k) * 4 + threadIdx.y;
        if (x < L.w && y < L.h) {
            const size_t ii = (size_t)y * L.w + x;
            if (L.dg[ii] > 0) {
                const float4 r = R[ii], z = Z[ii];
                rz[0] += (double)r.x * z.x; rz[1] += (double)r.y * z.y; rz[2] += (double)r.z * z.z;
            }
        }
    }
    block_sum3(rz[0], rz[1], rz[2], sc->rz_new);
}

// beta = rz_new / rz (0 in the first iteration);  p = z + beta p
__global__ __launch_bounds__(256) void k_pcg_dir(VmMgLevel L, float4 *P, const float4 *__restrict__ Z,
                                                 const VmPcgScalars *sc, int first)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= L.w || y >= L.h)
        return;
    const size_t ii = (size_t)y * L.w + x;
    float4 p = make_float4(0, 0, 0, 0);
    if (L.dg[ii] > 0) {
        float be[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            be[c] = (!first && sc->rz[c] > 0) ? (float)(sc->rz_new[c] / sc->rz[c]) : 0.0f;
        const float4 z = Z[ii];
        const float4 po = first ? make_float4(0, 0, 0, 0) : P[ii];
        p = make_float4(z.x + be[0] * po.x, z.y + be[1] * po.y, z.z + be[2] * po.z, 0);
    }
    P[ii] = p;
}

// rz <- rz_new; clear the accumulators of the next iteration
__global__ void k_pcg_rotate(VmPcgScalars *sc)
{
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        sc->rz[c] = sc->rz_new[c];
        sc->rz_new[c] = 0;
        sc->pq[c] = 0;
    }
}

// rr is accumulated by k_pcg_update: cleared just before it
__global__ void k_pcg_clear_rr(VmPcgScalars *sc)
{
    if (threadIdx.x < 3)
        sc->rr[threadIdx.x] = 0;
}

dim3 grid2(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }
dim3 grid_pcg(int w, int h) { return dim3((w + 63) / 64, (h + 4 * VM_PCG_ROWS - 1) / (4 * VM_PCG_ROWS)); }
const dim3 blk2(64, 4);

} // namespace

void vm_mg_launch_level0_type(const uint8_t *type, const VmMgLevel &L, hipStream_t s)
{
    hipLaunchKernelGGL(k_level0_type, grid2(L.w, L.h), blk2, 0, s, type, L);
}

void vm_mg_launch_level0_full(const VmMgLevel &L, hipStream_t s)
{
    hipLaunchKernelGGL(k_level0_full, grid2(L.w, L.h), blk2, 0, s, L);
}

void vm_mg_launch_coarsen(const VmMgLevel &F, const VmMgLevel &C, hipStream_t s)
{
    hipLaunchKernelGGL(k_coarsen, grid2(C.w, C.h), blk2, 0, s, F, C);
    // screening -> diagonal; C.t is free at set-up time and serves as the output buffer
    hipLaunchKernelGGL(k_diag, grid2(C.w, C.h), blk2, 0, s, C, (float *)C.t);
    (void)hipMemcpyAsync(C.dg, C.t, (size_t)C.w * C.h * sizeof(float), hipMemcpyDeviceToDevice, s);
}

void vm_mg_launch_jacobi0(const VmMgLevel &L, float omega, hipStream_t s)
{
    hipLaunchKernelGGL(k_jacobi0, grid2(L.w, L.h), blk2, 0, s, L, omega);
}

void vm_mg_launch_resid_restrict(const VmMgLevel &F, const VmMgLevel &C, hipStream_t s)
{
    hipLaunchKernelGGL(k_resid_restrict, grid2(C.w, C.h), blk2, 0, s, F, C);
}

void vm_mg_launch_prolong_smooth(const VmMgLevel &F, const VmMgLevel &C, float omega, hipStream_t s)
{
    hipLaunchKernelGGL(k_prolong_smooth, grid2(F.w, F.h), blk2, 0, s, F, C, omega);
}

void vm_mg_launch_coarsest(const VmMgLevel &L, float omega, int sweeps, hipStream_t s)
{
    hipLaunchKernelGGL(k_coarsest, dim3(1), dim3(1024), 0, s, L, omega, sweeps);
}

void vm_mg_launch_coarse_tail(const VmMgLevel &F, const VmMgLevel &C, float omega, int sweeps, hipStream_t s)
{
    hipLaunchKernelGGL(k_coarse_tail, dim3(1), dim3(1024), 0, s, F, C, omega, sweeps);
}

void vm_mg_launch_pcg_init(const VmMgLevel &L, const float4 *B, const float4 *X, float4 *R, VmPcgScalars *sc,
                           hipStream_t s)
{
    hipLaunchKernelGGL(k_pcg_init, grid_pcg(L.w, L.h), blk2, 0, s, L, B, X, R, sc);
}

void vm_mg_launch_pcg_spmv(const VmMgLevel &L, const float4 *P, float4 *Q, VmPcgScalars *sc, hipStream_t s)
{
    hipLaunchKernelGGL(k_pcg_spmv, grid_pcg(L.w, L.h), blk2, 0, s, L, P, Q, sc);
}

void vm_mg_launch_pcg_update(const VmMgLevel &L, float4 *X, float4 *R, const float4 *P, const float4 *Q,
                             VmPcgScalars *sc, hipStream_t s)
{
    hipLaunchKernelGGL(k_pcg_clear_rr, dim3(1), dim3(64), 0, s, sc);
    hipLaunchKernelGGL(k_pcg_update, grid_pcg(L.w, L.h), blk2, 0, s, L, X, R, P, Q, sc);
}

void vm_mg_launch_pcg_dot(const VmMgLevel &L, const float4 *R, const float4 *Z, VmPcgScalars *sc, hipStream_t s)
{
    hipLaunchKernelGGL(k_pcg_dot, grid_pcg(L.w, L.h), blk2, 0, s, L, R, Z, sc);
}

void vm_mg_launch_pcg_dir(const VmMgLevel &L, float4 *P, const float4 *Z, VmPcgScalars *sc, int first,
                          hipStream_t s)
{
    hipLaunchKernelGGL(k_pcg_dir, grid2(L.w, L.h), blk2, 0, s, L, P, Z, sc, first);
    hipLaunchKernelGGL(k_pcg_rotate, dim3(1), dim3(64), 0, s, sc);
}
