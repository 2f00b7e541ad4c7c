// vm_mgb.hip -- batched, ring-only, fused multigrid-preconditioned CG of the Poisson extension (gfx950).
// See vm_mgb.h for the design; vm_mg.hip holds the one-system, whole-canvas form it grew from (still the
// solver of the quadratic motion path, whose unknowns are the whole grid).  Every kernel: blockIdx.z = system,
// a workgroup = 64 x 4 threads = one block of the level's compact block list (or MGB_G of them where the kernel
// ends in a dot product); HBM-bound streams over the ring of unknowns.
#include "vm_mgb.h"

namespace {

#define MGB_G 4 // list entries per workgroup in the kernels that end in dot products (their atomics per byte / 4)

__device__ __forceinline__ float4 f4_axpy(float a, float4 x, float4 y) // a x + y
{
    return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), 0);
}

__device__ __forceinline__ bool sys_active(uint64_t active) { return (active >> blockIdx.z) & 1; }

// the cell this thread owns in entry `e` of the level's block list
__device__ __forceinline__ bool list_cell(const VmMgbLevel &L, int e, int nb, int &x, int &y)
{
    if (e >= nb)
        return false;
    const uint32_t b = L.blocks[e];
    x = (int)(b & 0xffffu) * 64 + (int)threadIdx.x;
    y = (int)(b >> 16) * 4 + (int)threadIdx.y;
    return x < L.w && y < L.h;
}

// (A u)(x, y) for an unknown cell with diagonal dg; u(q) by functor (E, W, S, N: the order of vm_mg.hip's mg_apply)
template <class U>
__device__ __forceinline__ float4 apply(const VmMgbLevel &L, const U &u, int x, int y, size_t ii, float dg)
{
    const float4 c = u(ii, x, y);
    float4 s = make_float4(dg * c.x, dg * c.y, dg * c.z, 0);
    if (x + 1 < L.w) {
        const float wgt = L.we[ii];
        if (wgt != 0.0f) s = f4_axpy(-wgt, u(ii + 1, x + 1, y), s);
    }
    if (x > 0) {
        const float wgt = L.we[ii - 1];
        if (wgt != 0.0f) s = f4_axpy(-wgt, u(ii - 1, x - 1, y), s);
    }
    if (y + 1 < L.h) {
        const float wgt = L.ws[ii];
        if (wgt != 0.0f) s = f4_axpy(-wgt, u(ii + L.w, x, y + 1), s);
    }
    if (y > 0) {
        const float wgt = L.ws[ii - L.w];
        if (wgt != 0.0f) s = f4_axpy(-wgt, u(ii - L.w, x, y - 1), s);
    }
    return s;
}

struct FromArray {
    const float4 *__restrict__ a;
    __device__ __forceinline__ float4 operator()(size_t q, int = 0, int = 0) const { return a[q]; }
};

// the pre-smoothed iterate of a level, never stored: x = omega b / dg (damped Jacobi from zero)
struct PreSmoothed {
    const float *__restrict__ dg;
    const float4 *__restrict__ b;
    float omega;
    __device__ __forceinline__ float4 operator()(size_t q, int = 0, int = 0) const
    {
        const float d = dg[q];
        if (!(d > 0))
            return make_float4(0, 0, 0, 0);
        const float k = omega / d;
        const float4 v = b[q];
        return make_float4(k * v.x, k * v.y, k * v.z, 0);
    }
};

// ... plus the coarse correction: x1 = x + P xc
struct Corrected {
    PreSmoothed pre;
    const float4 *__restrict__ xc;
    int cw;
    __device__ __forceinline__ float4 operator()(size_t q, int x, int y) const
    {
        const float4 f = pre(q), c = xc[(size_t)(y >> 1) * cw + (x >> 1)];
        return make_float4(f.x + c.x, f.y + c.y, f.z + c.z, 0);
    }
};

// block reduction of three doubles, then one double atomic per block and channel into the block's slot
__device__ __forceinline__ void block_sum3(double a, double b, double c, double (*dst)[16])
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
        if (s != 0) atomicAdd(&dst[blockIdx.x % VM_MGB_SLOTS][tid], s);
    }
    __syncthreads();
}

__device__ __forceinline__ double slot_sum(const double (*a)[16], int c)
{
    double s = 0;
#pragma unroll
    for (int k = 0; k < VM_MGB_SLOTS; ++k)
        s += a[k][c];
    return s;
}

__device__ __forceinline__ void slot_clear(double (*a)[16])
{
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (tid < VM_MGB_SLOTS * 3)
        a[tid / 3][tid % 3] = 0;
}

// ---------------------------------------------------------------------------
// set-up

// level 0 from the type map (PoissonExt.cpp:214-312: unknown <=> type > 0, ring pixels tied to their colour)
__global__ __launch_bounds__(256) void k_mgb_level0(const VmMgbSys *__restrict__ sys)
{
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &L = S.lv[0];
    const uint8_t *__restrict__ type = S.type;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    float dg = 0;
    if (x < L.w && y < L.h) {
        const size_t ii = (size_t)y * L.w + x;
        const uint8_t t = type[ii];
        float we = 0, ws = 0;
        if (t > 0) {
            dg = t == 1 ? 1.0f : 0.0f;
            if (x + 1 < L.w && type[ii + 1] > 0) { we = 1; dg += 1; }
            if (y + 1 < L.h && type[ii + L.w] > 0) { ws = 1; dg += 1; }
            if (x > 0 && type[ii - 1] > 0) dg += 1;
            if (y > 0 && type[ii - L.w] > 0) dg += 1;
        }
        L.we[ii] = we;
        L.ws[ii] = ws;
        L.dg[ii] = dg;
    }
    const int any = __syncthreads_or(dg > 0);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        L.flags[blockIdx.y * L.gx + blockIdx.x] = any ? 1u : 0u;
}

// Galerkin coarse operator of level l from level l - 1 (2x2 aggregates, piecewise-constant interpolation, the
// edge weights rescaled by 1/2), the diagonal in the same pass: the west / north weights of a coarse cell are
// those of fine edges that enter its block
__global__ __launch_bounds__(256) void k_mgb_coarsen(const VmMgbSys *__restrict__ sys, int l)
{
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &F = S.lv[l - 1], &C = S.lv[l];
    const int X = blockIdx.x * 64 + threadIdx.x, Y = blockIdx.y * 4 + threadIdx.y;
    float d = 0;
    if (X < C.w && Y < C.h) {
        float we = 0, ws = 0, sc = 0, ww = 0, wn = 0;
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
                if (a == 0 && x > 0) ww += F.we[ii - 1];   // ... and entering it from the west / north
                if (b == 0 && y > 0) wn += F.ws[ii - F.w];
            }
        const size_t k = (size_t)Y * C.w + X;
        we *= 0.5f; ws *= 0.5f; ww *= 0.5f; wn *= 0.5f;
        C.we[k] = we;
        C.ws[k] = ws;
        d = fmaxf(sc, 0.0f) + we + ws;
        if (X > 0) d += ww;
        if (Y > 0) d += wn;
        C.dg[k] = d;
    }
    const int any = __syncthreads_or(d > 0);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        C.flags[blockIdx.y * C.gx + blockIdx.x] = any ? 1u : 0u;
}

// block flags -> compact row-major block list, one workgroup per (level, system)
__global__ __launch_bounds__(1024) void k_mgb_compact(const VmMgbSys *__restrict__ sys)
{
    const VmMgbSys &S = sys[blockIdx.z];
    const int l = blockIdx.x;
    if (l >= S.nlev)
        return;
    const VmMgbLevel &L = S.lv[l];
    __shared__ int wcount[16];
    __shared__ int base;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, n = L.gx * L.gy;
    if (t == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + t;
        const bool f = i < n && L.flags[i] != 0;
        const unsigned long long m = __ballot(f);
        if (lane == 0) wcount[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int k = 0; k < wave; ++k) off += wcount[k];
        if (f)
            L.blocks[off + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)(i % L.gx) | ((uint32_t)(i / L.gx) << 16);
        __syncthreads();
        if (t == 0) {
            int tot = 0;
            for (int k = 0; k < 16; ++k) tot += wcount[k];
            base += tot;
        }
        __syncthreads();
    }
    if (t == 0) L.nblocks[0] = base;
}

// ---------------------------------------------------------------------------
// PCG, level 0

// r = b - A x in place of b;  bb = b.b, rr[1] = r.r (the iteration "before the first")
__global__ __launch_bounds__(256) void k_mgb_init(const VmMgbSys *__restrict__ sys, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &L = S.lv[0];
    const int nb = L.nblocks[0];
    const FromArray X{S.X};
    double bb[3] = {0, 0, 0}, rr[3] = {0, 0, 0};
    for (int g = 0; g < MGB_G; ++g) {
        int x, y;
        if (!list_cell(L, blockIdx.x * MGB_G + g, nb, x, y))
            continue;
        const size_t ii = (size_t)y * L.w + x;
        const float dg = L.dg[ii];
        if (dg > 0) {
            const float4 b = L.b[ii], ax = apply(L, X, x, y, ii, dg);
            const float4 r = make_float4(b.x - ax.x, b.y - ax.y, b.z - ax.z, 0);
            bb[0] += (double)b.x * b.x; bb[1] += (double)b.y * b.y; bb[2] += (double)b.z * b.z;
            rr[0] += (double)r.x * r.x; rr[1] += (double)r.y * r.y; rr[2] += (double)r.z * r.z;
            L.b[ii] = r;
        }
    }
    block_sum3(bb[0], bb[1], bb[2], S.sc->bb);
    block_sum3(rr[0], rr[1], rr[2], S.sc->rr[1]);
}

// beta = rz_k / rz_{k-1} (0 in the first iteration);  p = z + beta p_old at the cell and its neighbours, q = A p;
// pq[k & 1] += p.q.  Clears rr[k & 1], which k_mgb_update accumulates next.
__global__ __launch_bounds__(256) void k_mgb_dirspmv(const VmMgbSys *__restrict__ sys, int k, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &L = S.lv[0];
    const int nb = L.nblocks[0], par = k & 1;
    if (blockIdx.x == 0)
        slot_clear(S.sc->rr[par]);
    float be[3] = {0, 0, 0};
    if (k > 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double prev = slot_sum(S.sc->rz[par ^ 1], c);
            be[c] = prev > 0 ? (float)(slot_sum(S.sc->rz[par], c) / prev) : 0.0f;
        }
    }
    const float4 *__restrict__ Z = L.x;
    const float4 *__restrict__ Po = S.P[par ^ 1];
    float4 *Pn = S.P[par], *Q = S.Q;
    const bool first = k == 0;
    auto pnew = [&](size_t q, int = 0, int = 0) {
        const float4 z = Z[q];
        if (first)
            return make_float4(z.x, z.y, z.z, 0.0f);
        const float4 po = Po[q];
        return make_float4(z.x + be[0] * po.x, z.y + be[1] * po.y, z.z + be[2] * po.z, 0.0f);
    };
    double pq[3] = {0, 0, 0};
    for (int g = 0; g < MGB_G; ++g) {
        int x, y;
        if (!list_cell(L, blockIdx.x * MGB_G + g, nb, x, y))
            continue;
        const size_t ii = (size_t)y * L.w + x;
        const float dg = L.dg[ii];
        if (dg > 0) {
            const float4 p = pnew(ii), q = apply(L, pnew, x, y, ii, dg);
            Pn[ii] = p;
            Q[ii] = q;
            pq[0] += (double)p.x * q.x; pq[1] += (double)p.y * q.y; pq[2] += (double)p.z * q.z;
        }
    }
    block_sum3(pq[0], pq[1], pq[2], S.sc->pq[par]);
}

// alpha = rz_k / pq_k;  x += alpha p;  r -= alpha q;  rr[k & 1] += r.r.  Clears rz and pq of the other parity:
// their last readers (k_mgb_dirspmv of this iteration, k_mgb_update of the previous one) are done, their next
// writers (the coming cycle's level-0 prolongation, the coming k_mgb_dirspmv) have not started.
__global__ __launch_bounds__(256) void k_mgb_update(const VmMgbSys *__restrict__ sys, int k, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &L = S.lv[0];
    const int nb = L.nblocks[0], par = k & 1;
    if (blockIdx.x == 0) {
        slot_clear(S.sc->rz[par ^ 1]);
        slot_clear(S.sc->pq[par ^ 1]);
    }
    float al[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double pq = slot_sum(S.sc->pq[par], c);
        al[c] = pq > 0 ? (float)(slot_sum(S.sc->rz[par], c) / pq) : 0.0f;
    }
    const float4 *__restrict__ P = S.P[par], *__restrict__ Q = S.Q;
    float4 *X = S.X, *R = L.b;
    double rr[3] = {0, 0, 0};
    for (int g = 0; g < MGB_G; ++g) {
        int x, y;
        if (!list_cell(L, blockIdx.x * MGB_G + g, nb, x, y))
            continue;
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
    block_sum3(rr[0], rr[1], rr[2], S.sc->rr[par]);
}

// rz[k & 1] += r.z for hierarchies whose level 0 is solved inside the one-workgroup tail
__global__ __launch_bounds__(256) void k_mgb_dot_rz(const VmMgbSys *__restrict__ sys, int k, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &L = S.lv[0];
    const int nb = L.nblocks[0];
    double rz[3] = {0, 0, 0};
    for (int g = 0; g < MGB_G; ++g) {
        int x, y;
        if (!list_cell(L, blockIdx.x * MGB_G + g, nb, x, y))
            continue;
        const size_t ii = (size_t)y * L.w + x;
        if (L.dg[ii] > 0) {
            const float4 r = L.b[ii], z = L.x[ii];
            rz[0] += (double)r.x * z.x; rz[1] += (double)r.y * z.y; rz[2] += (double)r.z * z.z;
        }
    }
    block_sum3(rz[0], rz[1], rz[2], S.sc->rz[k & 1]);
}

// ---------------------------------------------------------------------------
// V(1,1) cycle

// C.b = P^T (F.b - A x),  x = omega F.b / dg recomputed at the five points of every fine cell
__global__ __launch_bounds__(256) void k_mgb_restrict(const VmMgbSys *__restrict__ sys, int l, float omega, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &F = S.lv[l], &C = S.lv[l + 1];
    int X, Y;
    if (!list_cell(C, blockIdx.x, C.nblocks[0], X, Y))
        return;
    const PreSmoothed xs{F.dg, F.b, omega};
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
            const float4 ax = apply(F, xs, x, y, ii, dg), bb = F.b[ii];
            acc.x += bb.x - ax.x;
            acc.y += bb.y - ax.y;
            acc.z += bb.z - ax.z;
        }
    C.b[(size_t)Y * C.w + X] = acc;
}

// F.x = x1 + omega (F.b - A x1) / dg,  x1 = x + P C.x (coarse correction + post-smoothing);
// DOT (level 0): rz[k & 1] += F.b . F.x  = r.z
template <bool DOT>
__global__ __launch_bounds__(256) void k_mgb_prolong(const VmMgbSys *__restrict__ sys, int l, float omega, int k,
                                                     uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &F = S.lv[l], &C = S.lv[l + 1];
    const int nb = F.nblocks[0];
    const Corrected x1{{F.dg, F.b, omega}, C.x, C.w};
    double rz[3] = {0, 0, 0};
    for (int g = 0; g < (DOT ? MGB_G : 1); ++g) {
        int x, y;
        if (!list_cell(F, DOT ? blockIdx.x * MGB_G + g : blockIdx.x, nb, x, y))
            continue;
        const size_t ii = (size_t)y * F.w + x;
        const float dg = F.dg[ii];
        float4 o = make_float4(0, 0, 0, 0);
        if (dg > 0) {
            const float4 c = x1(ii, x, y), s = apply(F, x1, x, y, ii, dg), b = F.b[ii];
            const float kk = omega / dg;
            o = make_float4(c.x + kk * (b.x - s.x), c.y + kk * (b.y - s.y), c.z + kk * (b.z - s.z), 0);
            if (DOT) {
                rz[0] += (double)b.x * o.x; rz[1] += (double)b.y * o.y; rz[2] += (double)b.z * o.z;
            }
        }
        F.x[ii] = o;
    }
    if (DOT)
        block_sum3(rz[0], rz[1], rz[2], S.sc->rz[k & 1]);
}

// the coarsest grid alone (hierarchies of one level: canvases of <= 1024 cells)
__global__ __launch_bounds__(1024) void k_mgb_coarsest(const VmMgbSys *__restrict__ sys, int l, float omega, int sweeps,
                                                       uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbLevel &L = sys[blockIdx.z].lv[l];
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

// The two coarsest grids of the cycle in ONE workgroup (F: at most 4096 cells, C: the coarsest, at most 1024):
// pre-smoothing of F (in LDS), residual restriction, the Jacobi sweeps on C, coarse correction + post-smoothing.
__global__ __launch_bounds__(1024) void k_mgb_tail(const VmMgbSys *__restrict__ sys, int l, float omega, int sweeps,
                                                   uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &F = S.lv[l], &C = S.lv[l + 1];
    __shared__ float4 xf[4096], xa[1024], xb[1024];
    const int t = threadIdx.x, nF = F.w * F.h, nC = C.w * C.h;
    const PreSmoothed xs{F.dg, F.b, omega};
    for (int i = t; i < nF; i += 1024)
        xf[i] = xs(i);
    __syncthreads();
    const FromArray XF{xf};
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
                const float4 ax = apply(F, XF, x, y, ii, fdg), fb = F.b[ii];
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
    // x1 = x + P xc (in place), then F.x = x1 + omega (F.b - A x1) / dg
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
            const float4 c = xf[i], s = apply(F, XF, x, y, (size_t)i, fdg), fb = F.b[i];
            const float kf = omega / fdg;
            o = make_float4(c.x + kf * (fb.x - s.x), c.y + kf * (fb.y - s.y), c.z + kf * (fb.z - s.z), 0);
        }
        F.x[i] = o;
    }
}

const dim3 blk2(64, 4);
inline int groups(int nb) { return (nb + MGB_G - 1) / MGB_G; }

} // namespace

void vm_mgb_launch_level0(const VmMgbSys *sys, int nsys, int gx, int gy, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_level0, dim3(gx, gy, nsys), blk2, 0, s, sys);
}

void vm_mgb_launch_coarsen(const VmMgbSys *sys, int nsys, int l, int gx, int gy, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_coarsen, dim3(gx, gy, nsys), blk2, 0, s, sys, l);
}

void vm_mgb_launch_compact(const VmMgbSys *sys, int nsys, int nlev_max, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_compact, dim3(nlev_max, 1, nsys), dim3(1024), 0, s, sys);
}

void vm_mgb_launch_init(const VmMgbSys *sys, int nsys, int nb0, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_init, dim3(groups(nb0), 1, nsys), blk2, 0, s, sys, active);
}

void vm_mgb_launch_restrict(const VmMgbSys *sys, int nsys, int l, int nb_coarse, float omega, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_restrict, dim3(nb_coarse, 1, nsys), blk2, 0, s, sys, l, omega, active);
}

void vm_mgb_launch_prolong(const VmMgbSys *sys, int nsys, int l, int nb_fine, float omega, int k, uint64_t active, hipStream_t s)
{
    if (l == 0)
        hipLaunchKernelGGL(k_mgb_prolong<true>, dim3(groups(nb_fine), 1, nsys), blk2, 0, s, sys, l, omega, k, active);
    else
        hipLaunchKernelGGL(k_mgb_prolong<false>, dim3(nb_fine, 1, nsys), blk2, 0, s, sys, l, omega, k, active);
}

void vm_mgb_launch_tail(const VmMgbSys *sys, int nsys, int l, float omega, int sweeps, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_tail, dim3(1, 1, nsys), dim3(1024), 0, s, sys, l, omega, sweeps, active);
}

void vm_mgb_launch_coarsest(const VmMgbSys *sys, int nsys, int l, float omega, int sweeps, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_coarsest, dim3(1, 1, nsys), dim3(1024), 0, s, sys, l, omega, sweeps, active);
}

void vm_mgb_launch_dot_rz(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_dot_rz, dim3(groups(nb0), 1, nsys), blk2, 0, s, sys, k, active);
}

void vm_mgb_launch_dirspmv(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_dirspmv, dim3(groups(nb0), 1, nsys), blk2, 0, s, sys, k, active);
}

void vm_mgb_launch_update(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_update, dim3(groups(nb0), 1, nsys), blk2, 0, s, sys, k, active);
}
