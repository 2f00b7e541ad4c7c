// vm_mgb.hip -- batched, ring-only, fused multigrid-preconditioned CG of the Poisson extension (gfx950).
// See vm_mgb.h for the design; vm_mg.hip holds the one-system, whole-canvas form it grew from (still the
// solver of the quadratic motion path, whose unknowns are the whole grid).  Every kernel: blockIdx.z = system,
// a workgroup = 256 threads = one 64 x 4-cell block of the level's compact block list (or MGB_G of them where the
// kernel ends in a dot product); HBM-bound streams over the ring of unknowns.
#include "vm_mgb.h"

namespace {

#ifndef MGB_G
#define MGB_G 4 // list entries per workgroup in the kernels that end in dot products (their atomics per byte / 4)
#endif

__device__ __forceinline__ float4 f4_axpy(float a, float4 x, float4 y) // a x + y
{
    return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), 0);
}
__device__ __forceinline__ float4 f4_sub(float4 s, float4 u) { return make_float4(s.x - u.x, s.y - u.y, s.z - u.z, 0); }
__device__ __forceinline__ float4 f4_sel(bool c, float4 a, float4 b) { return c ? a : b; }

// the solver's vectors carry three colour channels: 12 bytes per cell in memory (global_load / store_dwordx3), float4 in registers
__device__ __forceinline__ float4 ld3(const VmV3 *__restrict__ a, size_t q)
{
    const VmV3 v = a[q];
    return make_float4(v.x, v.y, v.z, 0);
}
__device__ __forceinline__ void st3(VmV3 *a, size_t q, float4 v) { a[q] = VmV3{v.x, v.y, v.z}; }

__device__ __forceinline__ bool sys_active(uint64_t active) { return (active >> blockIdx.z) & 1; }

// block `e` of the level's list: its origin (in cells)
__device__ __forceinline__ bool list_block(const VmMgbLevel &L, int e, int nb, int &x0, int &y0)
{
    if (e >= nb)
        return false;
    const uint32_t b = L.blocks[e];
    x0 = (int)(b & 0xffffu) * 64;
    y0 = (int)(b >> 16) * 4;
    return true;
}
// ... and the cell of it this thread owns, rows of 64 (the streaming kernels)
__device__ __forceinline__ bool list_cell(const VmMgbLevel &L, int e, int nb, int &x, int &y)
{
    int x0, y0;
    if (!list_block(L, e, nb, x0, y0))
        return false;
    x = x0 + (int)threadIdx.x;
    y = y0 + (int)threadIdx.y;
    return x < L.w && y < L.h;
}

// ---------------------------------------------------------------------------
// The operator of a level, by accessor.  (A u)(p) = dg u(p) - sum over the edges of p of w u(neighbour).

// omega / dg for the integer diagonals of level 0 (the float divisions the stored form would make, as constants)
__device__ __forceinline__ float k_of_dg0(uint32_t d)
{
    return d == 1 ? VM_MGB_OMEGA / 1.0f : d == 2 ? VM_MGB_OMEGA / 2.0f : d == 3 ? VM_MGB_OMEGA / 3.0f
         : d == 4 ? VM_MGB_OMEGA / 4.0f : d == 5 ? VM_MGB_OMEGA / 5.0f : 0.0f;
}

template <bool L0> struct Op;

// level 0: one byte per cell (vm_mgb.h), unit weights
template <> struct Op<true> {
    const uint8_t *__restrict__ info;
    int w, h;
    __device__ __forceinline__ explicit Op(const VmMgbLevel &L) : info(L.info), w(L.w), h(L.h) {}
    __device__ __forceinline__ float dg(size_t q) const { return (float)(info[q] >> 4); }
    __device__ __forceinline__ float k(size_t q) const { return k_of_dg0(info[q] >> 4); }
    // E, W, S, N: the order of vm_mg.hip's mg_apply; a unit weight's fma(-1, u, s) is s - u exactly.  All five values
    // are fetched whatever the edge bits say (index clamped to the canvas, the result dropped by a select): the
    // loads then do not wait for the info byte -- one round trip per cell instead of two
    template <class U>
    __device__ __forceinline__ float4 apply(const U &u, int x, int y, size_t ii, float d, uint32_t m) const
    {
        const bool ie = x + 1 < w, iw = x > 0, is = y + 1 < h, in = y > 0;
        const float4 c = u(ii, x, y);
        const float4 e = u(ie ? ii + 1 : ii, ie ? x + 1 : x, y), wv = u(iw ? ii - 1 : ii, iw ? x - 1 : x, y);
        const float4 sv = u(is ? ii + w : ii, x, is ? y + 1 : y), nv = u(in ? ii - w : ii, x, in ? y - 1 : y);
        float4 s = make_float4(d * c.x, d * c.y, d * c.z, 0);
        s = f4_sel((m & 1u) != 0, f4_sub(s, e), s);
        s = f4_sel((m & 2u) != 0, f4_sub(s, wv), s);
        s = f4_sel((m & 4u) != 0, f4_sub(s, sv), s);
        s = f4_sel((m & 8u) != 0, f4_sub(s, nv), s);
        return s;
    }
    template <class U>
    __device__ __forceinline__ float4 apply(const U &u, int x, int y, size_t ii, float d) const
    {
        return apply(u, x, y, ii, d, (uint32_t)info[ii]);
    }
    // weight of the edge towards east / south (for the coarsening)
    __device__ __forceinline__ float we(size_t q) const { return (float)(info[q] & 1u); }
    __device__ __forceinline__ float ws(size_t q) const { return (float)((info[q] >> 2) & 1u); }
};

// coarser levels: float weights, diagonal, k = omega / dg
template <> struct Op<false> {
    const float *__restrict__ pwe, *__restrict__ pws, *__restrict__ pdg, *__restrict__ pk;
    int w, h;
    __device__ __forceinline__ explicit Op(const VmMgbLevel &L) : pwe(L.we), pws(L.ws), pdg(L.dg), pk(L.k), w(L.w), h(L.h) {}
    __device__ __forceinline__ float dg(size_t q) const { return pdg[q]; }
    __device__ __forceinline__ float k(size_t q) const { return pk[q]; }
    // all four weights and all five values fetched at once (indices clamped to the grid, missing edges have weight 0
    // and drop out through a select -- never a multiplication: a cell nobody wrote may hold anything)
    template <class U>
    __device__ __forceinline__ float4 apply(const U &u, int x, int y, size_t ii, float d) const
    {
        const bool ie = x + 1 < w, iw = x > 0, is = y + 1 < h, in = y > 0;
        const float wE = ie ? pwe[ii] : 0.0f, wW = iw ? pwe[ii - 1] : 0.0f, wS = is ? pws[ii] : 0.0f, wN = in ? pws[ii - w] : 0.0f;
        const float4 c = u(ii, x, y);
        const float4 e = u(ie ? ii + 1 : ii, ie ? x + 1 : x, y), wv = u(iw ? ii - 1 : ii, iw ? x - 1 : x, y);
        const float4 sv = u(is ? ii + w : ii, x, is ? y + 1 : y), nv = u(in ? ii - w : ii, x, in ? y - 1 : y);
        float4 s = make_float4(d * c.x, d * c.y, d * c.z, 0);
        s = f4_sel(wE != 0.0f, f4_axpy(-wE, e, s), s);
        s = f4_sel(wW != 0.0f, f4_axpy(-wW, wv, s), s);
        s = f4_sel(wS != 0.0f, f4_axpy(-wS, sv, s), s);
        s = f4_sel(wN != 0.0f, f4_axpy(-wN, nv, s), s);
        return s;
    }
    __device__ __forceinline__ float we(size_t q) const { return pwe[q]; }
    __device__ __forceinline__ float ws(size_t q) const { return pws[q]; }
};

// the four edge weights of a cell in registers (the one-workgroup sweeps apply the operator 40 times)
struct Stencil {
    float wE, wW, wS, wN;
    template <class U>
    __device__ __forceinline__ float4 apply(const U *u, int t, int w, float d, float4 c) const
    {
        float4 s = make_float4(d * c.x, d * c.y, d * c.z, 0);
        if (wE != 0.0f) s = f4_axpy(-wE, u[t + 1], s);
        if (wW != 0.0f) s = f4_axpy(-wW, u[t - 1], s);
        if (wS != 0.0f) s = f4_axpy(-wS, u[t + w], s);
        if (wN != 0.0f) s = f4_axpy(-wN, u[t - w], s);
        return s;
    }
};
__device__ __forceinline__ Stencil stencil_of(const Op<true> &A, int x, int y, size_t t)
{
    const uint32_t m = A.info[t];
    return Stencil{(float)(m & 1u), (float)((m >> 1) & 1u), (float)((m >> 2) & 1u), (float)((m >> 3) & 1u)};
}
__device__ __forceinline__ Stencil stencil_of(const Op<false> &A, int x, int y, size_t t)
{
    return Stencil{x + 1 < A.w ? A.pwe[t] : 0.0f, x > 0 ? A.pwe[t - 1] : 0.0f, y + 1 < A.h ? A.pws[t] : 0.0f, y > 0 ? A.pws[t - A.w] : 0.0f};
}

struct FromArray {          // a vector in memory
    const VmV3 *__restrict__ a;
    __device__ __forceinline__ float4 operator()(size_t q, int = 0, int = 0) const { return ld3(a, q); }
};
struct FromLds {            // ... in LDS
    const float4 *a;
    __device__ __forceinline__ float4 operator()(size_t q, int = 0, int = 0) const { return a[q]; }
};

// the pre-smoothed iterate of a level, never stored: x = (omega / dg) b  (damped Jacobi from zero)
template <bool L0> struct PreSmoothed {
    Op<L0> op;
    const VmV3 *__restrict__ b;
    __device__ __forceinline__ float4 operator()(size_t q, int = 0, int = 0) const
    {
        // b is fetched beside k, not behind it (one round trip): level 0's is defined on the whole canvas (0 off the
        // ring); a coarse cell nobody restricts to holds a stale b, which the select drops
        const float k = op.k(q);
        const float4 v = ld3(b, q);
        return k == 0.0f ? make_float4(0, 0, 0, 0) : make_float4(k * v.x, k * v.y, k * v.z, 0);
    }
};

// ... plus the coarse correction: x1 = x + P xc
template <bool L0> struct Corrected {
    PreSmoothed<L0> pre;
    const VmV3 *__restrict__ xc;
    int cw;
    __device__ __forceinline__ float4 operator()(size_t q, int x, int y) const
    {
        const float4 f = pre(q), c = ld3(xc, (size_t)(y >> 1) * cw + (x >> 1));
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

// the three channel sums of two accumulators for the whole workgroup: six threads add the slots up (one round trip),
// everybody reads the results from LDS -- instead of 48 loads issued by every thread of every workgroup
__device__ __forceinline__ void slot_sums2(const double (*a)[16], const double (*b)[16], double *sa, double *sb)
{
    __shared__ double sh[6];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (tid < 6)
        sh[tid] = slot_sum(tid < 3 ? a : b, tid % 3);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        sa[c] = sh[c];
        sb[c] = sh[3 + c];
    }
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
    uint32_t dg = 0;
    if (x < L.w && y < L.h) {
        const size_t ii = (size_t)y * L.w + x;
        const uint8_t t = type[ii];
        uint32_t m = 0;
        if (t > 0) {
            dg = t == 1 ? 1u : 0u;
            if (x + 1 < L.w && type[ii + 1] > 0) { m |= 1u; ++dg; }
            if (x > 0 && type[ii - 1] > 0) { m |= 2u; ++dg; }
            if (y + 1 < L.h && type[ii + L.w] > 0) { m |= 4u; ++dg; }
            if (y > 0 && type[ii - L.w] > 0) { m |= 8u; ++dg; }
        }
        L.info[ii] = (uint8_t)(dg << 4 | m);
    }
    const int any = __syncthreads_or(dg > 0);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        L.flags[blockIdx.y * L.gx + blockIdx.x] = any ? 1u : 0u;
}

// Galerkin coarse operator of level l from level l - 1 (2x2 aggregates, piecewise-constant interpolation, the
// edge weights rescaled by 1/2), the diagonal in the same pass: the west / north weights of a coarse cell are
// those of fine edges that enter its block
template <bool L0>
__global__ __launch_bounds__(256) void k_mgb_coarsen(const VmMgbSys *__restrict__ sys, int l)
{
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &FL = S.lv[l - 1], &C = S.lv[l];
    const Op<L0> F(FL);
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
                const float e = F.we(ii), s = F.ws(ii);
                float inc = e + s;
                const float fw = x > 0 ? F.we(ii - 1) : 0.0f, fn = y > 0 ? F.ws(ii - F.w) : 0.0f;
                if (x > 0) inc += fw;
                if (y > 0) inc += fn;
                sc += F.dg(ii) - inc; // screening = diagonal - incident weights
                if (a == 1) we += e;  // edges leaving the block to the east / south
                if (b == 1) ws += s;
                if (a == 0) ww += fw; // ... and entering it from the west / north
                if (b == 0) wn += fn;
            }
        const size_t k = (size_t)Y * C.w + X;
        we *= 0.5f; ws *= 0.5f; ww *= 0.5f; wn *= 0.5f;
        C.we[k] = we;
        C.ws[k] = ws;
        d = fmaxf(sc, 0.0f) + we + ws;
        if (X > 0) d += ww;
        if (Y > 0) d += wn;
        C.dg[k] = d;
        C.k[k] = d > 0 ? VM_MGB_OMEGA / d : 0.0f;
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
    const Op<true> A(L);
    const int nb = L.nblocks[0];
    const FromArray X{S.X};
    double bb[3] = {0, 0, 0}, rr[3] = {0, 0, 0};
#pragma unroll
    for (int g = 0; g < MGB_G; ++g) {
        int x, y;
        if (!list_cell(L, blockIdx.x * MGB_G + g, nb, x, y))
            continue;
        const size_t ii = (size_t)y * L.w + x;
        const uint32_t m = L.info[ii];
        const float dg = (float)(m >> 4);
        const float4 b = ld3(L.b, ii), ax = A.apply(X, x, y, ii, dg, m);
        if (dg > 0) {
            const float4 r = make_float4(b.x - ax.x, b.y - ax.y, b.z - ax.z, 0);
            bb[0] += (double)b.x * b.x; bb[1] += (double)b.y * b.y; bb[2] += (double)b.z * b.z;
            rr[0] += (double)r.x * r.x; rr[1] += (double)r.y * r.y; rr[2] += (double)r.z * r.z;
            st3(L.b, ii, r);
        }
    }
    block_sum3(bb[0], bb[1], bb[2], S.sc->bb);
    block_sum3(rr[0], rr[1], rr[2], S.sc->rr[1]);
}

// beta = rz_k / rz_{k-1} (0 in the first iteration);  p = z + beta p_old at the cell and its neighbours, q = A p;
// pq[k & 1] += p.q.  Clears rr[k & 1], which k_mgb_update accumulates next.  p and q are written for every cell of
// an active block (zeros off the ring), so that nothing downstream of the loads hangs on the info byte.
template <bool FIRST>
__global__ __launch_bounds__(256) void k_mgb_dirspmv(const VmMgbSys *__restrict__ sys, int k, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &L = S.lv[0];
    const Op<true> A(L);
    const int nb = L.nblocks[0], par = k & 1;
    if (blockIdx.x == 0)
        slot_clear(S.sc->rr[par]);
    float be[3] = {0, 0, 0};
    if (!FIRST) {
        double cur[3], prev[3];
        slot_sums2(S.sc->rz[par], S.sc->rz[par ^ 1], cur, prev);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            be[c] = prev[c] > 0 ? (float)(cur[c] / prev[c]) : 0.0f;
    }
    const VmV3 *__restrict__ Z = L.x;
    const VmV3 *__restrict__ Po = S.P[par ^ 1];
    VmV3 *Pn = S.P[par], *Q = S.Q;
    auto pnew = [&](size_t q, int = 0, int = 0) {
        const float4 z = ld3(Z, q);
        if (FIRST)
            return z;
        const float4 po = ld3(Po, q);
        return make_float4(z.x + be[0] * po.x, z.y + be[1] * po.y, z.z + be[2] * po.z, 0.0f);
    };
    double pq[3] = {0, 0, 0};
#pragma unroll
    for (int g = 0; g < MGB_G; ++g) {
        int x, y;
        if (!list_cell(L, blockIdx.x * MGB_G + g, nb, x, y))
            continue;
        const size_t ii = (size_t)y * L.w + x;
        const uint32_t m = L.info[ii];
        const float dg = (float)(m >> 4);
        const bool unk = (m >> 4) != 0;
        const float4 zero = make_float4(0, 0, 0, 0);
        const float4 p = f4_sel(unk, pnew(ii), zero), q = f4_sel(unk, A.apply(pnew, x, y, ii, dg, m), zero);
        st3(Pn, ii, p);
        st3(Q, ii, q);
        pq[0] += (double)p.x * q.x; pq[1] += (double)p.y * q.y; pq[2] += (double)p.z * q.z;
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
    {
        double rz[3], pq[3];
        slot_sums2(S.sc->rz[par], S.sc->pq[par], rz, pq);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            al[c] = pq[c] > 0 ? (float)(rz[c] / pq[c]) : 0.0f;
    }
    const VmV3 *__restrict__ P = S.P[par], *__restrict__ Q = S.Q;
    VmV3 *X = S.X, *R = L.b;
    double rr[3] = {0, 0, 0};
#pragma unroll
    for (int g = 0; g < MGB_G; ++g) {
        int x, y;
        if (!list_cell(L, blockIdx.x * MGB_G + g, nb, x, y))
            continue;
        const size_t ii = (size_t)y * L.w + x;
        // p and q are zero off the ring (k_mgb_dirspmv), x and r stay what they are there: no test of the info byte
        const float4 p = ld3(P, ii), q = ld3(Q, ii);
        float4 xx = ld3(X, ii), r = ld3(R, ii);
        xx.x += al[0] * p.x; xx.y += al[1] * p.y; xx.z += al[2] * p.z;
        r.x -= al[0] * q.x; r.y -= al[1] * q.y; r.z -= al[2] * q.z;
        st3(X, ii, xx);
        st3(R, ii, r);
        if (L.info[ii] >> 4) {
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
        if (L.info[ii] >> 4) {
            const float4 r = ld3(L.b, ii), z = ld3(L.x, ii);
            rz[0] += (double)r.x * z.x; rz[1] += (double)r.y * z.y; rz[2] += (double)r.z * z.z;
        }
    }
    block_sum3(rz[0], rz[1], rz[2], S.sc->rz[k & 1]);
}

// ---------------------------------------------------------------------------
// V(1,1) cycle

// C.b = P^T (F.b - A x),  x = (omega / dg) F.b recomputed at the five points of every fine cell.  One thread per
// FINE cell of a block of F's list: a wave covers 32 x 2 fine cells (lane = x & 31 | (y & 1) << 5), so the 2 x 2
// aggregate is two lane exchanges, and a workgroup (4 waves) exactly one 64 x 4 fine block = 32 x 2 coarse cells.
// (A coarse cell no fine block reaches holds no unknown: its right-hand side is never read.)
template <bool L0>
__global__ __launch_bounds__(256) void k_mgb_restrict(const VmMgbSys *__restrict__ sys, int l, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &FL = S.lv[l], &C = S.lv[l + 1];
    int x0, y0;
    if (!list_block(FL, blockIdx.x, FL.nblocks[0], x0, y0))
        return;
    const Op<L0> F(FL);
    const PreSmoothed<L0> xs{F, FL.b};
    const int tid = threadIdx.y * 64 + threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int x = x0 + (wave & 1) * 32 + (lane & 31), y = y0 + (wave >> 1) * 2 + (lane >> 5);
    float4 r = make_float4(0, 0, 0, 0);
    if (x < F.w && y < F.h) {
        const size_t ii = (size_t)y * F.w + x;
        const float dg = F.dg(ii);
        // every load issued before the diagonal is known (apply fetches unconditionally; the select drops what a
        // cell that is no unknown computed from stale memory)
        const float4 ax = F.apply(xs, x, y, ii, dg), bb = ld3(FL.b, ii);
        r = f4_sel(dg > 0, make_float4(bb.x - ax.x, bb.y - ax.y, bb.z - ax.z, 0), r);
    }
    r.x += __shfl_xor(r.x, 1); r.y += __shfl_xor(r.y, 1); r.z += __shfl_xor(r.z, 1);
    r.x += __shfl_xor(r.x, 32); r.y += __shfl_xor(r.y, 32); r.z += __shfl_xor(r.z, 32);
    if ((lane & 33) == 0 && x < F.w && y < F.h)
        st3(C.b, (size_t)(y >> 1) * C.w + (x >> 1), r);
}

// F.x = x1 + (omega / dg) (F.b - A x1),  x1 = x + P C.x (coarse correction + post-smoothing);
// level 0 (L0): rz[k & 1] += F.b . F.x  = r.z
template <bool L0>
__global__ __launch_bounds__(256) void k_mgb_prolong(const VmMgbSys *__restrict__ sys, int l, int k, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &FL = S.lv[l], &C = S.lv[l + 1];
    const Op<L0> F(FL);
    const int nb = FL.nblocks[0];
    const Corrected<L0> x1{{F, FL.b}, C.x, C.w};
    double rz[3] = {0, 0, 0};
#pragma unroll
    for (int g = 0; g < (L0 ? MGB_G : 1); ++g) {
        int x, y;
        if (!list_cell(FL, L0 ? blockIdx.x * MGB_G + g : blockIdx.x, nb, x, y))
            continue;
        const size_t ii = (size_t)y * F.w + x;
        const float dg = F.dg(ii);
        float4 o = make_float4(0, 0, 0, 0);
        {
            const float4 c = x1(ii, x, y), s = F.apply(x1, x, y, ii, dg), b = ld3(FL.b, ii);
            const float kk = F.k(ii);
            o = f4_sel(dg > 0, make_float4(c.x + kk * (b.x - s.x), c.y + kk * (b.y - s.y), c.z + kk * (b.z - s.z), 0), o);
            if (L0 && dg > 0) {
                rz[0] += (double)b.x * o.x; rz[1] += (double)b.y * o.y; rz[2] += (double)b.z * o.z;
            }
        }
        st3(FL.x, ii, o);
    }
    if (L0)
        block_sum3(rz[0], rz[1], rz[2], S.sc->rz[k & 1]);
}

// the coarsest grid alone (hierarchies of one level: canvases of <= 1024 cells)
template <bool L0>
__global__ __launch_bounds__(1024) void k_mgb_coarsest(const VmMgbSys *__restrict__ sys, int l, int sweeps, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbLevel &L = sys[blockIdx.z].lv[l];
    const Op<L0> A(L);
    __shared__ float4 xa[1024], xb[1024];
    const int t = threadIdx.x, n = L.w * L.h;
    const int x = t % L.w, y = t / L.w;
    float dg = 0, k = 0;
    float4 b = make_float4(0, 0, 0, 0);
    Stencil st{0, 0, 0, 0};
    if (t < n) {
        dg = A.dg(t);
        k = A.k(t);
        if (dg > 0) b = ld3(L.b, t);
        st = stencil_of(A, x, y, (size_t)t);
    }
    float4 cur = make_float4(k * b.x, k * b.y, k * b.z, 0);
    float4 *src = xa, *dst = xb;
    src[t] = cur;
    __syncthreads();
    for (int it = 1; it < sweeps; ++it) {
        if (t < n && dg > 0) {
            const float4 s = st.apply(src, t, L.w, dg, cur);
            cur = make_float4(cur.x + k * (b.x - s.x), cur.y + k * (b.y - s.y), cur.z + k * (b.z - s.z), 0);
        }
        dst[t] = cur;
        __syncthreads();
        float4 *tmp = src;
        src = dst;
        dst = tmp;
    }
    if (t < n)
        st3(L.x, t, cur);
}

// The two coarsest grids of the cycle in ONE workgroup (F: at most 4096 cells, C: the coarsest, at most 1024):
// pre-smoothing of F (in LDS), residual restriction, the Jacobi sweeps on C, coarse correction + post-smoothing.
template <bool L0>
__global__ __launch_bounds__(1024) void k_mgb_tail(const VmMgbSys *__restrict__ sys, int l, int sweeps, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &FL = S.lv[l], &CL = S.lv[l + 1];
    const Op<L0> F(FL);
    const Op<false> C(CL);
    __shared__ float4 xf[4096], xa[1024], xb[1024];
    const int t = threadIdx.x, nF = F.w * F.h, nC = C.w * C.h;
    const PreSmoothed<L0> xs{F, FL.b};
    for (int i = t; i < nF; i += 1024)
        xf[i] = xs(i);
    __syncthreads();
    const FromLds XF{xf};
    const int X = t % C.w, Y = t / C.w;
    float dg = 0, k = 0;
    float4 b = make_float4(0, 0, 0, 0);
    Stencil st{0, 0, 0, 0};
    if (t < nC) {
        for (int bb = 0; bb < 2; ++bb)
            for (int a = 0; a < 2; ++a) {
                const int x = 2 * X + a, y = 2 * Y + bb;
                if (x >= F.w || y >= F.h)
                    continue;
                const size_t ii = (size_t)y * F.w + x;
                const float fdg = F.dg(ii);
                if (!(fdg > 0))
                    continue;
                const float4 ax = F.apply(XF, x, y, ii, fdg), fb = ld3(FL.b, ii);
                b.x += fb.x - ax.x;
                b.y += fb.y - ax.y;
                b.z += fb.z - ax.z;
            }
        dg = C.dg(t);
        k = C.k(t);
        st = stencil_of(C, X, Y, (size_t)t);
    }
    float4 cur = make_float4(k * b.x, k * b.y, k * b.z, 0);
    float4 *src = xa, *dst = xb;
    src[t] = cur;
    __syncthreads();
    for (int it = 1; it < sweeps; ++it) {
        if (t < nC && dg > 0) {
            const float4 s = st.apply(src, t, C.w, dg, cur);
            cur = make_float4(cur.x + k * (b.x - s.x), cur.y + k * (b.y - s.y), cur.z + k * (b.z - s.z), 0);
        }
        dst[t] = cur;
        __syncthreads();
        float4 *tmp = src;
        src = dst;
        dst = tmp;
    }
    // x1 = x + P xc (in place), then F.x = x1 + (omega / dg) (F.b - A x1)
    for (int i = t; i < nF; i += 1024) {
        const int x = i % F.w, y = i / F.w;
        const float4 c = src[(y >> 1) * C.w + (x >> 1)], f = xf[i];
        xf[i] = make_float4(f.x + c.x, f.y + c.y, f.z + c.z, 0);
    }
    __syncthreads();
    for (int i = t; i < nF; i += 1024) {
        const int x = i % F.w, y = i / F.w;
        const float fdg = F.dg(i);
        float4 o = make_float4(0, 0, 0, 0);
        if (fdg > 0) {
            const float4 c = xf[i], s = F.apply(XF, x, y, (size_t)i, fdg), fb = ld3(FL.b, i);
            const float kf = F.k(i);
            o = make_float4(c.x + kf * (fb.x - s.x), c.y + kf * (fb.y - s.y), c.z + kf * (fb.z - s.z), 0);
        }
        st3(FL.x, i, o);
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
    if (l == 1)
        hipLaunchKernelGGL(k_mgb_coarsen<true>, dim3(gx, gy, nsys), blk2, 0, s, sys, l);
    else
        hipLaunchKernelGGL(k_mgb_coarsen<false>, dim3(gx, gy, nsys), blk2, 0, s, sys, l);
}

void vm_mgb_launch_compact(const VmMgbSys *sys, int nsys, int nlev_max, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_compact, dim3(nlev_max, 1, nsys), dim3(1024), 0, s, sys);
}

void vm_mgb_launch_init(const VmMgbSys *sys, int nsys, int nb0, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_init, dim3(groups(nb0), 1, nsys), blk2, 0, s, sys, active);
}

void vm_mgb_launch_restrict(const VmMgbSys *sys, int nsys, int l, int nb_fine, uint64_t active, hipStream_t s)
{
    if (l == 0)
        hipLaunchKernelGGL(k_mgb_restrict<true>, dim3(nb_fine, 1, nsys), blk2, 0, s, sys, l, active);
    else
        hipLaunchKernelGGL(k_mgb_restrict<false>, dim3(nb_fine, 1, nsys), blk2, 0, s, sys, l, active);
}

void vm_mgb_launch_prolong(const VmMgbSys *sys, int nsys, int l, int nb_fine, int k, uint64_t active, hipStream_t s)
{
    if (l == 0)
        hipLaunchKernelGGL(k_mgb_prolong<true>, dim3(groups(nb_fine), 1, nsys), blk2, 0, s, sys, l, k, active);
    else
        hipLaunchKernelGGL(k_mgb_prolong<false>, dim3(nb_fine, 1, nsys), blk2, 0, s, sys, l, k, active);
}

void vm_mgb_launch_tail(const VmMgbSys *sys, int nsys, int l, int sweeps, uint64_t active, hipStream_t s)
{
    if (l == 0)
        hipLaunchKernelGGL(k_mgb_tail<true>, dim3(1, 1, nsys), dim3(1024), 0, s, sys, l, sweeps, active);
    else
        hipLaunchKernelGGL(k_mgb_tail<false>, dim3(1, 1, nsys), dim3(1024), 0, s, sys, l, sweeps, active);
}

void vm_mgb_launch_coarsest(const VmMgbSys *sys, int nsys, int l, int sweeps, uint64_t active, hipStream_t s)
{
    if (l == 0)
        hipLaunchKernelGGL(k_mgb_coarsest<true>, dim3(1, 1, nsys), dim3(1024), 0, s, sys, l, sweeps, active);
    else
        hipLaunchKernelGGL(k_mgb_coarsest<false>, dim3(1, 1, nsys), dim3(1024), 0, s, sys, l, sweeps, active);
}

void vm_mgb_launch_dot_rz(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_dot_rz, dim3(groups(nb0), 1, nsys), blk2, 0, s, sys, k, active);
}

void vm_mgb_launch_dirspmv(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s)
{
    if (k == 0)
        hipLaunchKernelGGL(k_mgb_dirspmv<true>, dim3(groups(nb0), 1, nsys), blk2, 0, s, sys, k, active);
    else
        hipLaunchKernelGGL(k_mgb_dirspmv<false>, dim3(groups(nb0), 1, nsys), blk2, 0, s, sys, k, active);
}

void vm_mgb_launch_update(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s)
{
    hipLaunchKernelGGL(k_mgb_update, dim3(groups(nb0), 1, nsys), blk2, 0, s, sys, k, active);
}
