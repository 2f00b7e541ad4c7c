// vm_mgb.hip -- batched, ring-only, fused multigrid-preconditioned CG (gfx950): the solver of the Poisson extension
// (PoissonExt.cpp:214-329) and of the quadratic motion path (QuadraticPath.cpp:111-305), whose unknowns are the whole
// grid.  See vm_mgb.h for the design.  Every kernel: blockIdx.z = system; the PCG kernels: a workgroup = 256 threads =
// MGB_G 64 x 4-cell blocks of level 0's compact block list, HBM-bound streams over the ring of unknowns; the cycle's
// kernels: a workgroup = one 64 x 16-cell tile of the level's tile list, staged in LDS with a two-cell apron.
#include "vm_mgb.h"

namespace {

#ifndef MGB_G
#define MGB_G 4 // list entries per workgroup in the kernels that end in dot products (their atomics per byte / 4)
#endif

// Every array pointer of this file comes out of a descriptor in memory (VmMgbSys), which makes it a GENERIC pointer to the
// compiler: flat_load / flat_store, which tick the LDS counter as well as the memory counter -- an s_waitcnt for an LDS
// read or a scalar load then also waits for every store in flight.  The arrays are hipMalloc'ed: say so.
#define VM_G __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ VM_G T *G(T *p) { return (VM_G T *)p; }

__device__ __forceinline__ float4 f4_axpy(float a, float4 x, float4 y) // a x + y
{
    return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), 0);
}
__device__ __forceinline__ float4 f4_sub(float4 s, float4 u) { return make_float4(s.x - u.x, s.y - u.y, s.z - u.z, 0); }
__device__ __forceinline__ float4 f4_sel(bool c, float4 a, float4 b) { return c ? a : b; }

// the solver's vectors carry three colour channels: 12 bytes per cell in memory (global_load / store_dwordx3), float4 in registers
// (a 12-byte copy, not three float accesses: those would be merged into a dwordx2 and a dword)
__device__ __forceinline__ float4 ld3(const VmV3 *__restrict__ a, size_t q)
{
    VmV3 v;
    __builtin_memcpy(&v, G(a) + q, sizeof(VmV3));
    return make_float4(v.x, v.y, v.z, 0);
}
__device__ __forceinline__ void st3(VmV3 *a, size_t q, float4 v)
{
    const VmV3 t{v.x, v.y, v.z};
    __builtin_memcpy(G(a) + q, &t, sizeof(VmV3));
}

__device__ __forceinline__ bool sys_active(uint64_t active) { return (active >> blockIdx.z) & 1; }

// block `e` of the level's list: its origin (in cells)
__device__ __forceinline__ bool list_block(const VmMgbLevel &L, int e, int nb, int &x0, int &y0)
{
    if (e >= nb)
        return false;
    const uint32_t b = G(L.blocks)[e];
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

// 1 / dg for the integer diagonals of level 0 (the float divisions the stored form would make, as constants)
__device__ __forceinline__ float k_of_dg0(uint32_t d)
{
    return d == 1 ? 1.0f : d == 2 ? 0.5f : d == 3 ? 1.0f / 3.0f : d == 4 ? 0.25f : d == 5 ? 0.2f : 0.0f;
}

template <bool L0> struct Op;

// level 0: one byte per cell (vm_mgb.h), unit weights
template <> struct Op<true> {
    const VM_G uint8_t *__restrict__ info;
    int w, h;
    __device__ __forceinline__ explicit Op(const VmMgbLevel &L) : info(G(L.info)), w(L.w), h(L.h) {}
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

// coarser levels: float weights, diagonal, k = 1 / dg
template <> struct Op<false> {
    const VM_G float *__restrict__ pwe, *__restrict__ pws, *__restrict__ pdg, *__restrict__ pk;
    int w, h;
    __device__ __forceinline__ explicit Op(const VmMgbLevel &L) : pwe(G(L.we)), pws(G(L.ws)), pdg(G(L.dg)), pk(G(L.k)), w(L.w), h(L.h) {}
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
    const VM_G uint8_t *__restrict__ type = G(S.type);
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
        G(L.info)[ii] = (uint8_t)(dg << 4 | m);
    }
    const int any = __syncthreads_or(dg > 0);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        G(L.flags)[blockIdx.y * L.gx + blockIdx.x] = any ? 1u : 0u;
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
        G(C.we)[k] = we;
        G(C.ws)[k] = ws;
        d = fmaxf(sc, 0.0f) + we + ws;
        if (X > 0) d += ww;
        if (Y > 0) d += wn;
        G(C.dg)[k] = d;
        G(C.k)[k] = d > 0 ? 1.0f / d : 0.0f;
    }
    const int any = __syncthreads_or(d > 0);
    if (threadIdx.x == 0 && threadIdx.y == 0)
        G(C.flags)[blockIdx.y * C.gx + blockIdx.x] = any ? 1u : 0u;
}

// block flags -> compact row-major lists of the blocks and of the 64 x 16-cell tiles (four blocks of a column) that
// hold an unknown, one workgroup per (level, system)
__global__ __launch_bounds__(1024) void k_mgb_compact(const VmMgbSys *__restrict__ sys)
{
    const VmMgbSys &S = sys[blockIdx.z];
    const int l = blockIdx.x;
    if (l >= S.nlev)
        return;
    const VmMgbLevel &L = S.lv[l];
    __shared__ int wcount[16];
    __shared__ int base;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int pass = 0; pass < 2; ++pass) {               // 0: blocks, 1: tiles
        const int rows = pass ? (L.gy + 3) / 4 : L.gy, n = L.gx * rows;
        uint32_t *list = pass ? L.tiles : L.blocks;
        if (t == 0) base = 0;
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += 1024) {
            const int i = c0 + t, bx = i % L.gx, r = i / L.gx;
            bool f = false;
            if (i < n) {
                if (pass) {
                    for (int q = 4 * r; q < 4 * r + 4 && q < L.gy; ++q)
                        f = f || L.flags[q * L.gx + bx] != 0;
                } else {
                    f = L.flags[i] != 0;
                }
            }
            const unsigned long long m = __ballot(f);
            if (lane == 0) wcount[wave] = __popcll(m);
            __syncthreads();
            int off = base;
            for (int k = 0; k < wave; ++k) off += wcount[k];
            if (f)
                list[off + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)bx | ((uint32_t)r << 16);
            __syncthreads();
            if (t == 0) {
                int tot = 0;
                for (int k = 0; k < 16; ++k) tot += wcount[k];
                base += tot;
            }
            __syncthreads();
        }
        if (t == 0) (pass ? L.ntiles : L.nblocks)[0] = base;
        __syncthreads();
    }
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
    const int nb = G(L.nblocks)[0];
    const FromArray X{S.X};
    double bb[3] = {0, 0, 0}, rr[3] = {0, 0, 0};
#pragma unroll
    for (int g = 0; g < MGB_G; ++g) {
        int x, y;
        if (!list_cell(L, blockIdx.x * MGB_G + g, nb, x, y))
            continue;
        const size_t ii = (size_t)y * L.w + x;
        const uint32_t m = G(L.info)[ii];
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
    const int nb = G(L.nblocks)[0], par = k & 1;
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
        const uint32_t m = G(L.info)[ii];
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
    const int nb = G(L.nblocks)[0], par = k & 1;
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
        if (G(L.info)[ii] >> 4) {
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
    const int nb = G(L.nblocks)[0];
    double rz[3] = {0, 0, 0};
    for (int g = 0; g < MGB_G; ++g) {
        int x, y;
        if (!list_cell(L, blockIdx.x * MGB_G + g, nb, x, y))
            continue;
        const size_t ii = (size_t)y * L.w + x;
        if (G(L.info)[ii] >> 4) {
            const float4 r = ld3(L.b, ii), z = ld3(L.x, ii);
            rz[0] += (double)r.x * z.x; rz[1] += (double)r.y * z.y; rz[2] += (double)r.z * z.z;
        }
    }
    block_sum3(rz[0], rz[1], rz[2], S.sc->rz[k & 1]);
}

// ---------------------------------------------------------------------------
// The cycle with ONE symmetric red-black Gauss-Seidel sweep each way (VmMgbLevel::nu == 1).  Colour of a cell: red <=> (x + y) even.
//
// Pre-smoothing from a zero guess, red then black:      x_r = b_r / dg_r,   x_b = (b_b + sum_nb w x_r) / dg_b.
// Its residual is zero on black cells (their equations were just solved) and sum_nb w x_b on red cells (dg_r x_r = b_r),
// so the restriction to a 2 x 2 aggregate is the sum over its two red cells of four weighted black neighbours.
// Post-smoothing of x1 = x + P xc, black then red:      x_b' = (b_b + sum w x1_r) / dg_b,   x_r' = (b_r + sum w x_b') / dg_r
// needs x1 on RED cells only, which is b_r / dg_r + xc(parent): pointwise.  Both kernels therefore stage ONE LDS tile
// of the right-hand side with a two-cell apron (64 x 16 cells + apron = 68 x 20), run two half-sweeps inside it and
// never store the pre-smoothed iterate.  A workgroup = 256 threads = one tile of the level's tile list; thread
// (tx, ty) owns the cells (tx, 4 ty .. 4 ty + 3) of the tile: two red, two black.

constexpr int TW = 64, TH = 16, HL = 2, LW = TW + 2 * HL, LH = TH + 2 * HL, LN = LW * LH;
constexpr int NHALO = LN - TW * TH;                 // apron cells of a tile: 336
constexpr int NRING1 = (TW + 2) * (TH + 2) - TW * TH; // the cells at distance 1 around the tile: 164

__device__ __forceinline__ float4 f4_scale(float k, float4 v) { return make_float4(k * v.x, k * v.y, k * v.z, 0); }

// the operator records of a staged tile, by cell of the LDS window (WW cells wide, WN cells)
template <bool L0, int WW, int WN> struct TileOpT;
template <int WW, int WN> struct TileOpT<true, WW, WN> {       // level 0: the info byte (unit weights)
    uint8_t m[WN];
    __device__ __forceinline__ float4 nbsum(const float4 *v, int c) const
    {
        const uint32_t b = m[c];
        const float4 e = v[c + 1], w = v[c - 1], s = v[c + WW], n = v[c - WW];
        float4 a = make_float4(0, 0, 0, 0);
        if (b & 1u) { a.x += e.x; a.y += e.y; a.z += e.z; }
        if (b & 2u) { a.x += w.x; a.y += w.y; a.z += w.z; }
        if (b & 4u) { a.x += s.x; a.y += s.y; a.z += s.z; }
        if (b & 8u) { a.x += n.x; a.y += n.y; a.z += n.z; }
        return a;
    }
};
template <int WW, int WN> struct TileOpT<false, WW, WN> {      // coarser levels: weights of the edges to the east / south neighbour
    float2 w[WN];
    __device__ __forceinline__ float4 nbsum(const float4 *v, int c) const
    {
        const float wE = w[c].x, wW = w[c - 1].x, wS = w[c].y, wN = w[c - WW].y;
        const float4 e = v[c + 1], ww = v[c - 1], s = v[c + WW], n = v[c - WW];
        // a missing edge has weight 0 and its neighbour's LDS value is 0 or finite (stage_cell): plain multiply-adds
        float4 a = make_float4(wE * e.x, wE * e.y, wE * e.z, 0);
        a = f4_axpy(wW, ww, a);
        a = f4_axpy(wS, s, a);
        a = f4_axpy(wN, n, a);
        return a;
    }
};
template <bool L0> using TileOp = TileOpT<L0, LW, LN>;

// Stage cell c of the window (grid cell (x, y)): the operator record, 1 / dg in .w, and the value the first half-sweep
// leaves there -- red: b / dg (+ the coarse correction of its aggregate when xc is given), black: b.  A cell outside
// the grid or without an unknown gets zeros (selected, never multiplied: memory nobody wrote may hold anything).
// Returns b (zeros likewise).
template <bool L0>
__device__ __forceinline__ float4 stage_cell(const VmMgbLevel &L, const VmV3 *__restrict__ bsrc, const Op<L0> &A, TileOp<L0> &op, float4 *vals, int c,
                                             int x, int y, const VmV3 *__restrict__ xc, int cw)
{
    const bool in = x >= 0 && x < L.w && y >= 0 && y < L.h;
    const size_t ii = in ? (size_t)y * L.w + x : 0;
    float inv;
    if constexpr (L0) {
        const uint32_t m = in ? (uint32_t)A.info[ii] : 0u;
        op.m[c] = (uint8_t)m;
        inv = k_of_dg0(m >> 4);
    } else {
        const float k = A.pk[ii], we = A.pwe[ii], ws = A.pws[ii];
        inv = in ? k : 0.0f;
        op.w[c] = make_float2(in && x + 1 < L.w ? we : 0.0f, in && y + 1 < L.h ? ws : 0.0f);
    }
    const float4 braw = ld3(bsrc, ii);
    float4 cor = make_float4(0, 0, 0, 0);
    if (xc)
        cor = ld3(xc, in ? (size_t)(y >> 1) * cw + (x >> 1) : 0);
    const bool unk = inv > 0.0f;
    const float4 b = f4_sel(unk, braw, make_float4(0, 0, 0, 0));
    const bool red = ((x + y) & 1) == 0;
    float4 v = b;
    if (red)
        v = f4_sel(unk, make_float4(inv * b.x + cor.x, inv * b.y + cor.y, inv * b.z + cor.z, 0), make_float4(0, 0, 0, 0));
    v.w = inv;
    vals[c] = v;
    return b;
}

// window index of apron cell number k (0 .. NHALO - 1): two rows above, two below, two columns left and right
__device__ __forceinline__ int halo_cell(int k)
{
    if (k < 2 * LW) return k;                                   // rows 0, 1
    k -= 2 * LW;
    if (k < 2 * LW) return (LH - 2) * LW + k;                   // rows LH - 2, LH - 1
    k -= 2 * LW;
    const int row = HL + (k >> 2), q = k & 3;                   // 4 cells per interior row: columns 0, 1, LW - 2, LW - 1
    return row * LW + (q < 2 ? q : LW - 4 + q);
}
// ... and of cell number k (0 .. NRING1 - 1) of the ring at distance 1 around the tile
__device__ __forceinline__ int ring1_cell(int k)
{
    if (k < TW + 2) return (HL - 1) * LW + (HL - 1) + k;        // the row above
    k -= TW + 2;
    if (k < TW + 2) return (HL + TH) * LW + (HL - 1) + k;       // the row below
    k -= TW + 2;
    return (HL + (k >> 1)) * LW + ((k & 1) ? HL + TW : HL - 1); // left / right column
}

// black half-sweep on window cell c (its .w = 1 / dg): x = (b + sum_nb w x_red) / dg; returns what stood there (b)
template <bool L0>
__device__ __forceinline__ float4 black_update(const TileOp<L0> &op, float4 *vals, int c, float4 &xnew)
{
    const float4 me = vals[c];
    const float4 s = op.nbsum(vals, c);
    xnew = f4_sel(me.w > 0.0f, make_float4(me.w * (me.x + s.x), me.w * (me.y + s.y), me.w * (me.z + s.z), me.w), make_float4(0, 0, 0, 0));
    return me;
}

// C.b = P^T (F.b - A x),  x = red-black pre-smoothing of F.b from zero, over F's tile list.  Level 0's right-hand side is
// the PCG residual of iteration k, S.R[k & 1].
// UPD (level 0, k >= 1): the PCG update of iteration k - 1 rides in front -- alpha = rz / pq of that iteration;
// x += alpha p and r_k = r_{k-1} - alpha q on the tile's own cells (stored; r.r accumulated where k_mgb_update would have),
// r_k recomputed on the apron.  r ping-pongs (S.R[(k - 1) & 1] -> S.R[k & 1]): a tile's apron is another tile's interior,
// which may or may not have been rewritten yet.  The separate update kernel streamed 73 B per unknown at the HBM ceiling;
// here its loads travel with a kernel that waits for latency, not for bytes, and r is not read twice.
template <bool L0, bool UPD>
__global__ __launch_bounds__(256) void k_mgb_restrict(const VmMgbSys *__restrict__ sys, int l, int k, uint64_t active)
{
    static_assert(L0 || !UPD, "the PCG update rides on level 0 only");
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &FL = S.lv[l], &C = S.lv[l + 1];
    const bool live = (int)blockIdx.x < G(FL.ntiles)[0];
    if (!UPD && !live)
        return;
    double rr[3] = {0, 0, 0};
    // UPD: every load of the tile is ISSUED before anything is computed or stored (the stores to x and r could alias the
    // loads as far as the compiler knows: cell by cell, each cell's loads would wait for the previous cell's stores -- six
    // dependent round trips), and before alpha is looked up: the scalars' round trip and barrier pass under the loads
    constexpr int NH = (NHALO + 255) / 256;                  // apron cells per thread: 2
    float4 o_r[4], o_q[4], o_p[4], o_x[4], h_r[NH], h_q[NH];
    uint32_t o_m[4], h_m[NH];
    size_t o_i[4];
    int x0 = 0, y0 = 0;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
    if constexpr (UPD) {
        if (live) {
            const uint32_t tb = G(FL.tiles)[blockIdx.x];
            x0 = (int)(tb & 0xffffu) * TW;
            y0 = (int)(tb >> 16) * TH;
            const VM_G uint8_t *__restrict__ info = G(FL.info);
            const VmV3 *__restrict__ Ro = S.R[(k - 1) & 1], *__restrict__ Q = S.Q, *__restrict__ P = S.P[(k - 1) & 1], *__restrict__ X = S.X;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x0 + tx, y = y0 + 4 * ty + j;
                const bool in = x < FL.w && y < FL.h;
                o_i[j] = in ? (size_t)y * FL.w + x : 0;
                o_m[j] = in ? (uint32_t)info[o_i[j]] : 0u;
                o_r[j] = ld3(Ro, o_i[j]); o_q[j] = ld3(Q, o_i[j]); o_p[j] = ld3(P, o_i[j]); o_x[j] = ld3(X, o_i[j]);
            }
#pragma unroll
            for (int e = 0; e < NH; ++e) {
                const int kk = tid + 256 * e;
                h_m[e] = 0u;
                h_r[e] = h_q[e] = make_float4(0, 0, 0, 0);
                if (kk < NHALO) {
                    const int c = halo_cell(kk), x = x0 - HL + c % LW, y = y0 - HL + c / LW;
                    const bool in = x >= 0 && x < FL.w && y >= 0 && y < FL.h;
                    const size_t ii = in ? (size_t)y * FL.w + x : 0;
                    h_m[e] = in ? (uint32_t)info[ii] : 0u;
                    h_r[e] = ld3(Ro, ii); h_q[e] = ld3(Q, ii);
                }
            }
        }
    }
    float al[3] = {0, 0, 0};
    if constexpr (UPD) {        // (what k_mgb_update(k - 1) does first: its scalars, its clears)
        const int par = (k - 1) & 1;
        if (blockIdx.x == 0) {
            slot_clear(S.sc->rz[par ^ 1]);
            slot_clear(S.sc->pq[par ^ 1]);
        }
        double rz[3], pq[3];
        slot_sums2(S.sc->rz[par], S.sc->pq[par], rz, pq);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            al[c] = pq[c] > 0 ? (float)(rz[c] / pq[c]) : 0.0f;
    }
    if (live) {
        if constexpr (!UPD) {
            const uint32_t tb = G(FL.tiles)[blockIdx.x];
            x0 = (int)(tb & 0xffffu) * TW;
            y0 = (int)(tb >> 16) * TH;
        }
        const Op<L0> F(FL);
        __shared__ float4 vals[LN];
        __shared__ TileOp<L0> op;
        if constexpr (UPD) {
            VmV3 *Rn = S.R[k & 1], *X = S.X;
            // window cell c, its operator byte m, r_{k-1} and q there: stage r_k (zeros without an unknown) like stage_cell stages b
            auto stage_upd = [&](int c, bool red, uint32_t m, float4 ro, float4 q) {
                op.m[c] = (uint8_t)m;
                const float inv = k_of_dg0(m >> 4);
                // (p, q are only written inside blocks that hold an unknown: selected, never trusted, elsewhere)
                const float4 r = f4_sel(inv > 0.0f, make_float4(ro.x - al[0] * q.x, ro.y - al[1] * q.y, ro.z - al[2] * q.z, 0), make_float4(0, 0, 0, 0));
                float4 v = red ? make_float4(inv * r.x, inv * r.y, inv * r.z, 0) : r;
                v.w = inv;
                vals[c] = v;
                return r;
            };
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 r = stage_upd((HL + 4 * ty + j) * LW + HL + tx, ((tx + j) & 1) == 0, o_m[j], o_r[j], o_q[j]);     // x0, y0 are even
                if ((o_m[j] >> 4) != 0) {
                    st3(X, o_i[j], make_float4(o_x[j].x + al[0] * o_p[j].x, o_x[j].y + al[1] * o_p[j].y, o_x[j].z + al[2] * o_p[j].z, 0));
                    st3(Rn, o_i[j], r);
                    rr[0] += (double)r.x * r.x; rr[1] += (double)r.y * r.y; rr[2] += (double)r.z * r.z;
                }
            }
#pragma unroll
            for (int e = 0; e < NH; ++e) {
                const int kk = tid + 256 * e;
                if (kk < NHALO) {
                    const int c = halo_cell(kk);
                    stage_upd(c, ((c % LW + c / LW) & 1) == 0, h_m[e], h_r[e], h_q[e]);                                    // window parity == grid parity
                }
            }
        } else {
            const VmV3 *__restrict__ bsrc = L0 ? S.R[k & 1] : FL.b;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                stage_cell<L0>(FL, bsrc, F, op, vals, (HL + 4 * ty + j) * LW + HL + tx, x0 + tx, y0 + 4 * ty + j, nullptr, 0);
            for (int kk = tid; kk < NHALO; kk += 256) {
                const int c = halo_cell(kk);
                stage_cell<L0>(FL, bsrc, F, op, vals, c, x0 - HL + c % LW, y0 - HL + c / LW, nullptr, 0);
            }
        }
        __syncthreads();
        // black half-sweep: the thread's own two black cells, and the black cells of the ring at distance 1.  A black
        // cell reads red neighbours only and writes itself: in place.
        const int jb = (tx & 1) ^ 1;                         // own cells j = jb, jb + 2 are black ((tx + j) odd)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = (HL + 4 * ty + jb + 2 * m) * LW + HL + tx;
            float4 xn;
            black_update<L0>(op, vals, c, xn);
            vals[c] = xn;
        }
        if (tid < NRING1) {
            const int c = ring1_cell(tid);
            if (((c % LW + c / LW) & 1) != 0) {              // window parity == grid parity (HL, x0, y0 are even)
                float4 xn;
                black_update<L0>(op, vals, c, xn);
                vals[c] = xn;
            }
        }
        __syncthreads();
        // residual on the thread's two red cells, summed over the aggregate (its other red cell is the neighbouring lane's)
        const int jr = tx & 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = 4 * ty + jr + 2 * m, c = (HL + row) * LW + HL + tx;
            float4 r = vals[c].w > 0.0f ? op.nbsum(vals, c) : make_float4(0, 0, 0, 0);
            r.x += __shfl_xor(r.x, 1); r.y += __shfl_xor(r.y, 1); r.z += __shfl_xor(r.z, 1);
            const int X = (x0 + tx) >> 1, Y = (y0 + row) >> 1;
            if ((tx & 1) == 0 && X < C.w && Y < C.h)
                st3(C.b, (size_t)Y * C.w + X, r);
        }
    }
    if constexpr (UPD)
        block_sum3(rr[0], rr[1], rr[2], S.sc->rr[(k - 1) & 1]);
}

// F.x = black, red post-smoothing of x + P C.x (x = the red-black pre-smoothing of F.b from zero), over F's tile list;
// level 0 (L0): rz[k & 1] += F.b . F.x  = r.z
template <bool L0>
__global__ __launch_bounds__(256) void k_mgb_prolong(const VmMgbSys *__restrict__ sys, int l, int k, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &FL = S.lv[l], &C = S.lv[l + 1];
    const bool live = (int)blockIdx.x < G(FL.ntiles)[0];
    double rz[3] = {0, 0, 0};
    if (live) {
        const uint32_t tb = G(FL.tiles)[blockIdx.x];
        const int x0 = (int)(tb & 0xffffu) * TW, y0 = (int)(tb >> 16) * TH;
        const Op<L0> F(FL);
        __shared__ float4 vals[LN];
        __shared__ TileOp<L0> op;
        const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
        const VmV3 *__restrict__ bsrc = L0 ? S.R[k & 1] : FL.b;     // level 0: the PCG residual of iteration k
        float4 bown[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bown[j] = stage_cell<L0>(FL, bsrc, F, op, vals, (HL + 4 * ty + j) * LW + HL + tx, x0 + tx, y0 + 4 * ty + j, C.x, C.w);
        for (int kk = tid; kk < NHALO; kk += 256) {
            const int c = halo_cell(kk);
            stage_cell<L0>(FL, bsrc, F, op, vals, c, x0 - HL + c % LW, y0 - HL + c / LW, C.x, C.w);
        }
        __syncthreads();
        // (the cells' results and right-hand sides sit in registers indexed by j: the colour-dependent cell is picked by
        //  selects -- a run-time index would put the arrays in scratch)
        const int jb = (tx & 1) ^ 1, jr = tx & 1;
        const bool odd = (tx & 1) != 0;                  // own cells 1, 3 are red, 0, 2 black
        float4 out[4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = (HL + 4 * ty + jb + 2 * m) * LW + HL + tx;
            float4 xn;
            black_update<L0>(op, vals, c, xn);
            vals[c] = xn;
            out[2 * m] = xn;                             // the black one of the pair (2 m, 2 m + 1); the red one follows
            out[2 * m + 1] = xn;
        }
        if (tid < NRING1) {
            const int c = ring1_cell(tid);
            if (((c % LW + c / LW) & 1) != 0) {
                float4 xn;
                black_update<L0>(op, vals, c, xn);
                vals[c] = xn;
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = (HL + 4 * ty + jr + 2 * m) * LW + HL + tx;
            const float inv = vals[c].w;
            const float4 s = op.nbsum(vals, c), b = f4_sel(odd, bown[2 * m + 1], bown[2 * m]);
            const float4 xr = f4_sel(inv > 0.0f, make_float4(inv * (b.x + s.x), inv * (b.y + s.y), inv * (b.z + s.z), 0), make_float4(0, 0, 0, 0));
            out[2 * m] = f4_sel(odd, out[2 * m], xr);            // even tx: cell 2 m is red
            out[2 * m + 1] = f4_sel(odd, xr, out[2 * m + 1]);    // odd tx: cell 2 m + 1 is red
        }
        const int fw = FL.w, fh = FL.h;
        VmV3 *const fx = FL.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + tx, y = y0 + 4 * ty + j;
            if (x < fw && y < fh) {
                st3(fx, (size_t)y * fw + x, out[j]);
                if (L0) {       // b is zero where there is no unknown (stage_cell)
                    rz[0] += (double)bown[j].x * out[j].x; rz[1] += (double)bown[j].y * out[j].y; rz[2] += (double)bown[j].z * out[j].z;
                }
            }
        }
    }
    if (L0)
        block_sum3(rz[0], rz[1], rz[2], S.sc->rz[k & 1]);
}

// ---------------------------------------------------------------------------
// The same two kernels with TWO red-black sweeps each way (VmMgbLevel::nu == 2: V(2,2)).  The dependence cone of a cell
// grows by one cell per half-sweep, so the window carries a four-cell apron (72 x 24) and every half-sweep runs over the
// cells of its colour within a shrinking distance of the tile.  Written in differences, the right-hand side enters the
// first two half-sweeps only:
//   pre (from zero):   x_r1 = b_r / dg_r                          (red,   staged, distance <= 4)
//                      x_b1 = (b_b + sum w x_r1) / dg_b           (black, <= 3)
//                      d_r  = x_r2 - x_r1 = (sum w x_b1) / dg_r   (red,   <= 2)   x_r2 of the tile's own cells -> xr (packed)
//                      d_b  = x_b2 - x_b1 = (sum w d_r) / dg_b    (black, <= 1)
//                      residual: zero on black cells, sum w d_b on red ones (the tile's)
//   post of x_r2 + P xc (red cells; the black ones are recomputed from them):
//                      x_b3 = (b_b + sum w (x_r2 + xc)) / dg_b    (black, <= 3)
//                      e_r  = x_r3 - (x_r2 + xc) = (b_r + sum w x_b3) / dg_r - (x_r2 + xc)   (red, <= 2)
//                      x_b4 = x_b3 + (sum w e_r) / dg_b           (black, <= 1)
//                      x_r4 = (b_r + sum w x_b4) / dg_r           (red, the tile's)
// every half-sweep reads the other colour only and writes its own cells: in place, one float4 per window cell.
// Costs 27 % more staged cells and 6 + 10 bytes per cell for xr; on the 2304 x 1464 canvas two sweeps on every level take
// 7 PCG iterations to 1e-5 where one takes 11, two from level 2 down 9 (tools/exp/mg_prototype.py; vm_mgb.h: the choice).

constexpr int HL2 = 4, LW2 = TW + 2 * HL2, LH2 = TH + 2 * HL2, LN2 = LW2 * LH2, NHALO2 = LN2 - TW * TH;   // 72 x 24, 704 apron cells
constexpr int NSLOT2 = (NHALO2 + 255) / 256;             // apron cells per thread: 3
// (the weight records of the window's last row are never read -- a cell within 3 of the tile looks west and north, never at
//  the row below the apron's last -- and leaving them out brings the coarse levels' window to 40 KB: four workgroups per CU)
template <bool L0> using WideOp = TileOpT<L0, LW2, LN2 - LW2>;

// window index of apron cell number k (0 .. NHALO2 - 1): four rows above, four below, four columns left and right
__device__ __forceinline__ int halo2_cell(int k)
{
    if (k < HL2 * LW2) return k;
    k -= HL2 * LW2;
    if (k < HL2 * LW2) return (LH2 - HL2) * LW2 + k;
    k -= HL2 * LW2;
    const int row = HL2 + (k >> 3), q = k & 7;           // 8 cells per interior row: columns 0 .. 3, LW2 - 4 .. LW2 - 1
    return row * LW2 + (q < HL2 ? q : LW2 - 2 * HL2 + q);
}
// distance (maximum norm) of window cell (cx, cy) from the tile
__device__ __forceinline__ int tile_dist(int cx, int cy)
{
    const int dx = max(max(HL2 - cx, cx - (HL2 + TW - 1)), 0), dy = max(max(HL2 - cy, cy - (HL2 + TH - 1)), 0);
    return max(dx, dy);
}

// operator record of window cell c = grid cell (x, y): returns 1 / dg (0: no unknown, or outside the grid)
template <bool L0>
__device__ __forceinline__ float wide_stage_op(const VmMgbLevel &L, const Op<L0> &A, WideOp<L0> &op, int c, int x, int y, bool &in, size_t &ii)
{
    in = x >= 0 && x < L.w && y >= 0 && y < L.h;
    ii = in ? (size_t)y * L.w + x : 0;
    if constexpr (L0) {
        const uint32_t m = in ? (uint32_t)A.info[ii] : 0u;
        if (c < LN2 - LW2)
            op.m[c] = (uint8_t)m;
        return k_of_dg0(m >> 4);
    } else {
        const float k = A.pk[ii], we = A.pwe[ii], ws = A.pws[ii];
        if (c < LN2 - LW2)
            op.w[c] = make_float2(in && x + 1 < L.w ? we : 0.0f, in && y + 1 < L.h ? ws : 0.0f);
        return in ? k : 0.0f;
    }
}

// one half-sweep: f(window cell, index, own) for the thread's two own cells of the colour (index = 0, 1: the pair of
// rows) and for its apron cells of that colour within maxd of the tile (index = the slot)
template <class Fn>
__device__ __forceinline__ void wide_sweep(int tx, int ty, int tid, bool black, int maxd, Fn f)
{
    const int j0 = black ? ((tx & 1) ^ 1) : (tx & 1);    // own rows 4 ty + j: black <=> (tx + j) odd
#pragma unroll
    for (int m = 0; m < 2; ++m)
        f((HL2 + 4 * ty + j0 + 2 * m) * LW2 + HL2 + tx, m, true);
#pragma unroll
    for (int sl = 0; sl < NSLOT2; ++sl) {
        const int k = tid + 256 * sl;
        if (k < NHALO2) {
            const int c = halo2_cell(k), cx = c % LW2, cy = c / LW2;
            if ((((cx + cy) & 1) != 0) == black && tile_dist(cx, cy) <= maxd)
                f(c, sl, false);
        }
    }
}

template <bool L0>
__global__ __launch_bounds__(256) void k_mgb_restrict2(const VmMgbSys *__restrict__ sys, int l, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &FL = S.lv[l], &C = S.lv[l + 1];
    if ((int)blockIdx.x >= G(FL.ntiles)[0])
        return;
    const uint32_t tb = G(FL.tiles)[blockIdx.x];
    const int x0 = (int)(tb & 0xffffu) * TW, y0 = (int)(tb >> 16) * TH;
    const Op<L0> F(FL);
    __shared__ float4 vals[LN2];
    __shared__ WideOp<L0> op;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
    const float4 zero = make_float4(0, 0, 0, 0);
    // stage: red cells b / dg, black cells b, 1 / dg in .w (no unknown: zeros, selected)
    auto stage = [&](int c, int x, int y) {
        bool in;
        size_t ii;
        const float inv = wide_stage_op<L0>(FL, F, op, c, x, y, in, ii);
        const float4 b = f4_sel(inv > 0.0f, ld3(FL.b, ii), zero);
        float4 v = ((x + y) & 1) == 0 ? f4_scale(inv, b) : b;
        v.w = inv;
        vals[c] = v;
    };
#pragma unroll
    for (int j = 0; j < 4; ++j)
        stage((HL2 + 4 * ty + j) * LW2 + HL2 + tx, x0 + tx, y0 + 4 * ty + j);
#pragma unroll
    for (int sl = 0; sl < NSLOT2; ++sl) {
        const int k = tid + 256 * sl;
        if (k < NHALO2) {
            const int c = halo2_cell(k);
            stage(c, x0 - HL2 + c % LW2, y0 - HL2 + c / LW2);
        }
    }
    __syncthreads();
    wide_sweep(tx, ty, tid, true, 3, [&](int c, int, bool) {           // x_b1
        const float4 me = vals[c], s = op.nbsum(vals, c);
        vals[c] = f4_sel(me.w > 0.0f, make_float4(me.w * (me.x + s.x), me.w * (me.y + s.y), me.w * (me.z + s.z), me.w), zero);
    });
    __syncthreads();
    float4 xr2[2] = {zero, zero};
    wide_sweep(tx, ty, tid, false, 2, [&](int c, int m, bool own) {    // d_r in place of x_r1; the tile's x_r2
        const float4 me = vals[c], s = op.nbsum(vals, c);
        const float4 d = f4_sel(me.w > 0.0f, make_float4(me.w * s.x, me.w * s.y, me.w * s.z, me.w), zero);
        vals[c] = d;
        if (own)
            xr2[m & 1] = make_float4(me.x + d.x, me.y + d.y, me.z + d.z, 0);
    });
    {
        const int jr = tx & 1, fw = FL.w, fh = FL.h;
        VmV3 *const xr = FL.xr;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int x = x0 + tx, y = y0 + 4 * ty + jr + 2 * m;
            if (x < fw && y < fh)
                st3(xr, ((size_t)y * fw + x) >> 1, xr2[m]);
        }
    }
    __syncthreads();
    wide_sweep(tx, ty, tid, true, 1, [&](int c, int, bool) {           // d_b in place of x_b1
        const float4 me = vals[c], s = op.nbsum(vals, c);
        vals[c] = f4_sel(me.w > 0.0f, make_float4(me.w * s.x, me.w * s.y, me.w * s.z, me.w), zero);
    });
    __syncthreads();
    // residual on the thread's two red cells, summed over the aggregate (its other red cell is the neighbouring lane's)
    const int jr = tx & 1;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int row = 4 * ty + jr + 2 * m, c = (HL2 + row) * LW2 + HL2 + tx;
        float4 r = vals[c].w > 0.0f ? op.nbsum(vals, c) : zero;
        r.x += __shfl_xor(r.x, 1); r.y += __shfl_xor(r.y, 1); r.z += __shfl_xor(r.z, 1);
        const int X = (x0 + tx) >> 1, Y = (y0 + row) >> 1;
        if ((tx & 1) == 0 && X < C.w && Y < C.h)
            st3(C.b, (size_t)Y * C.w + X, r);
    }
}

template <bool L0>
__global__ __launch_bounds__(256) void k_mgb_prolong2(const VmMgbSys *__restrict__ sys, int l, int k, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    const VmMgbLevel &FL = S.lv[l], &C = S.lv[l + 1];
    const bool live = (int)blockIdx.x < G(FL.ntiles)[0];
    double rz[3] = {0, 0, 0};
    if (live) {
        const uint32_t tb = G(FL.tiles)[blockIdx.x];
        const int x0 = (int)(tb & 0xffffu) * TW, y0 = (int)(tb >> 16) * TH;
        const Op<L0> F(FL);
        __shared__ float4 vals[LN2];
        __shared__ WideOp<L0> op;
        const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
        const float4 zero = make_float4(0, 0, 0, 0);
        const int cw = C.w;
        // stage: red cells x_r2 + xc(aggregate), black cells b; returns b where a later half-sweep of this thread needs it
        auto stage = [&](int c, int x, int y, int dist) {
            bool in;
            size_t ii;
            const float inv = wide_stage_op<L0>(FL, F, op, c, x, y, in, ii);
            const bool unk = inv > 0.0f, red = ((x + y) & 1) == 0;
            float4 b = zero, v;
            if (red ? dist <= 2 : dist <= 3)
                b = f4_sel(unk, ld3(FL.b, ii), zero);
            if (red) {
                const float4 xr = ld3(FL.xr, ii >> 1), cor = ld3(C.x, in ? (size_t)(y >> 1) * cw + (x >> 1) : 0);
                v = f4_sel(unk, make_float4(xr.x + cor.x, xr.y + cor.y, xr.z + cor.z, 0), zero);
            } else {
                v = b;
            }
            v.w = inv;
            vals[c] = v;
            return b;
        };
        float4 bown[4], bapr[NSLOT2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bown[j] = stage((HL2 + 4 * ty + j) * LW2 + HL2 + tx, x0 + tx, y0 + 4 * ty + j, 0);
#pragma unroll
        for (int sl = 0; sl < NSLOT2; ++sl) {
            const int kk = tid + 256 * sl;
            bapr[sl] = zero;
            if (kk < NHALO2) {
                const int c = halo2_cell(kk), cx = c % LW2, cy = c / LW2;
                bapr[sl] = stage(c, x0 - HL2 + cx, y0 - HL2 + cy, tile_dist(cx, cy));
            }
        }
        __syncthreads();
        const bool odd = (tx & 1) != 0;                  // own cells 1, 3 are red, 0, 2 black
        wide_sweep(tx, ty, tid, true, 3, [&](int c, int, bool) {       // x_b3
            const float4 me = vals[c], s = op.nbsum(vals, c);
            vals[c] = f4_sel(me.w > 0.0f, make_float4(me.w * (me.x + s.x), me.w * (me.y + s.y), me.w * (me.z + s.z), me.w), zero);
        });
        __syncthreads();
        wide_sweep(tx, ty, tid, false, 2, [&](int c, int m, bool own) { // e_r in place of x_r2 + xc
            const float4 me = vals[c], s = op.nbsum(vals, c);
            const float4 b = own ? f4_sel(odd, bown[(2 * m + 1) & 3], bown[(2 * m) & 3]) : bapr[m < NSLOT2 ? m : 0];
            vals[c] = f4_sel(me.w > 0.0f, make_float4(me.w * (b.x + s.x) - me.x, me.w * (b.y + s.y) - me.y, me.w * (b.z + s.z) - me.z, me.w), zero);
        });
        __syncthreads();
        float4 out[4];
        wide_sweep(tx, ty, tid, true, 1, [&](int c, int m, bool own) {  // x_b4
            const float4 me = vals[c], s = op.nbsum(vals, c);
            const float4 xn = f4_sel(me.w > 0.0f, make_float4(me.x + me.w * s.x, me.y + me.w * s.y, me.z + me.w * s.z, me.w), zero);
            vals[c] = xn;
            if (own) {
                out[(2 * m) & 3] = xn;                   // the black one of the pair (2 m, 2 m + 1); the red one follows
                out[(2 * m + 1) & 3] = xn;
            }
        });
        __syncthreads();
        const int jr = tx & 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {                    // x_r4
            const int c = (HL2 + 4 * ty + jr + 2 * m) * LW2 + HL2 + tx;
            const float inv = vals[c].w;
            const float4 s = op.nbsum(vals, c), b = f4_sel(odd, bown[2 * m + 1], bown[2 * m]);
            const float4 xr = f4_sel(inv > 0.0f, make_float4(inv * (b.x + s.x), inv * (b.y + s.y), inv * (b.z + s.z), 0), zero);
            out[2 * m] = f4_sel(odd, out[2 * m], xr);            // even tx: cell 2 m is red
            out[2 * m + 1] = f4_sel(odd, xr, out[2 * m + 1]);    // odd tx: cell 2 m + 1 is red
        }
        const int fw = FL.w, fh = FL.h;
        VmV3 *const fx = FL.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + tx, y = y0 + 4 * ty + j;
            if (x < fw && y < fh) {
                st3(fx, (size_t)y * fw + x, out[j]);
                if (L0) {       // b is zero where there is no unknown (stage)
                    rz[0] += (double)bown[j].x * out[j].x; rz[1] += (double)bown[j].y * out[j].y; rz[2] += (double)bown[j].z * out[j].z;
                }
            }
        }
    }
    if (L0)
        block_sum3(rz[0], rz[1], rz[2], S.sc->rz[k & 1]);
}

// ---------------------------------------------------------------------------
// The tail of the cycle: every level from l0 on (their cells fit VM_MGB_TAIL_X / _B / _PAIRS, vm_mgb.h) in ONE workgroup.
// All of their state lives in LDS for the whole cycle -- iterates with 1 / dg in .w, the right-hand sides of the levels
// below l0, the edge weights -- and level l0's right-hand side in registers: the ~40 dependent half-sweeps and transfers
// of the tail touch memory only at the two ends.  One workgroup on one CU: what bounds it is its own instruction stream
// and LDS round trips per half-sweep.  Measured on the four tail levels of a 1080p canvas (4410 cells), two sweeps per
// level each way: 59.6 us per cycle with threads dealt single cells, dependent weight -> value reads and an integer
// division per cell; 51.0 with all reads of a cell issued at once and float index arithmetic; 39.4 with threads dealt
// pairs of cells (below).

constexpr int TAILT = 1024;                              // threads of the tail's workgroup
constexpr int TAILKP = VM_MGB_TAIL_PAIRS / TAILT;        // pairs of cells of a tail level a thread owns at most

// A half-sweep touches the cells of ONE colour.  Threads are dealt PAIRS of cells -- (2 px, cy) and (2 px + 1, cy), of which
// exactly one is red -- so every lane of a wave has a cell in every half-sweep (dealt single cells, half the lanes idle while
// the wave still pays every LDS instruction: the tail is one workgroup on one CU, its ~40 half-sweeps are its whole time).
struct TailLevel {
    int w, h, n;
    int xo, bo;          // offsets of the level's iterate (and weights) / right-hand side in the LDS pools (bo < 0: level l0)
    int nu;              // red-black sweeps each way
    int pw, np;          // pairs per row = ceil(w / 2), pairs of the level
    float rw, rpw;       // 1 / w, 1 / pw
};
// index -> (column, row) on a grid of `per_row` columns without an integer division (~40 instructions on this hardware, per
// cell and half-sweep): (q + 1/2) / per_row lies at least 1 / (2 per_row) from an integer, the float product is off by < 1e-5
__device__ __forceinline__ void tail_split(int q, int per_row, float r, int &col, int &row)
{
    row = (int)(((float)q + 0.5f) * r);
    col = q - row * per_row;
}
__device__ __forceinline__ void tail_xy(const TailLevel &T, int i, int &cx, int &cy) { tail_split(i, T.w, T.rw, cx, cy); }
// the cell of colour `red` of pair q: its index, coordinates, whether it is the pair's second cell; false: outside the grid
__device__ __forceinline__ bool tail_cell(const TailLevel &T, int q, bool red, int &i, int &cx, int &cy, bool &second)
{
    int px;
    tail_split(q, T.pw, T.rpw, px, cy);
    second = ((cy & 1) != 0) == red;                     // red <=> (cx + cy) even
    cx = 2 * px + (second ? 1 : 0);
    i = cy * T.w + cx;
    return cx < T.w;
}

// sum over the neighbours of cell i = (cx, cy) of weight x value.  All eight LDS reads are issued at once, whatever the
// weights say (indices clamped to the level, a missing edge's weight made 0 by a select): one round trip per cell instead of
// a chain of up to seven dependent ones.  A zero weight times the finite value that stands at the clamped index adds an
// exact zero: the sums are what skipping gave.
__device__ __forceinline__ float4 tail_nbsum(const float4 *x, const float2 *wt, int i, int cx, int cy, int w, int h)
{
    const bool hw = cx > 0, hn = cy > 0, he = cx + 1 < w, hs = cy + 1 < h;
    const int iw = hw ? i - 1 : i, in = hn ? i - w : i, ie = he ? i + 1 : i, is = hs ? i + w : i;
    const float2 me = wt[i], ww = wt[iw], wn = wt[in];
    const float4 xe = x[ie], xw = x[iw], xs = x[is], xn = x[in];
    const float wE = he ? me.x : 0.0f, wW = hw ? ww.x : 0.0f, wS = hs ? me.y : 0.0f, wN = hn ? wn.y : 0.0f;
    float4 s = make_float4(wE * xe.x, wE * xe.y, wE * xe.z, 0);
    s = f4_axpy(wW, xw, s);
    s = f4_axpy(wS, xs, s);
    s = f4_axpy(wN, xn, s);
    return s;
}

// one Gauss-Seidel half-sweep of colour `red` on a tail level: x = (b + sum_nb w x) / dg.  b: the level's right-hand
// side in LDS, or (level l0) nullptr: the thread's registers, b0[k][0 / 1] = the first / second cell of its pair k.
// first: the opening red half-sweep from zero, x = b / dg.
__device__ __forceinline__ void tail_half(const TailLevel &T, float4 *x, const float2 *wt, const float4 *bl, const float4 (*b0)[2], bool red, bool first)
{
#pragma unroll
    for (int k = 0; k < TAILKP; ++k) {
        const int q = threadIdx.x + TAILT * k;
        if (q >= T.np)
            break;
        int i, cx, cy;
        bool second;
        if (!tail_cell(T, q, red, i, cx, cy, second))
            continue;
        // (every read before the test of 1 / dg: one LDS round trip per cell)
        const float4 me = x[i];
        const float4 b = bl ? bl[i] : f4_sel(second, b0[k][1], b0[k][0]);
        const float4 s = first ? make_float4(0, 0, 0, 0) : tail_nbsum(x, wt, i, cx, cy, T.w, T.h);
        if (me.w > 0.0f)                                 // no unknown: stays (0, 0, 0, 0)
            x[i] = make_float4(me.w * (b.x + s.x), me.w * (b.y + s.y), me.w * (b.z + s.z), me.w);
    }
    __syncthreads();
}

// restriction of the pre-smoothed level's residual: bc[C] = sum over the aggregate's red cells of sum_nb w x_black
__device__ __forceinline__ void tail_restrict(const TailLevel &T, const TailLevel &TC, const float4 *x, const float2 *wt, float4 *bc)
{
    for (int i = threadIdx.x; i < TC.n; i += TAILT) {
        int X, Y;
        tail_xy(TC, i, X, Y);
        float4 r = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 2; ++d) {                    // the aggregate's red cells: (2X, 2Y) and (2X + 1, 2Y + 1)
            const int cx = 2 * X + d, cy = 2 * Y + d;
            if (cx >= T.w || cy >= T.h)
                continue;
            const int q = cy * T.w + cx;
            if (!(x[q].w > 0.0f))
                continue;
            const float4 s = tail_nbsum(x, wt, q, cx, cy, T.w, T.h);
            r.x += s.x; r.y += s.y; r.z += s.z;
        }
        bc[i] = r;
    }
    __syncthreads();
}

// ... after more than one sweep: the residual of a red cell is dg x (what one more red half-sweep would change), still
// zero on the black ones; the two red cells of an aggregate add themselves to its entry (two adds commute: deterministic),
// which the caller has zeroed before the level's half-sweeps (their barriers order the zeros before the adds)
__device__ __forceinline__ void tail_restrict_any(const TailLevel &T, const TailLevel &TC, const float4 *x, const float2 *wt, const float4 *bl,
                                                  const float4 (*b0)[2], float4 *bc)
{
#pragma unroll
    for (int k = 0; k < TAILKP; ++k) {
        const int q = threadIdx.x + TAILT * k;
        if (q >= T.np)
            break;
        int i, cx, cy;
        bool second;
        if (!tail_cell(T, q, true, i, cx, cy, second))
            continue;
        const float4 me = x[i];
        const float4 b = bl ? bl[i] : f4_sel(second, b0[k][1], b0[k][0]);
        const float4 s = tail_nbsum(x, wt, i, cx, cy, T.w, T.h);
        if (!(me.w > 0.0f))
            continue;
        const float dg = 1.0f / me.w;
        float4 *dst = bc + (cy >> 1) * TC.w + (cx >> 1);
        atomicAdd(&dst->x, dg * (me.w * (b.x + s.x) - me.x));
        atomicAdd(&dst->y, dg * (me.w * (b.y + s.y) - me.y));
        atomicAdd(&dst->z, dg * (me.w * (b.z + s.z) - me.z));
    }
    __syncthreads();
}

template <bool L0>
__global__ __launch_bounds__(TAILT) void k_mgb_tail(const VmMgbSys *__restrict__ sys, int l0, uint64_t active)
{
    if (!sys_active(active))
        return;
    const VmMgbSys &S = sys[blockIdx.z];
    __shared__ float4 xp[VM_MGB_TAIL_X], bp[VM_MGB_TAIL_B];
    __shared__ float2 wp[VM_MGB_TAIL_X];
    __shared__ TailLevel T[VM_MGB_MAXLEV];
    const int nl = S.nlev - l0;                          // levels of the tail: T[0] = level l0
    if (threadIdx.x == 0) {
        int xo = 0, bo = 0;
        for (int j = 0; j < nl; ++j) {
            const VmMgbLevel &L = S.lv[l0 + j];
            const int pw = (L.w + 1) / 2;
            T[j] = TailLevel{L.w, L.h, L.w * L.h, xo, j ? bo : -1, L.nu, pw, pw * L.h, 1.0f / (float)L.w, 1.0f / (float)pw};
            xo += L.w * L.h;
            if (j) bo += L.w * L.h;
        }
    }
    __syncthreads();
    // stage: 1 / dg and the weights of the edges to the east / south of every tail cell; level l0's right-hand side, by
    // the pairs the thread will sweep
    float4 b0[TAILKP][2];
    {
        const VmMgbLevel &L = S.lv[l0];
        const Op<L0> A(L);
#pragma unroll
        for (int k = 0; k < TAILKP; ++k) {
            const int q = threadIdx.x + TAILT * k;
            b0[k][0] = b0[k][1] = make_float4(0, 0, 0, 0);
            if (q < T[0].np) {
                int px, cy;
                tail_split(q, T[0].pw, T[0].rpw, px, cy);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int cx = 2 * px + e, i = cy * T[0].w + cx;
                    if (cx < T[0].w) {
                        const Stencil st = stencil_of(A, cx, cy, (size_t)i);
                        const float inv = A.k((size_t)i);
                        xp[i] = make_float4(0, 0, 0, inv);
                        wp[i] = make_float2(st.wE, st.wS);
                        if (inv > 0.0f) b0[k][e] = ld3(L.b, (size_t)i);
                    }
                }
            }
        }
    }
    for (int j = 1; j < nl; ++j) {
        const VmMgbLevel &L = S.lv[l0 + j];
        const Op<false> A(L);
        for (int i = threadIdx.x; i < T[j].n; i += TAILT) {
            int cx, cy;
            tail_xy(T[j], i, cx, cy);
            const Stencil st = stencil_of(A, cx, cy, (size_t)i);
            xp[T[j].xo + i] = make_float4(0, 0, 0, A.k((size_t)i));
            wp[T[j].xo + i] = make_float2(st.wE, st.wS);
        }
    }
    __syncthreads();
    // down: pre-smoothing (red from zero, black), restriction
    for (int j = 0; j + 1 < nl; ++j) {
        float4 *x = xp + T[j].xo;
        const float2 *wt = wp + T[j].xo;
        const float4 *b = j ? bp + T[j].bo : nullptr;
        if (T[j].nu != 1)
            for (int i = threadIdx.x; i < T[j + 1].n; i += TAILT)
                bp[T[j + 1].bo + i] = make_float4(0, 0, 0, 0);
        for (int sw = 0; sw < T[j].nu; ++sw) {
            tail_half(T[j], x, wt, b, b0, true, sw == 0);
            tail_half(T[j], x, wt, b, b0, false, false);
        }
        if (T[j].nu == 1)
            tail_restrict(T[j], T[j + 1], x, wt, bp + T[j + 1].bo);
        else
            tail_restrict_any(T[j], T[j + 1], x, wt, b, b0, bp + T[j + 1].bo);
    }
    // the coarsest grid: symmetric sweeps from zero (red, black ... then black, red ...)
    {
        const int j = nl - 1;
        float4 *x = xp + T[j].xo;
        const float2 *wt = wp + T[j].xo;
        const float4 *b = j ? bp + T[j].bo : nullptr;
        for (int sw = 0; sw < 2 * VM_MGB_COARSE_SWEEPS; ++sw) {
            const bool fwd = sw < VM_MGB_COARSE_SWEEPS;
            tail_half(T[j], x, wt, b, b0, fwd, sw == 0);
            tail_half(T[j], x, wt, b, b0, !fwd, false);
        }
    }
    // up: coarse correction on the red cells (the black half-sweep that follows overwrites the black ones from red
    // values only), post-smoothing black, red
    for (int j = nl - 2; j >= 0; --j) {
        float4 *x = xp + T[j].xo;
        const float2 *wt = wp + T[j].xo;
        const float4 *xc = xp + T[j + 1].xo;
        const float4 *b = j ? bp + T[j].bo : nullptr;
        const int cw = T[j + 1].w;
        for (int q = threadIdx.x; q < T[j].np; q += TAILT) {
            int i, cx, cy;
            bool second;
            if (!tail_cell(T[j], q, true, i, cx, cy, second))
                continue;
            const float4 f = x[i], c = xc[(cy >> 1) * cw + (cx >> 1)];
            // (a red cell without an unknown holds 0 and stays 0: its aggregate may hold unknowns and a correction)
            if (f.w > 0.0f)
                x[i] = make_float4(f.x + c.x, f.y + c.y, f.z + c.z, f.w);
        }
        __syncthreads();
        for (int sw = 0; sw < T[j].nu; ++sw) {
            tail_half(T[j], x, wt, b, b0, false, false);
            tail_half(T[j], x, wt, b, b0, true, false);
        }
    }
    const VmMgbLevel &L = S.lv[l0];
    for (int i = threadIdx.x; i < T[0].n; i += TAILT)
        st3(L.x, (size_t)i, xp[i]);
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

void vm_mgb_launch_restrict(const VmMgbSys *sys, int nsys, int l, int nu, int nt_fine, int k, bool upd, uint64_t active, hipStream_t s)
{
    const dim3 grid(nt_fine, 1, nsys);
    if (nu == 2) {
        if (l == 0)
            hipLaunchKernelGGL(k_mgb_restrict2<true>, grid, blk2, 0, s, sys, l, active);
        else
            hipLaunchKernelGGL(k_mgb_restrict2<false>, grid, blk2, 0, s, sys, l, active);
    } else if (l == 0 && upd) {
        hipLaunchKernelGGL((k_mgb_restrict<true, true>), grid, blk2, 0, s, sys, l, k, active);
    } else if (l == 0) {
        hipLaunchKernelGGL((k_mgb_restrict<true, false>), grid, blk2, 0, s, sys, l, k, active);
    } else {
        hipLaunchKernelGGL((k_mgb_restrict<false, false>), grid, blk2, 0, s, sys, l, k, active);
    }
}

void vm_mgb_launch_prolong(const VmMgbSys *sys, int nsys, int l, int nu, int nt_fine, int k, uint64_t active, hipStream_t s)
{
    const dim3 grid(nt_fine, 1, nsys);
    if (nu == 2) {
        if (l == 0)
            hipLaunchKernelGGL(k_mgb_prolong2<true>, grid, blk2, 0, s, sys, l, k, active);
        else
            hipLaunchKernelGGL(k_mgb_prolong2<false>, grid, blk2, 0, s, sys, l, k, active);
    } else if (l == 0) {
        hipLaunchKernelGGL(k_mgb_prolong<true>, grid, blk2, 0, s, sys, l, k, active);
    } else {
        hipLaunchKernelGGL(k_mgb_prolong<false>, grid, blk2, 0, s, sys, l, k, active);
    }
}

void vm_mgb_launch_tail(const VmMgbSys *sys, int nsys, int l, uint64_t active, hipStream_t s)
{
    if (l == 0)
        hipLaunchKernelGGL(k_mgb_tail<true>, dim3(1, 1, nsys), dim3(TAILT), 0, s, sys, l, active);
    else
        hipLaunchKernelGGL(k_mgb_tail<false>, dim3(1, 1, nsys), dim3(TAILT), 0, s, sys, l, active);
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
