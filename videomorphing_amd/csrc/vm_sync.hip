// vm_sync.hip -- the synchronisation stage on gfx950 ("(later)" row):
//   CSyncThread::optimize_level (Algorithm/SyncThread.cpp:290-480): three conjugate-gradient
//   solves (x, y and frame displacement) of one sparse system A = 2 w_tps (thin-plate operators in
//   x, y, t) + diag(UI), per level of the sync pyramid;  Kernel_upsample (upsample.cu:343-375);
//   CSyncThread::update_result (SyncThread.cpp:482-521);  kernel_render_resample_image0/1
//   (render.cu:99-199).
//
// The reference assembles A in CSR on the host (genMatrix, :129-289: every voxel visits every
// constraint), ships it, and runs cusparseScsrmv + 5 cuBLAS level-1 calls per component and
// iteration with a host round trip for each of the 2 dot products.  Here the matrix is never
// formed: a row is 24 off-diagonal values that depend only on the voxel's border state (a 125-row
// table) plus a per-voxel diagonal, and ONE CG iteration of all three components is two launches:
//   k_sync_A  p = r + beta p (recomputed on the halo), omega = A p from an LDS brick, p.omega
//   k_sync_B  x += alpha p, r -= alpha omega, r.r
// with the scalars (alpha, beta, r0, r1, "is this component still active") living on the device:
// the host only enqueues.  HBM-bound: 124 B per voxel and iteration.
//
// Arithmetic: built with -ffp-contract=off; every sum is evaluated in ONE fixed order (DESIGN.md
// 3.9: rows in CSR order, dot products as float products accumulated in double over 32x8x8 bricks
// by a fixed tree), so the result is independent of scheduling and bit-comparable with the CPU
// restatement the tests hold it against.  Brick partials cross workgroups by write-through (sc1) stores +
// an arrival ticket; the workgroup that arrives last folds them (no extra launch, no host sync).
#include "vm_sync.h"

namespace {

__device__ __forceinline__ int sync_state(int p, int n)
{
    if (n <= 5 || p < 2) return p;
    if (p == n - 1) return 4;
    if (p == n - 2) return 3;
    return 2;
}

// Consecutive workgroup ids land on different XCDs; give every XCD a contiguous run of bricks so
// the halo lines neighbouring bricks share are fetched into ONE L2.
__device__ __forceinline__ int brick_of_block(const VmSyncGrid &g) { return (blockIdx.x & 7) * g.per_xcd + (blockIdx.x >> 3); }

__device__ __forceinline__ double wave_butterfly(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// sum over the 256 threads: butterfly inside each wave, the four waves in sequence
__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_butterfly(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double t = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return t;
}

// One lane publishes the workgroup's partial write-through, drains it, and takes a ticket; the
// workgroup whose ticket is the last one may read every partial (with sc1 loads, after the
// barrier the ticket-taking wave joins).  MI355X_MICROARCH.md, "inter-workgroup visibility".
// Tickets are two-level -- 32 workgroups share a counter (each on a line of its own), the last of
// a group arrives at the component's top counter -- because agent-scope atomics on ONE line
// serialise at ~45 ns each (measured: 6600 arrivals on one line held a 60 us kernel for 350 us).
#define VM_TK_GROUP 32
#define VM_TK_STRIDE 32 // words between counters: one 128-B line each
__device__ __forceinline__ bool publish_and_arrive(const VmSyncSys &S, int c, int nb, int lin, bool has, double part, int *s_last)
{
    if (threadIdx.x == 0) {
        if (has)
            __hip_atomic_store((unsigned long long *)&S.part[(size_t)c * nb + lin], (unsigned long long)__double_as_longlong(part),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ngr = (gridDim.x + VM_TK_GROUP - 1) / VM_TK_GROUP, grp = blockIdx.x / VM_TK_GROUP;
        const unsigned gsize = min((unsigned)VM_TK_GROUP, gridDim.x - grp * VM_TK_GROUP);
        unsigned *tk = S.ticket + (size_t)c * (ngr + 1) * VM_TK_STRIDE;
        int last = 0;
        if (__hip_atomic_fetch_add(tk + (size_t)(1 + grp) * VM_TK_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
            __hip_atomic_store(tk + (size_t)(1 + grp) * VM_TK_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ngr == 1) // one group: its last arrival is the last of the launch
                last = 1;
            else if (__hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ngr - 1) {
                __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        *s_last = last;
    }
    __syncthreads();
    return *s_last != 0;
}

// brick partials -> total: thread t takes partials t, t + 256, ... in sequence, then block_sum
__device__ __forceinline__ double total_of(const double *part, int nb, double *red)
{
    double s = 0;
    for (int i = threadIdx.x; i < nb; i += 256) {
        const unsigned long long u =
            __hip_atomic_load((const unsigned long long *)&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s += __longlong_as_double((long long)u);
    }
    return block_sum(s, red);
}

#define TILE_X (VM_SB_X + 4)
#define TILE_Y (VM_SB_Y + 4)
#define TILE_Z (VM_SB_Z + 4)
#define TILE_P (TILE_X * TILE_Y)
#define TILE_N (TILE_Z * TILE_P)
#define TILE_LOADS ((TILE_N + 255) / 256)

#define TILE_ROW_LOADS ((TILE_P + 255) / 256) // in-plane slots per thread: 432 = 256 + 176

// brick partials of p . omega (kernel A) and of r . r (kernel B, ping-pong by iteration parity)
__device__ __forceinline__ double *part_a(const VmSyncSys &S, int nb, int c) { return S.part + (size_t)c * nb; }
__device__ __forceinline__ double *part_b(const VmSyncSys &S, int nb, int par, int c) { return S.part + (size_t)(3 + par * 3 + c) * nb; }

// the same fold as total_of on partials a PREVIOUS launch stored (plain loads)
__device__ __forceinline__ double total_plain(const double *part, int nb, double *red)
{
    double s = 0;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    return block_sum(s, red);
}

// Iteration k (1-based), component blockIdx.y: p <- r (k == 1) or r + beta p; omega = A p;
// dot = p . omega.
// TABLDS: the coefficient table is staged in LDS (launch-bound small levels: no round trip when
// the border state changes along z) or read through L1 (large levels: 12.5 KB less LDS keeps 7
// workgroups per CU, which is what hides the latency there).
// SELF (levels of <= 512 bricks): no tickets -- every workgroup folds the partials the PREVIOUS
// launch left (same fixed order, so every workgroup gets the same bits) while its tile loads are
// in flight, and ends by storing its own partial; the launch boundary publishes it.  Saves the
// ~4 us store-drain + atomic + re-read tail per launch that a launch-bound level cannot hide.
template <bool FIRST, bool TABLDS, bool SELF>
__global__ __launch_bounds__(256) void k_sync_A(VmSyncSys S, VmSyncGrid g, int k)
{
    __shared__ float tile[TILE_N];
    __shared__ float tab_s[TABLDS ? 125 * 25 : 1];
    __shared__ double red[4];
    __shared__ int s_last;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int c = blockIdx.y;
    const int q = brick_of_block(g);
    const bool has = q < g.nb;
    if (SELF && !has) return;
    const int cur = (k + 1) & 1; // r1 as iteration k - 1 left it
    const float tol = 1e-12f;
    // dispatch order is y-slab major (an XCD's run of bricks spans all of z: the z halo, a third of
    // the tile, is then shared inside ONE L2); the partial's slot keeps the z-major brick order
    const int qc = has ? q : 0;
    const int by = qc / (g.nbx * g.nbz), rem = qc - by * (g.nbx * g.nbz), bz = rem / g.nbx, bx = rem - bz * g.nbx;
    const int lin = has ? (bz * g.nby + by) * g.nbx + bx : g.nb;
    const int x0 = bx * VM_SB_X, y0 = by * VM_SB_Y, z0 = bz * VM_SB_Z;
    const int x = x0 + tx, y = y0 + ty;
    const bool mine = has && x < g.w && y < g.h;
    const int sx = sync_state(mine ? x : 0, g.w), sy = sync_state(mine ? y : 0, g.h);
    const int plane = g.w * g.h;
    const int col = y * g.w + x;
    const float *__restrict__ rc = S.r[c];
    const float *__restrict__ po = S.p[(k + 1) & 1][c];
    float *__restrict__ pn = S.p[k & 1][c];
    float *__restrict__ om = S.om[c];
    // Stage p over the brick + halo.  A thread owns the same <= 2 in-plane slots on every plane
    // (their offsets are worked out once), and every load of the workgroup -- the tile, the
    // diagonal, the coefficient table, in SELF mode the partials -- is issued before the first use.
    int off_xy[TILE_ROW_LOADS];
    bool ok_xy[TILE_ROW_LOADS];
#pragma unroll
    for (int j = 0; j < TILE_ROW_LOADS; ++j) {
        const int i = t + 256 * j, ly = i / TILE_X, lx = i - ly * TILE_X;
        const int gx = x0 + lx - 2, gy = y0 + ly - 2;
        ok_xy[j] = has && i < TILE_P && gx >= 0 && gx < g.w && gy >= 0 && gy < g.h;
        off_xy[j] = gy * g.w + gx;
    }
    float rv[TILE_Z][TILE_ROW_LOADS], pv[TILE_Z][TILE_ROW_LOADS];
#pragma unroll
    for (int lz = 0; lz < TILE_Z; ++lz) {
        const int gz = z0 + lz - 2;
        const bool zin = gz >= 0 && gz < g.d;
#pragma unroll
        for (int j = 0; j < TILE_ROW_LOADS; ++j) {
            const bool in = zin && ok_xy[j];
            const size_t gi = in ? (size_t)gz * plane + off_xy[j] : 0;
            rv[lz][j] = in ? rc[gi] : 0.0f;
            pv[lz][j] = (!FIRST && in) ? po[gi] : 0.0f;
        }
    }
    float dg[VM_SB_Z];
#pragma unroll
    for (int zz = 0; zz < VM_SB_Z; ++zz) dg[zz] = (mine && z0 + zz < g.d) ? S.diag[(size_t)(z0 + zz) * plane + col] : 0.0f;
    if (TABLDS)
        for (int i = t; i < 125 * 25; i += 256) tab_s[i] = S.tab[i];
    float r1c, r0;
    if (SELF) {
        r1c = (float)total_plain(part_b(S, g.nb, cur, c), g.nb, red);
        r0 = FIRST ? 0.0f : (float)total_plain(part_b(S, g.nb, k & 1, c), g.nb, red);
    } else {
        r1c = S.sc[3 + cur * 3 + c];
        r0 = S.sc[c];
    }
    if (!(r1c > tol * tol)) return; // this component is finished (SyncThread.cpp:382)
    const float beta = FIRST ? 0.0f : r1c / r0;
    double part = 0;
    if (has) {
#pragma unroll
        for (int lz = 0; lz < TILE_Z; ++lz)
#pragma unroll
            for (int j = 0; j < TILE_ROW_LOADS; ++j) {
                float v = rv[lz][j];
                if (!FIRST) {
                    const float tb = beta * pv[lz][j];  // cublasSscal
                    v = fmaf(1.0f, rv[lz][j], tb);      // cublasSaxpy(1, r, p)
                }
                if (t + 256 * j < TILE_P) tile[lz * TILE_P + t + 256 * j] = v;
            }
        __syncthreads();
        double acc = 0;
        if (mine) {
            int szp = -1;
            float cf[25];
#pragma unroll
            for (int zz = 0; zz < VM_SB_Z; ++zz) {
                const int z = z0 + zz;
                if (z < g.d) {
                    const int sz = sync_state(z, g.d);
                    if (sz != szp) {
                        const float *row = (TABLDS ? (const float *)tab_s : S.tab) + ((sz * 5 + sy) * 5 + sx) * 25;
#pragma unroll
                        for (int q = 0; q < 25; ++q) cf[q] = row[q];
                        szp = sz;
                    }
                    const float *T = tile + ((zz + 2) * TILE_Y + (ty + 2)) * TILE_X + (tx + 2);
                    // the row in CSR order (z, then y, then x ascending), one FMA per entry
                    float sum = 0.0f;
                    sum = fmaf(cf[0], T[-2 * TILE_P], sum);
                    sum = fmaf(cf[1], T[-TILE_P - TILE_X], sum);
                    sum = fmaf(cf[2], T[-TILE_P - 1], sum);
                    sum = fmaf(cf[3], T[-TILE_P], sum);
                    sum = fmaf(cf[4], T[-TILE_P + 1], sum);
                    sum = fmaf(cf[5], T[-TILE_P + TILE_X], sum);
                    sum = fmaf(cf[6], T[-2 * TILE_X], sum);
                    sum = fmaf(cf[7], T[-TILE_X - 1], sum);
                    sum = fmaf(cf[8], T[-TILE_X], sum);
                    sum = fmaf(cf[9], T[-TILE_X + 1], sum);
                    sum = fmaf(cf[10], T[-2], sum);
                    sum = fmaf(cf[11], T[-1], sum);
                    sum = fmaf(dg[zz], T[0], sum);
                    sum = fmaf(cf[13], T[1], sum);
                    sum = fmaf(cf[14], T[2], sum);
                    sum = fmaf(cf[15], T[TILE_X - 1], sum);
                    sum = fmaf(cf[16], T[TILE_X], sum);
                    sum = fmaf(cf[17], T[TILE_X + 1], sum);
                    sum = fmaf(cf[18], T[2 * TILE_X], sum);
                    sum = fmaf(cf[19], T[TILE_P - TILE_X], sum);
                    sum = fmaf(cf[20], T[TILE_P - 1], sum);
                    sum = fmaf(cf[21], T[TILE_P], sum);
                    sum = fmaf(cf[22], T[TILE_P + 1], sum);
                    sum = fmaf(cf[23], T[TILE_P + TILE_X], sum);
                    sum = fmaf(cf[24], T[2 * TILE_P], sum);
                    const size_t gi = (size_t)z * plane + col;
                    const float pc = T[0];
                    om[gi] = sum;
                    pn[gi] = pc;
                    const float pr = pc * sum;
                    acc += (double)pr;
                }
            }
        }
        part = block_sum(acc, red);
    }
    if (SELF) {
        if (t == 0) part_a(S, g.nb, c)[lin] = part;
    } else if (publish_and_arrive(S, c, g.nb, lin, has, part, &s_last)) {
        const double tot = total_of(part_a(S, g.nb, c), g.nb, red);
        if (t == 0) S.sc[9 + c] = (float)tot;
    }
}

// component blockIdx.y: x += alpha p, r -= alpha omega, r1 = r . r  (INIT: only r1 = r . r before
// the first iteration).  SELF as in k_sync_A; the r . r partials ping-pong by iteration parity so
// that the next k_sync_A can fold both r1 (this launch's) and r0 (the previous one's).
template <bool INIT, bool SELF>
__global__ __launch_bounds__(256) void k_sync_B(VmSyncSys S, VmSyncGrid g, int k)
{
    __shared__ double red[4];
    __shared__ int s_last;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int q = brick_of_block(g);
    const bool has = q < g.nb;
    if (SELF && !has) return;
    const int cur = (k + 1) & 1, nxt = k & 1;
    const float tol = 1e-12f;
    const int c = blockIdx.y;
    const int qc = has ? q : 0; // same brick order as k_sync_A
    const int by = qc / (g.nbx * g.nbz), rem = qc - by * (g.nbx * g.nbz), bz = rem / g.nbx, bx = rem - bz * g.nbx;
    const int lin = has ? (bz * g.nby + by) * g.nbx + bx : g.nb;
    const int x = bx * VM_SB_X + tx, y = by * VM_SB_Y + ty, z0 = bz * VM_SB_Z;
    const bool mine = has && x < g.w && y < g.h;
    const size_t plane = (size_t)g.w * g.h;
    const size_t col = (size_t)y * g.w + x;
    float *__restrict__ xc = S.x[c];
    float *__restrict__ rc = S.r[c];
    const float *__restrict__ pn = S.p[k & 1][c];
    const float *__restrict__ om = S.om[c];
    float pv[VM_SB_Z], ov[VM_SB_Z], xv[VM_SB_Z], rv[VM_SB_Z];
#pragma unroll
    for (int zz = 0; zz < VM_SB_Z; ++zz) {
        const bool in = mine && z0 + zz < g.d;
        const size_t gi = in ? (size_t)(z0 + zz) * plane + col : 0;
        pv[zz] = (!INIT && in) ? pn[gi] : 0.0f;
        ov[zz] = (!INIT && in) ? om[gi] : 0.0f;
        xv[zz] = (!INIT && in) ? xc[gi] : 0.0f;
        rv[zz] = in ? rc[gi] : 0.0f;
    }
    float r1c = 0.0f, dot = 0.0f;
    if (!INIT) {
        if (SELF) {
            r1c = (float)total_plain(part_b(S, g.nb, cur, c), g.nb, red);
            dot = (float)total_plain(part_a(S, g.nb, c), g.nb, red);
        } else {
            r1c = S.sc[3 + cur * 3 + c];
            dot = S.sc[9 + c];
        }
        if (!(r1c > tol * tol)) { // finished: r1 is carried over unchanged
            if (SELF) {
                if (t == 0) part_b(S, g.nb, nxt, c)[lin] = part_b(S, g.nb, cur, c)[lin];
            } else if (blockIdx.x == 0 && t == 0)
                S.sc[3 + nxt * 3 + c] = r1c;
            return;
        }
    }
    const float al = INIT ? 0.0f : r1c / dot, nal = -al;
    double part = 0;
    if (has) {
        double acc = 0;
        if (mine) {
            if (!INIT) {
#pragma unroll
                for (int zz = 0; zz < VM_SB_Z; ++zz)
                    if (z0 + zz < g.d) {
                        const size_t gi = (size_t)(z0 + zz) * plane + col;
                        xc[gi] = fmaf(al, pv[zz], xv[zz]);  // cublasSaxpy(alpha, p, x)
                        rv[zz] = fmaf(nal, ov[zz], rv[zz]); // cublasSaxpy(-alpha, omega, r)
                        rc[gi] = rv[zz];
                    }
            }
#pragma unroll
            for (int zz = 0; zz < VM_SB_Z; ++zz)
                if (z0 + zz < g.d) {
                    const float pr = rv[zz] * rv[zz];
                    acc += (double)pr;
                }
        }
        part = block_sum(acc, red);
    }
    if (SELF) {
        if (t == 0) part_b(S, g.nb, nxt, c)[lin] = part;
    } else if (publish_and_arrive(S, c, g.nb, lin, has, part, &s_last)) {
        const double tot = total_of(part_a(S, g.nb, c), g.nb, red);
        if (t == 0) {
            if (!INIT) S.sc[c] = r1c; // r0 = r1
            S.sc[3 + nxt * 3 + c] = (float)tot;
        }
    }
}

// SELF mode: r1 of the last iteration into the scalar block (for the host's read-back)
__global__ __launch_bounds__(256) void k_sync_total(VmSyncSys S, int nb, int par)
{
    __shared__ double red[4];
    const int c = blockIdx.x;
    const double tot = total_plain(part_b(S, nb, par, c), nb, red);
    if (threadIdx.x == 0) S.sc[3 + par * 3 + c] = (float)tot;
}

// the diagonal: diag[] arrives holding the UI term; the stencil's increments follow in
// genMatrix's order (second differences per axis x, y, z: +2w, +8w, +2w; then the twelve mixed
// 2x2 cells, +4w each), SyncThread.cpp:190-240
__global__ void k_sync_diag(VmSyncGrid g, float *diag, float wt)
{
    const size_t n = (size_t)g.w * g.h * g.d;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % g.w), y = (int)((i / g.w) % g.h), z = (int)(i / ((size_t)g.w * g.h));
    const float a2 = 1.0f * 2.0f * wt, a8 = 4.0f * 2.0f * wt, a4 = 2.0f * 2.0f * wt;
    float c = diag[i];
    const int pp[3] = {x, y, z}, nn[3] = {g.w, g.h, g.d};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (pp[a] > 1) c += a2;
        if (pp[a] > 0 && pp[a] < nn[a] - 1) c += a8;
        if (pp[a] < nn[a] - 2) c += a2;
    }
    const bool xl = x > 0, xh = x < g.w - 1, yl = y > 0, yh = y < g.h - 1, zl = z > 0, zh = z < g.d - 1;
    if (xl && yl) c += a4;
    if (xh && yl) c += a4;
    if (xl && yh) c += a4;
    if (xh && yh) c += a4;
    if (zl && yl) c += a4;
    if (zl && yh) c += a4;
    if (zh && yl) c += a4;
    if (zh && yh) c += a4;
    if (xl && zl) c += a4;
    if (xl && zh) c += a4;
    if (xh && zl) c += a4;
    if (xh && zh) c += a4;
    diag[i] = c;
}

__global__ void k_sync_scatter(float *dst, const int *idx, const float *val, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = val[i];
}

// tex2D, linear filtering, clamp addressing, at unnormalised coordinates: exact float weights
// (the 1.8 fixed-point weights of the texture unit are not emulated)
struct TexPos {
    int i0, i1, j0, j1;
    float a, b;
};

__device__ __forceinline__ TexPos tex_locate(int w, int h, float x, float y)
{
    TexPos t;
    const float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    t.a = xb - fi;
    t.b = yb - fj;
    fi = fminf(fmaxf(fi, -1.0f), (float)w);
    fj = fminf(fmaxf(fj, -1.0f), (float)h);
    t.i0 = min(max((int)fi, 0), w - 1);
    t.i1 = min(max((int)fi + 1, 0), w - 1);
    t.j0 = min(max((int)fj, 0), h - 1);
    t.j1 = min(max((int)fj + 1, 0), h - 1);
    return t;
}

__device__ __forceinline__ float tex_mix(const TexPos &t, float t00, float t10, float t01, float t11)
{
    const float a = t.a, b = t.b;
    return (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
}

__device__ __forceinline__ float tex1(const float *img, int w, int h, float x, float y)
{
    const TexPos t = tex_locate(w, h, x, y);
    return tex_mix(t, img[(size_t)t.j0 * w + t.i0], img[(size_t)t.j0 * w + t.i1], img[(size_t)t.j1 * w + t.i0],
                   img[(size_t)t.j1 * w + t.i1]);
}

// Kernel_upsample, upsample.cu:343-353 (with the bounds check the reference lacks)
__global__ void k_sync_upsample(float *dst, int dw, int dh, const float *src, int sw, int sh, float ratio)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= dw || y >= dh) return;
    dst += (size_t)blockIdx.z * dw * dh;
    src += (size_t)blockIdx.z * sw * sh;
    const float px = (float)((x + 0.5) / (float)dw), py = (float)((y + 0.5) / (float)dh);
    dst[(size_t)y * dw + x] = tex1(src, sw, sh, px * (float)sw, py * (float)sh) * ratio;
}

// CSyncThread::update_result, SyncThread.cpp:482-521: cv::resize(INTER_LINEAR) of
// (X ratio_x, Y ratio_y, Z, 0) to full resolution -- OpenCV 3.0's generic 32F linear resize,
// (scale = 1 / (dst / src) in double; f = (float)((dx + 0.5) scale - 0.5); border columns collapse
// onto one sample, border rows only clip their index; horizontal pass, then vertical, in float)
__global__ void k_sync_result(const float *X, const float *Y, const float *Z, int w, int h, int w0, int h0,
                              float ratio_x, float ratio_y, double sx, double sy, float4 *out)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
    if (dx >= w0 || dy >= h0) return;
    float fx = (float)((dx + 0.5) * sx - 0.5);
    int s = (int)floorf(fx);
    fx -= s;
    if (s < 0) { fx = 0; s = 0; }
    const bool edge = s + 1 >= w; // from this column on OpenCV reads one sample, times 1
    if (edge && s >= w - 1) { fx = 0; s = w - 1; }
    const float a0 = 1.f - fx, a1 = fx;
    float fy = (float)((dy + 0.5) * sy - 0.5);
    const int sy0 = (int)floorf(fy);
    fy -= sy0;
    const float b0 = 1.f - fy, b1 = fy;
    const float *src[3] = {X, Y, Z};
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float rows[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int yy = min(max(sy0 + k, 0), h - 1);
            const float *S = src[c] + (size_t)yy * w;
            float v0 = S[s], v1 = edge ? 0.0f : S[s + 1];
            if (c == 0) { v0 = v0 * ratio_x; v1 = v1 * ratio_x; }
            if (c == 1) { v0 = v0 * ratio_y; v1 = v1 * ratio_y; }
            rows[k] = edge ? v0 * 1.f : v0 * a0 + v1 * a1;
        }
        o[c] = rows[0] * b0 + rows[1] * b1;
    }
    out[(size_t)dy * w0 + dx] = make_float4(o[0], o[1], o[2], 0.0f);
}

__device__ __forceinline__ float4 tex4(const float4 *img, int w, int h, float x, float y)
{
    const TexPos t = tex_locate(w, h, x, y);
    const float4 t00 = img[(size_t)t.j0 * w + t.i0], t10 = img[(size_t)t.j0 * w + t.i1];
    const float4 t01 = img[(size_t)t.j1 * w + t.i0], t11 = img[(size_t)t.j1 * w + t.i1];
    return make_float4(tex_mix(t, t00.x, t10.x, t01.x, t11.x), tex_mix(t, t00.y, t10.y, t01.y, t11.y),
                       tex_mix(t, t00.z, t10.z, t01.z, t11.z), 0.0f);
}

__device__ __forceinline__ float2 tex2(const float2 *img, int w, int h, float x, float y)
{
    const TexPos t = tex_locate(w, h, x, y);
    const float2 t00 = img[(size_t)t.j0 * w + t.i0], t10 = img[(size_t)t.j0 * w + t.i1];
    const float2 t01 = img[(size_t)t.j1 * w + t.i0], t11 = img[(size_t)t.j1 * w + t.i1];
    return make_float2(tex_mix(t, t00.x, t10.x, t01.x, t11.x), tex_mix(t, t00.y, t10.y, t01.y, t11.y));
}

// RGBA8 frame sampled as float4 0..255 (the reference converts before the upload, pyramid.cu:93-96)
__device__ __forceinline__ float3 tex_rgb8(const uchar4 *img, int w, int h, float x, float y)
{
    const TexPos t = tex_locate(w, h, x, y);
    const uchar4 t00 = img[(size_t)t.j0 * w + t.i0], t10 = img[(size_t)t.j0 * w + t.i1];
    const uchar4 t01 = img[(size_t)t.j1 * w + t.i0], t11 = img[(size_t)t.j1 * w + t.i1];
    return make_float3(tex_mix(t, (float)t00.x, (float)t10.x, (float)t01.x, (float)t11.x),
                       tex_mix(t, (float)t00.y, (float)t10.y, (float)t01.y, (float)t11.y),
                       tex_mix(t, (float)t00.z, (float)t10.z, (float)t01.z, (float)t11.z));
}

// one side of render_resample_image: sign = +1 -> kernel_render_resample_image0 (video0, p = q + v,
// q.z -= v.z), -1 -> ...image1 (render.cu:99-199).  Colours are fetched at the pixel itself; the
// field only moves the point where the frame shift v.z is read, and the flow carries the pixel
// between the two frames the shifted time falls between.
__device__ float3 resample_side(int x, int y, int w, int h, int d, int frame, float sign, const float4 *vec,
                                const uchar4 *video, const float2 *flow)
{
    const size_t page = (size_t)w * h;
    const float alpha = 0.5f;
    const float qx = (float)x, qy = (float)y;
    float qz = (float)frame;
    float px = qx, py = qy;
    float4 v = tex4(vec, w, h, (float)(px + 0.5), (float)(py + 0.5));
    for (int i = 0; i < 50; ++i) {
        px = qx + sign * v.x;
        py = qy + sign * v.y;
        const float4 tv = tex4(vec, w, h, (float)(px + 0.5), (float)(py + 0.5));
        v.x = alpha * tv.x + (1 - alpha) * v.x;
        v.y = alpha * tv.y + (1 - alpha) * v.y;
    }
    v = tex4(vec, w, h, (float)(px + 0.5), (float)(py + 0.5));
    qz = qz - sign * v.z;
    if (qz <= 0) return tex_rgb8(video, w, h, (float)(qx + 0.5), (float)(qy + 0.5));
    if (qz >= d - 1) return tex_rgb8(video + (size_t)(d - 1) * page, w, h, (float)(qx + 0.5), (float)(qy + 0.5));
    float pz = floorf(qz);
    const float fa_z = qz - pz;
    int lz = min(max((int)(pz + 0.5), 0), d - 1);
    float2 f = tex2(flow + (size_t)lz * page, w, h, (float)(qx + 0.5), (float)(qy + 0.5));
    for (int i = 0; i < 50; ++i) {
        px = qx - f.x * fa_z;
        py = qy - f.y * fa_z;
        pz = qz - 1.0f * fa_z;
        lz = min(max((int)(pz + 0.5), 0), d - 1);
        const float2 gf = tex2(flow + (size_t)lz * page, w, h, (float)(px + 0.5), (float)(py + 0.5));
        f.x = alpha * gf.x + (1 - alpha) * f.x;
        f.y = alpha * gf.y + (1 - alpha) * f.y;
    }
    const int l0 = min(max((int)(pz + 0.5), 0), d - 1), l1 = min(max((int)(pz + 0.5 + 1), 0), d - 1);
    const float3 c0 = tex_rgb8(video + (size_t)l0 * page, w, h, (float)(px + 0.5), (float)(py + 0.5));
    const float3 c1 = tex_rgb8(video + (size_t)l1 * page, w, h, (float)(px + 0.5 + f.x), (float)(py + 0.5 + f.y));
    return make_float3(c0.x * (1 - fa_z) + c1.x * fa_z, c0.y * (1 - fa_z) + c1.y * fa_z, c0.z * (1 - fa_z) + c1.z * fa_z);
}

// render_resample_image, render.cu:203-246: out = 0; video0's side if fa < 1, video1's if fa > 0;
// each adds (c + 0.5) * weight to the byte already there and truncates
__global__ __launch_bounds__(256) void k_render_resample(uint8_t *out, int out_pitch, int w, int h, int d, float fa, int frame,
                                                         const float4 *vec, const uchar4 *video0, const uchar4 *video1,
                                                         const float2 *forw0, const float2 *forw1)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    uint8_t o[3] = {0, 0, 0};
    if (fa < 1) {
        const float3 c = resample_side(x, y, w, h, d, frame, 1.0f, vec, video0, forw0);
        o[0] = (uint8_t)(o[0] + (c.x + 0.5) * (1 - fa));
        o[1] = (uint8_t)(o[1] + (c.y + 0.5) * (1 - fa));
        o[2] = (uint8_t)(o[2] + (c.z + 0.5) * (1 - fa));
    }
    if (fa > 0) {
        const float3 c = resample_side(x, y, w, h, d, frame, -1.0f, vec, video1, forw1);
        o[0] = (uint8_t)(o[0] + (c.x + 0.5) * fa);
        o[1] = (uint8_t)(o[1] + (c.y + 0.5) * fa);
        o[2] = (uint8_t)(o[2] + (c.z + 0.5) * fa);
    }
    uint8_t *q = out + (size_t)y * out_pitch + 3 * x;
    q[0] = o[0];
    q[1] = o[1];
    q[2] = o[2];
}

} // namespace

void vm_sync_launch_diag(const VmSyncGrid &g, float *diag, float w_tps, hipStream_t s)
{
    const size_t n = (size_t)g.w * g.h * g.d;
    hipLaunchKernelGGL(k_sync_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, diag, w_tps);
}

void vm_sync_launch_scatter(float *dst, const int *idx, const float *val, int n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_sync_scatter, dim3((n + 255) / 256), dim3(256), 0, s, dst, idx, val, n);
}

bool vm_sync_self_mode(const VmSyncGrid &g) { return g.nb <= 512; }

void vm_sync_launch_rr(const VmSyncGrid &g, const VmSyncSys &S, hipStream_t s)
{
    const dim3 grid(8 * g.per_xcd, 3);
    if (vm_sync_self_mode(g))
        hipLaunchKernelGGL((k_sync_B<true, true>), grid, dim3(256), 0, s, S, g, 0);
    else
        hipLaunchKernelGGL((k_sync_B<true, false>), grid, dim3(256), 0, s, S, g, 0);
}

void vm_sync_launch_iteration(const VmSyncGrid &g, const VmSyncSys &S, int k, hipStream_t s)
{
    const dim3 grid(8 * g.per_xcd, 3);
    if (vm_sync_self_mode(g)) {
        if (k == 1)
            hipLaunchKernelGGL((k_sync_A<true, true, true>), grid, dim3(256), 0, s, S, g, k);
        else
            hipLaunchKernelGGL((k_sync_A<false, true, true>), grid, dim3(256), 0, s, S, g, k);
        hipLaunchKernelGGL((k_sync_B<false, true>), grid, dim3(256), 0, s, S, g, k);
    } else {
        const bool small = g.nb * 3 <= 1024;
        if (k == 1) {
            if (small)
                hipLaunchKernelGGL((k_sync_A<true, true, false>), grid, dim3(256), 0, s, S, g, k);
            else
                hipLaunchKernelGGL((k_sync_A<true, false, false>), grid, dim3(256), 0, s, S, g, k);
        } else {
            if (small)
                hipLaunchKernelGGL((k_sync_A<false, true, false>), grid, dim3(256), 0, s, S, g, k);
            else
                hipLaunchKernelGGL((k_sync_A<false, false, false>), grid, dim3(256), 0, s, S, g, k);
        }
        hipLaunchKernelGGL((k_sync_B<false, false>), grid, dim3(256), 0, s, S, g, k);
    }
}

// after the loop: make r1 of iteration k readable in the scalar block (SELF mode keeps it as partials)
void vm_sync_launch_finish(const VmSyncGrid &g, const VmSyncSys &S, int k, hipStream_t s)
{
    if (vm_sync_self_mode(g)) hipLaunchKernelGGL(k_sync_total, dim3(3), dim3(256), 0, s, S, g.nb, k & 1);
}

void vm_sync_launch_upsample(float *dst, int dw, int dh, const float *src, int sw, int sh, float ratio, int pages,
                             hipStream_t s)
{
    hipLaunchKernelGGL(k_sync_upsample, dim3((dw + 31) / 32, (dh + 7) / 8, pages), dim3(32, 8), 0, s, dst, dw, dh, src, sw, sh,
                       ratio);
}

void vm_sync_launch_result(const float *X, const float *Y, const float *Z, int w, int h, int w0, int h0, float4 *out,
                           hipStream_t s)
{
    const float ratio_x = (float)w0 / (float)w, ratio_y = (float)h0 / (float)h;
    const double sx = 1.0 / ((double)w0 / w), sy = 1.0 / ((double)h0 / h);
    hipLaunchKernelGGL(k_sync_result, dim3((w0 + 31) / 32, (h0 + 7) / 8), dim3(32, 8), 0, s, X, Y, Z, w, h, w0, h0, ratio_x,
                       ratio_y, sx, sy, out);
}

void vm_sync_launch_render(uint8_t *out, int out_pitch, int w, int h, int d, float fa, int frame, const float4 *vec,
                           const uchar4 *video0, const uchar4 *video1, const float2 *forw0, const float2 *forw1,
                           hipStream_t s)
{
    hipLaunchKernelGGL(k_render_resample, dim3((w + 31) / 32, (h + 7) / 8), dim3(32, 8), 0, s, out, out_pitch, w, h, d, fa,
                       frame, vec, video0, video1, forw0, forw1);
}
