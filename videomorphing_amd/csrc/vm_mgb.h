// vm_mgb.h -- the compositor's linear solver: multigrid-preconditioned CG, batched over systems, ring-only, fused
// (launchers of vm_mgb.hip).  The system is PoissonExt.cpp:214-312's screened 5-point operator (MKL DSS is the
// reference's solver, :321-329) or the quadratic path's Neumann Laplacian (QuadraticPath.cpp:111-215); one V
// cycle of a 2x2-aggregation multigrid per PCG iteration:
//   * a SYSTEM is one side of one frame; blockIdx.z = system, so both sides of a frame (CPoissonExt::run's two
//     prepare / poissonExtend pairs, PoissonExt.cpp:29-35, which do not depend on each other) -- and several frames --
//     share every launch: the coarse grids of the cycle are launch-bound, not byte-bound;
//   * every grid of the hierarchy is swept over the cells that hold an unknown only (compact lists of 64x4-cell
//     blocks for the streaming PCG kernels and of 64x16-cell tiles for the cycle, built once per extension): the
//     unknowns are the ring around the original image, 39 % of the canvas of a 1080p frame with ex = 192;
//   * the smoother is RED-BLACK GAUSS-SEIDEL, symmetric (red, black before the coarse correction; black, red after
//     it: the preconditioner stays symmetric positive definite), computed inside the restriction and prolongation
//     kernels from an LDS tile of the right-hand side with a two-cell apron: the pre-smoothed iterate is never
//     stored, its residual is zero on black cells and a sum over four black neighbours on red ones, and the
//     post-smoothing needs the red cells' values only.  Same bytes per cycle as one damped-Jacobi sweep each way
//     (rounds 4-5), 11 instead of 17 PCG iterations to 1e-5 on the 2304x1464 canvas; from level 2 down the cycle makes
//     TWO sweeps each way (a four-cell apron, the recurrences written in differences: vm_mgb.hip): 9 iterations;
//   * r.z rides in the level-0 prolongation, p = z + beta p in the operator application (p ping-pongs), the scalar
//     bookkeeping lives in accumulators indexed by the iteration's parity, cleared by the kernel that provably runs
//     between their last reader and next writer;
//   * all grids of <= ~6 k cells together (the last four of nine for a 1080p canvas, down to <= 64 cells) are one
//     workgroup's work in LDS: 13 launches per PCG iteration;
//   * the PCG update (x += alpha p, r -= alpha q, r.r) rides in front of the level-0 restriction of the next cycle, whose
//     loads wait for latency while k_mgb_update streamed at the HBM ceiling: 12 launches, the residual read once (it
//     ping-pongs between two buffers: a tile's apron is another tile's interior).  k_mgb_update by itself remains for
//     hierarchies whose level 0 sits inside the tail or makes two sweeps.
#ifndef VM_MGB_H
#define VM_MGB_H

#include "vm_internal.h"

#define VM_MGB_MAXLEV 14
#define VM_MGB_SLOTS 8          // dot-product accumulators are spread over 8 lines: same-address double atomics serialise in the L2
#define VM_MGB_MAXSYS 64

// a vector entry in memory: three colour channels, 12 bytes (dwordx3 loads / stores: a quarter less traffic than float4)
struct VmV3 {
    float x, y, z;
};

// One grid of one system's hierarchy:
//   (A u)(p) = dg(p) u(p) - we(p) u(p + x) - we(p - x) u(p - x) - ws(p) u(p + y) - ws(p - y) u(p - y)
// dg == 0: p is not an unknown.  b = the level's right-hand side (level 0: the PCG residual r), x = the result
// of the level's cycle (level 0: z = M^-1 r).
struct VmMgbLevel {
    int w, h;
    int gx, gy;              // blocks of 64 x 4 cells covering the grid
    // the operator.  Level 0 (weights 0 / 1, diagonal 0 .. 5): ONE byte per cell, info = dg << 4 | N << 3 | S << 2 |
    // W << 1 | E (edge present towards that neighbour) -- 1 byte of operator per cell and kernel instead of 16;
    // coarser levels: the edge weights to the east / south neighbour, the diagonal, and k = 1 / dg (0: no
    // unknown), which is what the neighbours of a cell are needed for
    uint8_t *info;
    float *we, *ws, *dg, *k;
    VmV3 *b, *x;
    // smoothing sweeps each way on this level (1 or 2).  With 2 the restriction kernel leaves the pre-smoothed iterate of
    // the RED cells in xr, packed: entry (y w + x) >> 1 (the post-smoothing starts from it; black cells are recomputed)
    int nu;
    VmV3 *xr;
    uint32_t *flags;         // per block: does it hold an unknown (set-up scratch)
    uint32_t *blocks;        // the blocks that do, packed bx | by << 16, row-major
    int *nblocks;            // their number (device)
    uint32_t *tiles;         // the 64 x 16-cell tiles (four blocks of a column) that hold an unknown, packed bx | ty << 16
    int *ntiles;
};

struct VmMgbScalars {
    double bb[VM_MGB_SLOTS][16];        // [slot][channel], one 128-byte line per slot
    double rr[2][VM_MGB_SLOTS][16];     // [iteration parity] ...
    double rz[2][VM_MGB_SLOTS][16];
    double pq[2][VM_MGB_SLOTS][16];
};

struct VmMgbSys {
    int nlev;
    VmMgbLevel lv[VM_MGB_MAXLEV];
    VmV3 *X, *P[2], *Q;
    // the PCG residual of iteration k: R[k & 1].  R[0] == lv[0].b always; R[1] is a buffer of its own when the update rides in
    // the level-0 restriction (vm_mgb.hip: k_mgb_restrict<true, true>), else R[0] again (k_mgb_update works in place)
    VmV3 *R[2];
    const uint8_t *type;     // level 0's type map (PoissonExt.cpp:59-101)
    VmMgbScalars *sc;
};

#define VM_MGB_COARSEST 64      // the hierarchy ends at a grid of at most this many cells ...
// Red-black sweeps each way per level, from level 0 on (comma list, the last entry repeats; vm_poisson_api.cpp: mg_nu).
// Measured on the 2304 x 1464 canvas, tol 1e-5 / 1e-6 (tools/exp/nu_sweep.sh, nu_ab.sh on one box; tools/exp/mg_prototype.py
// is the CPU model that predicted the iteration counts), ms per frame in 4-frame batches in the bench line's setting:
//   1 everywhere      11 / 13 iterations   2.20 / 2.52
//   1, 1, 2           9 / 10               2.00 / 2.17   <- the extra sweeps go where the cycle is launch-bound, not byte-bound
//   2 everywhere      7-8 / 8              as slow as 1 everywhere: level 0's wider window costs what the iterations save
// The quadratic path's whole-grid system (1920 x 1080, tol 1e-4, a solved field) stays at 1 everywhere: 1.72 ms (8 iterations)
// against 2.02 (8) with 1, 1, 2 and 1.89 (6) with 2 everywhere.
#ifndef VM_MGB_NU_POISSON
#define VM_MGB_NU_POISSON 1, 1, 2
#define VM_MGB_NU_QPATH 1
#endif
#define VM_MGB_COARSE_SWEEPS 2  // ... which gets this many symmetric Gauss-Seidel sweeps each way (R B R B, B R B R) from zero
// the tail of the cycle -- every level from `tail` on -- runs in ONE workgroup with the iterates in LDS: the levels'
// cell counts must fit these pools: all of them (a float4 iterate + a float2 of edge weights per cell) / all but the first
// (a float4 right-hand side): 120 + 32 KB of the CU's 160 KB of LDS
#ifndef VM_MGB_TAIL_X
#define VM_MGB_TAIL_X 5120
#define VM_MGB_TAIL_B 2048
#endif
// ... and no level of the tail may hold more than this many PAIRS of cells, ceil(w / 2) h (the tail's threads are dealt
// pairs, three each)
#define VM_MGB_TAIL_PAIRS 3072

// set-up: level 0 from the type map, Galerkin coarsening (2x2 aggregates, edge weights x 1/2), block flags on the way
void vm_mgb_launch_level0(const VmMgbSys *sys, int nsys, int gx, int gy, hipStream_t s);
void vm_mgb_launch_coarsen(const VmMgbSys *sys, int nsys, int l, int gx, int gy, hipStream_t s);   // level l from l - 1
void vm_mgb_launch_compact(const VmMgbSys *sys, int nsys, int nlev_max, hipStream_t s);
// r = b - A x (in place, level 0's b), bb, rr[1]
void vm_mgb_launch_init(const VmMgbSys *sys, int nsys, int nb0, uint64_t active, hipStream_t s);
// V-cycle pieces
// (nu: the level's sweeps each way, VmMgbLevel::nu of every system of the batch)
// lv[l+1].b from lv[l], over lv[l]'s tiles; k: the PCG iteration (level 0 reads R[k & 1]); upd (l == 0, nu == 1, k >= 1): the
// update of iteration k - 1 first (x += alpha p, R[k & 1] = R[(k - 1) & 1] - alpha q, rr[(k - 1) & 1]) -- instead of vm_mgb_launch_update
void vm_mgb_launch_restrict(const VmMgbSys *sys, int nsys, int l, int nu, int nt_fine, int k, bool upd, uint64_t active, hipStream_t s);
void vm_mgb_launch_prolong(const VmMgbSys *sys, int nsys, int l, int nu, int nt_fine, int k, uint64_t active, hipStream_t s);   // lv[l].x; l == 0: rz[k & 1] += r.z
void vm_mgb_launch_tail(const VmMgbSys *sys, int nsys, int l, uint64_t active, hipStream_t s);             // levels l .. nlev - 1 in one workgroup
void vm_mgb_launch_dot_rz(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s);                           // hierarchies that are all tail only
// PCG on level 0
void vm_mgb_launch_dirspmv(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s);   // p = z + beta p, q = A p, pq[k & 1]
void vm_mgb_launch_update(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s);    // x += alpha p, r -= alpha q, rr[k & 1]

#endif
