// vm_mgb.h -- the Poisson extension's multigrid-preconditioned CG, batched over systems, ring-only, fused
// (round 5; launchers of vm_mgb.hip).  Same mathematics as vm_mg.hip (one V(1,1) cycle of the 2x2-aggregation
// multigrid per PCG iteration, damped Jacobi 0.8, 40 sweeps on the coarsest grid; PoissonExt.cpp:214-329 is the
// system, MKL DSS the reference's solver), restructured around what the counters of round 4's form showed:
//   * a SYSTEM is one side of one frame; blockIdx.z = system, so both sides of a frame (CPoissonExt::run's two
//     prepare / poissonExtend pairs, PoissonExt.cpp:29-35, which do not depend on each other) -- and several frames --
//     share every launch: the coarse grids of the cycle are launch-bound, not byte-bound;
//   * every grid of the hierarchy is swept over the 64x4-cell blocks that hold an unknown only (a compact block
//     list per level, built once per extension): the unknowns are the ring around the original image, 39 % of the
//     canvas of a 1080p frame with ex = 192;
//   * the pre-smoothed iterate x = omega b / dg is never stored (restriction and prolongation recompute it from b
//     and dg at the five points they touch), r.z rides in the level-0 prolongation, p = z + beta p in the
//     operator application (p ping-pongs), the scalar bookkeeping kernels are gone (accumulators indexed by the
//     iteration's parity, cleared by the kernel that provably runs between their last reader and next writer):
//     13 launches per PCG iteration instead of 22, ~250 instead of ~370 bytes per unknown.
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
    // coarser levels: the edge weights to the east / south neighbour, the diagonal, and k = omega / dg (0: no
    // unknown), which is what the neighbours of a cell are needed for
    uint8_t *info;
    float *we, *ws, *dg, *k;
    VmV3 *b, *x;
    uint32_t *flags;         // per block: does it hold an unknown (set-up scratch)
    uint32_t *blocks;        // the blocks that do, packed bx | by << 16, row-major
    int *nblocks;            // their number (device)
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
    const uint8_t *type;     // level 0's type map (PoissonExt.cpp:59-101)
    VmMgbScalars *sc;
};

#define VM_MGB_OMEGA 0.8f       // damped Jacobi (a compile-time constant of the kernels: omega / dg of level 0 is a 5-entry table)

// set-up: level 0 from the type map, Galerkin coarsening (2x2 aggregates, edge weights x 1/2), block flags on the way
void vm_mgb_launch_level0(const VmMgbSys *sys, int nsys, int gx, int gy, hipStream_t s);
void vm_mgb_launch_coarsen(const VmMgbSys *sys, int nsys, int l, int gx, int gy, hipStream_t s);   // level l from l - 1
void vm_mgb_launch_compact(const VmMgbSys *sys, int nsys, int nlev_max, hipStream_t s);
// r = b - A x (in place, level 0's b), bb, rr[1]
void vm_mgb_launch_init(const VmMgbSys *sys, int nsys, int nb0, uint64_t active, hipStream_t s);
// V-cycle pieces
void vm_mgb_launch_restrict(const VmMgbSys *sys, int nsys, int l, int nb_fine, uint64_t active, hipStream_t s);   // lv[l+1].b from lv[l], over lv[l]'s blocks
void vm_mgb_launch_prolong(const VmMgbSys *sys, int nsys, int l, int nb_fine, int k, uint64_t active, hipStream_t s);   // lv[l].x; l == 0: rz[k & 1] += r.z
void vm_mgb_launch_tail(const VmMgbSys *sys, int nsys, int l, int sweeps, uint64_t active, hipStream_t s);             // levels l, l + 1 (the coarsest) in one workgroup
void vm_mgb_launch_coarsest(const VmMgbSys *sys, int nsys, int l, int sweeps, uint64_t active, hipStream_t s);
void vm_mgb_launch_dot_rz(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s);                           // hierarchies of <= 2 levels only
// PCG on level 0
void vm_mgb_launch_dirspmv(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s);   // p = z + beta p, q = A p, pq[k & 1]
void vm_mgb_launch_update(const VmMgbSys *sys, int nsys, int nb0, int k, uint64_t active, hipStream_t s);    // x += alpha p, r -= alpha q, rr[k & 1]

#endif
