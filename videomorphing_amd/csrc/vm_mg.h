// vm_mg.h -- multigrid-preconditioned conjugate gradients for the screened 5-point systems of
// the compositor (Poisson boundary extension, PoissonExt.cpp:214-329; quadratic motion path,
// QuadraticPath.cpp:111-215).  Launchers of vm_mg.hip.
#ifndef VM_MG_H
#define VM_MG_H

#include "vm_internal.h"

// One grid of the hierarchy.  The operator is a weighted graph Laplacian plus screening:
//   (A u)(p) = dg(p) u(p) - we(p) u(p + x) - we(p - x) u(p - x) - ws(p) u(p + y) - ws(p - y) u(p - y)
// we/ws = weight of the edge to the east / south neighbour (0: no edge), dg = screening + the
// incident weights, dg == 0: p is not an unknown.  Vectors carry three channels in a float4.
struct VmMgLevel {
    int w, h;
    float *we, *ws, *dg;
    float4 *x, *b, *t;
};

// scalars of the PCG iteration, resident in device memory (one set per channel)
struct VmPcgScalars {
    double rz[3], rz_new[3], pq[3], rr[3], bb[3];
};

// level 0 from the type map of the Poisson extension (unknown: type > 0, screening 1 on type 1)
void vm_mg_launch_level0_type(const uint8_t *type, const VmMgLevel &L, hipStream_t s);
// level 0 of a full grid without screening (pure Neumann: singular, constants in the null space)
void vm_mg_launch_level0_full(const VmMgLevel &L, hipStream_t s);
// Galerkin coarsening over 2x2 blocks with piecewise-constant interpolation, the edge weights
// rescaled by 1/2 (the energy of a linear profile across a block boundary)
void vm_mg_launch_coarsen(const VmMgLevel &F, const VmMgLevel &C, hipStream_t s);
// x = omega b / dg (damped Jacobi from a zero guess)
void vm_mg_launch_jacobi0(const VmMgLevel &L, float omega, hipStream_t s);
// C.b = P^T (F.b - A F.x)
void vm_mg_launch_resid_restrict(const VmMgLevel &F, const VmMgLevel &C, hipStream_t s);
// F.t = x1 + omega (F.b - A x1) / dg,  x1 = F.x + P C.x   (coarse correction + post-smoothing)
void vm_mg_launch_prolong_smooth(const VmMgLevel &F, const VmMgLevel &C, float omega, hipStream_t s);
// coarsest grid (w h <= 1024): `sweeps` damped-Jacobi sweeps from zero inside one workgroup
void vm_mg_launch_coarsest(const VmMgLevel &L, float omega, int sweeps, hipStream_t s);
// the two coarsest grids of a cycle in one workgroup (F.w F.h <= 4096, C the coarsest): F's
// pre-smoothing, restriction, C's sweeps, correction + post-smoothing; the result ends in F.t
void vm_mg_launch_coarse_tail(const VmMgLevel &F, const VmMgLevel &C, float omega, int sweeps, hipStream_t s);

// PCG on level 0 (vectors: X solution, B right-hand side, R residual, P direction, Q = A P)
void vm_mg_launch_pcg_init(const VmMgLevel &L, const float4 *B, const float4 *X, float4 *R, VmPcgScalars *sc,
                           hipStream_t s);
void vm_mg_launch_pcg_spmv(const VmMgLevel &L, const float4 *P, float4 *Q, VmPcgScalars *sc, hipStream_t s);
void vm_mg_launch_pcg_update(const VmMgLevel &L, float4 *X, float4 *R, const float4 *P, const float4 *Q,
                             VmPcgScalars *sc, hipStream_t s);
void vm_mg_launch_pcg_dot(const VmMgLevel &L, const float4 *R, const float4 *Z, VmPcgScalars *sc, hipStream_t s);
void vm_mg_launch_pcg_dir(const VmMgLevel &L, float4 *P, const float4 *Z, VmPcgScalars *sc, int first,
                          hipStream_t s);

#endif
