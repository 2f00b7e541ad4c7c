// vm_poisson.h -- launchers of vm_poisson.hip (Poisson boundary extension).
#ifndef VM_POISSON_H
#define VM_POISSON_H

#include "vm_internal.h"

// CG scalars, resident in device memory (one set per colour channel)
struct VmCgScalars {
    double rz[3], rz_new[3], pq[3], rr[3], rr_new[3], bb[3];
    int iters;
    int pad;
};

void vm_poisson_launch_crop(uchar4 *dst, const uchar4 *ext, int w, int h, int ex, hipStream_t s);
void vm_poisson_launch_prepare(uchar4 *ext, uint8_t *type, const uchar4 *other, const float2 *v,
                               int w, int h, int rs, int ex, int sign, hipStream_t s);
void vm_poisson_launch_setup(const uchar4 *ext, const uint8_t *type, float4 *B, float4 *X, int cw, int ch,
                             hipStream_t s);
void vm_poisson_launch_cg_init(const float4 *B, const float4 *X, float4 *R, float4 *P, const uint8_t *type,
                               VmCgScalars *sc, int cw, int ch, hipStream_t s);
void vm_poisson_launch_coarsen(const uchar4 *ext, const uint8_t *type, uchar4 *ext_c, uint8_t *type_c, int cw,
                               int ch, int cw2, int ch2, hipStream_t s);
void vm_poisson_launch_prolong(const float4 *Xc, const uint8_t *type_c, float4 *X, const uint8_t *type, int cw,
                               int ch, int cw2, int ch2, hipStream_t s);
void vm_poisson_launch_iter(float4 *X, float4 *R, float4 *P, float4 *Q, const float4 *B,
                            const uint8_t *type, VmCgScalars *sc, int cw, int ch, hipStream_t s);
void vm_poisson_launch_paste(uchar4 *ext, const uint8_t *type, const float4 *X, int cw, int ch,
                             hipStream_t s);

// the same two for the batched solver's 12-byte vectors (vm_mgb.h)
struct VmV3;
void vm_poisson_launch_setup3(const uchar4 *ext, const uint8_t *type, VmV3 *B, VmV3 *X, int cw, int ch, hipStream_t s);
void vm_poisson_launch_paste3(uchar4 *ext, const uint8_t *type, const VmV3 *X, int cw, int ch, hipStream_t s);

// quadratic motion path (QuadraticPath.cpp:24-223)
void vm_qpath_launch_rhs(const float2 *v, int rs, int w, int h, float4 *B, float4 *X, hipStream_t s);
void vm_qpath_launch_sum(const float4 *X, int w, int h, double *sums, hipStream_t s);
void vm_qpath_launch_shift(float4 *X, int w, int h, const double *sums, float2 *u, int rs, hipStream_t s);
// the same on 12-byte vectors (the batched solver's layout, vm_mgb.h); sums: VM_QP_SLOTS lines of 16 doubles
#define VM_QP_SLOTS 8
void vm_qpath_launch_rhs3(const float2 *v, int rs, int w, int h, VmV3 *B, VmV3 *X, hipStream_t s);
void vm_qpath_launch_sum3(const VmV3 *X, int w, int h, double *sums, hipStream_t s);
void vm_qpath_launch_shift3(VmV3 *X, int w, int h, const double *sums, float2 *u, int rs, hipStream_t s);

#endif
