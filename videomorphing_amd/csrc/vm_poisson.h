// vm_poisson.h -- launchers of vm_poisson.hip (Poisson boundary extension).
#ifndef VM_POISSON_H
#define VM_POISSON_H

#include "vm_internal.h"

void vm_poisson_launch_crop(uchar4 *dst, const uchar4 *ext, int w, int h, int ex, hipStream_t s);
// the extended canvas of Pyramid::build (pyramid.cu:186-200) from a tight RGB8 frame: (255, 255, 255, 255) around it, the
// frame at (ex, ex) with alpha 0; and the frame's RGBA copy (the crop the other side's fill samples)
void vm_poisson_launch_canvas(uchar4 *ext, uchar4 *crop, const uint8_t *rgb, int w, int h, int ex, hipStream_t s);
void vm_poisson_launch_prepare(uchar4 *ext, uint8_t *type, const uchar4 *other, const float2 *v,
                               int w, int h, int rs, int ex, int sign, hipStream_t s);
// right-hand side + initial guess, and the paste of the solution, on the solver's 12-byte vectors (vm_mgb.h)
struct VmV3;
void vm_poisson_launch_setup3(const uchar4 *ext, const uint8_t *type, VmV3 *B, VmV3 *X, int cw, int ch, hipStream_t s);
void vm_poisson_launch_paste3(uchar4 *ext, const uint8_t *type, const VmV3 *X, int cw, int ch, hipStream_t s);

// quadratic motion path (QuadraticPath.cpp:24-223)
// on the solver's 12-byte vectors (vm_mgb.h); sums: VM_QP_SLOTS lines of 16 doubles
#define VM_QP_SLOTS 8
void vm_qpath_launch_rhs3(const float2 *v, int rs, int w, int h, VmV3 *B, VmV3 *X, hipStream_t s);
void vm_qpath_launch_sum3(const VmV3 *X, int w, int h, double *sums, hipStream_t s);
void vm_qpath_launch_shift3(VmV3 *X, int w, int h, const double *sums, float2 *u, int rs, hipStream_t s);

#endif
