// vm_pyramid.h -- launchers of vm_pyramid.hip (device-side luma pyramid builder).
#ifndef VM_PYRAMID_H
#define VM_PYRAMID_H
#include "vm_internal.h"
void vm_pyr_launch_load(const uint8_t *rgb, int pitch, float *img, int w, int h, hipStream_t s);
void vm_pyr_launch_curve(float *img, size_t n, int to_gamma, hipStream_t s);
void vm_pyr_launch_down(const float *src, float *dst, int win, int hin, int nout, int axis, hipStream_t s);
void vm_pyr_launch_up(const float *src, float *dst, int win, int hin, int nout, int axis, hipStream_t s);
void vm_pyr_launch_tri_solve(float *img, const float *A, int w, int h, int axis, hipStream_t s);
void vm_pyr_launch_store_gray(const float *img, float *luma, int w, int h, int rs, hipStream_t s);
#endif
