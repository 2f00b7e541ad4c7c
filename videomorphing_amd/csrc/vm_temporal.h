// vm_temporal.h -- launchers of vm_temporal.hip (temporal coherence path + flow pyramid)
#ifndef VM_TEMPORAL_H
#define VM_TEMPORAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

void vm_temp_launch_splat(int w, int h, int rs, const float2 *v_prev, const float2 *f0, const float2 *f1,
                          const float *ssim, long long *acc, hipStream_t s);
void vm_temp_launch_finish(int w, int h, int rs, const long long *acc, float2 *ref_out, float *mask_out, int init_temp,
                           hipStream_t s);
void vm_temp_launch_smooth(int w, int h, int rs, float2 *v_out, const float2 *v_cur, const float *weight, hipStream_t s);
void vm_temp_launch_fill_zeros_x(int w, int h, int rs, float2 *v_out, const float *weight, hipStream_t s);
void vm_flow_launch_load(const float2 *flow, int pitch, float *img, int w, int h, hipStream_t s);
void vm_flow_launch_store(const float *img, float2 *flow, int pitch, int w, int h, float ratiox, float ratioy,
                          hipStream_t s);
void vm_flow_launch_concat(float2 *f, const float2 *f_next, int pitch, int w, int h, hipStream_t s);
#endif
