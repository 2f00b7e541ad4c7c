// experiment: do plain loads in a LATER kernel see data an earlier kernel stored write-through (sc1)?
// A: every workgroup plain-loads X (warms all eight L2s); B: one subset of workgroups rewrites X with
// sc1 stores (or plain stores); C: every workgroup plain-loads (or sc1-loads) X and counts stale words.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const uint32_t g_cu32;
typedef __attribute__((address_space(1))) uint32_t g_u32;
__global__ void k_read(const uint32_t *x, int n, uint32_t expect, int coh, uint32_t *stale, uint32_t *sink)
{
    uint32_t s = 0, bad = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t v = coh ? __hip_atomic_load((g_cu32 *)(x + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : x[i];
        s += v;
        bad += v != expect;
    }
    if (bad) atomicAdd(stale, bad);
    if (s == 0x12345678u) *sink = s;
}
__global__ void k_write(uint32_t *x, int n, uint32_t val, int coh)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (coh) __hip_atomic_store((g_u32 *)(x + i), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else x[i] = val;
    }
}
int main()
{
    const int n = 16384;
    uint32_t *x, *stale, *sink;
    hipMalloc(&x, n * 4); hipMalloc(&stale, 4); hipMalloc(&sink, 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int wcoh = 0; wcoh < 2; ++wcoh)
        for (int rcoh = 0; rcoh < 2; ++rcoh)
            for (int wblocks = 1; wblocks <= 256; wblocks *= 16) {
                uint32_t tot = 0;
                for (int rep = 0; rep < 50; ++rep) {
                    const uint32_t v0 = 1000 + rep * 2, v1 = v0 + 1;
                    hipMemsetAsync(stale, 0, 4, s);
                    hipLaunchKernelGGL(k_write, dim3(64), dim3(256), 0, s, x, n, v0, 0);
                    hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, s, x, n, v0, 0, sink, sink); // warm (stale count discarded)
                    hipLaunchKernelGGL(k_write, dim3(wblocks), dim3(256), 0, s, x, n, v1, wcoh);
                    hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, s, x, n, v1, rcoh, stale, sink);
                    uint32_t h = 0;
                    hipMemcpyAsync(&h, stale, 4, hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                    tot += h;
                }
                printf("store %s (%3d blocks), later-kernel load %s: stale words over 50 reps x 256 readers x %d = %u\n", wcoh ? "sc1  " : "plain", wblocks,
                       rcoh ? "sc1  " : "plain", n, tot);
            }
    return 0;
}
