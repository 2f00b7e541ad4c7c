// dev: GPU-side cost of a dependent chain of early-exit kernels as a function of footprint
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
template <int LDS, int T>
__global__ __launch_bounds__(T) void k_exit(const unsigned *mask, unsigned *out)
{
    __shared__ unsigned sh[LDS / 4 > 0 ? LDS / 4 : 1];
    unsigned m = 0;
    if (threadIdx.x < 96) m = mask[(blockIdx.x * 96 + threadIdx.x) & 0xFFFF];
    if (LDS > 0) sh[threadIdx.x] = m;
    if (!__syncthreads_or(m != 0)) return;
    out[blockIdx.x] = sh[0];
}
template <int LDS, int T>
void run(const char *name, int grid, unsigned *mask, unsigned *out, hipStream_t s)
{
    for (int rep = 0; rep < 2; ++rep) {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL((k_exit<LDS, T>), dim3(grid), dim3(T), 0, s, mask, out);
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
        if (rep) printf("%-28s grid %5d: %.2f us per launch\n", name, grid, us);
    }
}
int main()
{
    unsigned *mask, *out;
    hipMalloc(&mask, 65536 * 4); hipMemset(mask, 0, 65536 * 4);
    hipMalloc(&out, 65536 * 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int grid : {91, 375, 1456}) {
        run<0, 256>("no LDS, 256 thr", grid, mask, out, s);
        run<0, 512>("no LDS, 512 thr", grid, mask, out, s);
        run<16384, 512>("16 KB LDS, 512 thr", grid, mask, out, s);
        run<59392, 512>("58 KB LDS, 512 thr", grid, mask, out, s);
    }
    return 0;
}
