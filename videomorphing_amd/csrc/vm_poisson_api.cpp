// vm_poisson_api.cpp -- C-ABI of the Poisson boundary extension
// (CPoissonExt::run body for one side, Algorithm/PoissonExt.cpp:19-41) and the
// RCCL broadcast helper.
#include "vm_host.h"
#include "vm_poisson.h"

#include <cmath>
#include <dlfcn.h>

namespace {
struct PoissonWs {
    uint8_t *type;
    float4 *B, *X, *R, *P, *Q;
    VmCgScalars *sc;
};

int ws_get(vm_frame *f, PoissonWs &ws)
{
    const size_t N = (size_t)f->cw * f->ch;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t need = al(N) + 5 * al(N * 16) + al(sizeof(VmCgScalars));
    if (f->pws_bytes < need) {
        hipFree(f->pws);
        f->pws = nullptr;
        f->pws_bytes = 0;
        VM_HIP(hipMalloc(&f->pws, need));
        f->pws_bytes = need;
    }
    char *b = (char *)f->pws;
    ws.type = (uint8_t *)b; b += al(N);
    ws.B = (float4 *)b; b += al(N * 16);
    ws.X = (float4 *)b; b += al(N * 16);
    ws.R = (float4 *)b; b += al(N * 16);
    ws.P = (float4 *)b; b += al(N * 16);
    ws.Q = (float4 *)b; b += al(N * 16);
    ws.sc = (VmCgScalars *)b;
    return VM_OK;
}
} // namespace

extern "C" int vm_poisson_extend(vm_frame *f, int side, float tol, int max_it, int *iters,
                                 float *rel_res, float *elapsed_ms)
{
    if (!f || (side != 1 && side != 2) || !(tol > 0) || max_it < 1)
        return vm_fail(VM_E_INVALID, "vm_poisson_extend: bad argument");
    vm_ctx *c = f->ctx;
    hipStream_t s = c->stream;
    PoissonWs ws;
    int rc = ws_get(f, ws);
    if (rc != VM_OK) return rc;
    uchar4 *ext = f->ext[side - 1];
    const uchar4 *other = f->crop[side == 1 ? 1 : 0]; // PoissonExt.cpp:54-57
    const int sign = side == 1 ? 1 : -1;
    VM_HIP(hipEventRecord(c->ev0, s));
    vm_poisson_launch_prepare(ext, ws.type, other, f->v, f->w, f->h, f->rs, f->ex, sign, s);
    VM_HIP(hipMemsetAsync(ws.sc, 0, sizeof(VmCgScalars), s));
    vm_poisson_launch_setup(ext, ws.type, ws.B, ws.X, ws.R, ws.P, ws.sc, f->cw, f->ch, s);
    VM_HIP(hipGetLastError());
    VmCgScalars h;
    int it = 0;
    double worst = 0;
    const int check = 32;
    bool converged = false;
    while (true) {
        VM_HIP(hipMemcpyAsync(&h, ws.sc, sizeof(h), hipMemcpyDeviceToHost, s));
        VM_HIP(hipStreamSynchronize(s));
        worst = 0;
        for (int k = 0; k < 3; ++k)
            if (h.bb[k] > 0) worst = std::max(worst, std::sqrt(h.rr[k] / h.bb[k]));
        if (!(worst == worst)) return vm_fail(VM_E_NUMERIC, "vm_poisson_extend: CG broke down (NaN)");
        if (worst <= tol) { converged = true; break; }
        if (it >= max_it) break;
        const int nb = std::min(check, max_it - it);
        for (int k = 0; k < nb; ++k)
            vm_poisson_launch_iter(ws.X, ws.R, ws.P, ws.Q, ws.B, ws.type, ws.sc, f->cw, f->ch, s);
        VM_HIP(hipGetLastError());
        it += nb;
    }
    vm_poisson_launch_paste(ext, ws.type, ws.X, f->cw, f->ch, s);
    VM_HIP(hipGetLastError());
    VM_HIP(hipEventRecord(c->ev1, s));
    VM_HIP(hipEventSynchronize(c->ev1));
    float ms = 0;
    VM_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (iters) *iters = it;
    if (rel_res) *rel_res = (float)worst;
    if (elapsed_ms) *elapsed_ms = ms;
    if (!converged)
        return vm_fail(VM_E_NUMERIC, "vm_poisson_extend: residual %.3g after %d iterations (tol %.3g)", worst, it, (double)tol);
    return VM_OK;
}

// RCCL is resolved at first use so that the library loads (and the CPU-side
// tests run) on hosts without a usable librccl.
extern "C" int vm_rccl_bcast(vm_ctx *c, void *comm, void *dev_buf, uint64_t bytes, int root)
{
    if (!c || !comm || !dev_buf) return vm_fail(VM_E_INVALID, "vm_rccl_bcast: NULL argument");
    typedef int (*bcast_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    static bcast_fn fn = nullptr;
    if (!fn) {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return vm_fail(VM_E_DEVICE, "vm_rccl_bcast: cannot load librccl: %s", dlerror());
        fn = (bcast_fn)dlsym(h, "ncclBroadcast");
        if (!fn) return vm_fail(VM_E_DEVICE, "vm_rccl_bcast: ncclBroadcast not found");
    }
    const int ncclInt8 = 0; // ncclDataType_t: ncclInt8 / ncclChar
    int rc = fn(dev_buf, dev_buf, (size_t)bytes, ncclInt8, root, comm, c->stream);
    if (rc != 0) return vm_fail(VM_E_DEVICE, "vm_rccl_bcast: ncclBroadcast returned %d", rc);
    return VM_OK;
}
