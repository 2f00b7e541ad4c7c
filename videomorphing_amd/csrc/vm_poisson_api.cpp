// vm_poisson_api.cpp -- C-ABI of the Poisson boundary extension
// (CPoissonExt::run body for one side, Algorithm/PoissonExt.cpp:19-41) and the
// RCCL broadcast helper.
#include "vm_host.h"
#include "vm_poisson.h"

#include <cmath>
#include <dlfcn.h>

namespace {

// one grid of the nested iteration (level 0 = the frame's canvas)
struct Grid {
    int cw, ch;
    uchar4 *ext;     // level 0: the frame's canvas; coarser: in the workspace
    uint8_t *type;
    float4 *B, *X, *R, *P, *Q;
};

size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t grid_bytes(int cw, int ch, bool own_ext)
{
    const size_t N = (size_t)cw * ch;
    return al256(N) + 5 * al256(N * 16) + (own_ext ? al256(N * 4) : 0);
}

char *grid_carve(Grid &g, char *b, bool own_ext)
{
    const size_t N = (size_t)g.cw * g.ch;
    g.type = (uint8_t *)b; b += al256(N);
    g.B = (float4 *)b; b += al256(N * 16);
    g.X = (float4 *)b; b += al256(N * 16);
    g.R = (float4 *)b; b += al256(N * 16);
    g.P = (float4 *)b; b += al256(N * 16);
    g.Q = (float4 *)b; b += al256(N * 16);
    if (own_ext) { g.ext = (uchar4 *)b; b += al256(N * 4); }
    return b;
}

// Jacobi-PCG on one grid from the X it holds, until the relative residual <= tol
int run_cg(vm_ctx *c, Grid &g, VmCgScalars *sc, float tol, int max_it, int *iters, double *rel)
{
    hipStream_t s = c->stream;
    VM_HIP(hipMemsetAsync(sc, 0, sizeof(VmCgScalars), s));
    vm_poisson_launch_cg_init(g.B, g.X, g.R, g.P, g.type, sc, g.cw, g.ch, s);
    VmCgScalars h;
    int it = 0;
    double worst = 0;
    const int check = 32;
    while (true) {
        VM_HIP(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, s));
        VM_HIP(hipStreamSynchronize(s));
        worst = 0;
        for (int k = 0; k < 3; ++k)
            if (h.bb[k] > 0) worst = std::max(worst, std::sqrt(h.rr[k] / h.bb[k]));
        if (!(worst == worst)) return vm_fail(VM_E_NUMERIC, "vm_poisson_extend: CG broke down (NaN)");
        if (worst <= tol || it >= max_it) break;
        const int nb = std::min(check, max_it - it);
        for (int k = 0; k < nb; ++k)
            vm_poisson_launch_iter(g.X, g.R, g.P, g.Q, g.B, g.type, sc, g.cw, g.ch, s);
        VM_HIP(hipGetLastError());
        it += nb;
    }
    *iters = it;
    *rel = worst;
    return VM_OK;
}
} // namespace

extern "C" int vm_poisson_extend(vm_frame *f, int side, float tol, int max_it, int *iters,
                                 float *rel_res, float *elapsed_ms)
{
    if (!f || (side != 1 && side != 2) || !(tol > 0) || max_it < 1)
        return vm_fail(VM_E_INVALID, "vm_poisson_extend: bad argument");
    vm_ctx *c = f->ctx;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    hipStream_t s = c->stream;
    // grids: the canvas and 4x / 16x coarser copies while they stay >= 64 pixels wide
    Grid g[3];
    int ng = 1;
    g[0].cw = f->cw; g[0].ch = f->ch;
    while (ng < 3 && std::min(g[ng - 1].cw, g[ng - 1].ch) >= 256) {
        g[ng].cw = (g[ng - 1].cw + 3) / 4;
        g[ng].ch = (g[ng - 1].ch + 3) / 4;
        ++ng;
    }
    size_t need = al256(sizeof(VmCgScalars));
    for (int k = 0; k < ng; ++k) need += grid_bytes(g[k].cw, g[k].ch, k > 0);
    if (f->pws_bytes < need) {
        hipFree(f->pws);
        f->pws = nullptr;
        f->pws_bytes = 0;
        VM_HIP(hipMalloc(&f->pws, need));
        f->pws_bytes = need;
    }
    char *b = (char *)f->pws;
    VmCgScalars *sc = (VmCgScalars *)b;
    b += al256(sizeof(VmCgScalars));
    for (int k = 0; k < ng; ++k) b = grid_carve(g[k], b, k > 0);
    g[0].ext = f->ext[side - 1];
    const uchar4 *other = f->crop[side == 1 ? 1 : 0]; // PoissonExt.cpp:54-57
    const int sign = side == 1 ? 1 : -1;

    VM_HIP(hipEventRecord(c->ev0, s));
    vm_poisson_launch_prepare(g[0].ext, g[0].type, other, f->v, f->w, f->h, f->rs, f->ex, sign, s);
    for (int k = 1; k < ng; ++k)
        vm_poisson_launch_coarsen(g[k - 1].ext, g[k - 1].type, g[k].ext, g[k].type, g[k - 1].cw, g[k - 1].ch,
                                  g[k].cw, g[k].ch, s);
    VM_HIP(hipGetLastError());
    // nested iteration, coarsest grid first; the coarse solves only feed initial guesses,
    // the finest grid is the reference's system and is solved to `tol`
    int total_it = 0, it = 0;
    double rel = 0;
    for (int k = ng - 1; k >= 0; --k) {
        vm_poisson_launch_setup(g[k].ext, g[k].type, g[k].B, g[k].X, g[k].cw, g[k].ch, s);
        if (k < ng - 1)
            vm_poisson_launch_prolong(g[k + 1].X, g[k + 1].type, g[k].X, g[k].type, g[k].cw, g[k].ch,
                                      g[k + 1].cw, g[k + 1].ch, s);
        int rc = run_cg(c, g[k], sc, k == 0 ? tol : std::max(tol, 1e-4f), k == 0 ? max_it : 4000, &it, &rel);
        if (rc != VM_OK) return rc;
        if (k == 0) total_it = it;
    }
    vm_poisson_launch_paste(g[0].ext, g[0].type, g[0].X, g[0].cw, g[0].ch, s);
    VM_HIP(hipGetLastError());
    VM_HIP(hipEventRecord(c->ev1, s));
    VM_HIP(hipEventSynchronize(c->ev1));
    float ms = 0;
    VM_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (iters) *iters = total_it;
    if (rel_res) *rel_res = (float)rel;
    if (elapsed_ms) *elapsed_ms = ms;
    if (rel > tol)
        return vm_fail(VM_E_NUMERIC, "vm_poisson_extend: residual %.3g after %d iterations (tol %.3g)", rel, total_it, (double)tol);
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
