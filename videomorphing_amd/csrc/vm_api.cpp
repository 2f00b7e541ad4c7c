// vm_api.cpp -- implementation of the C-ABI declared in include/vmorph.h:
// contexts, pyramids, host<->device copies, the host-side coarse solve and the
// coarse-to-fine driver.  Kernels live in vm_morph_kernels.hip (optimizer),
// vm_render.hip (compositor) and vm_poisson.hip (boundary extension).
#include "vm_internal.h"
#include <atomic>
#include "vm_host.h"

#ifndef VM_STEP_MAX_TILES
#define VM_STEP_MAX_TILES 64 // AUTO: levels of a batch with at most this many tiles per pass may run STEP
#endif
// (VM_STEP_MAX_TILES in the environment overrides it: dev switch)
static int vm_step_max_tiles()
{
    static const char *e = getenv("VM_STEP_MAX_TILES");
    static const int v = e ? atoi(e) : VM_STEP_MAX_TILES;
    return v;
}
#ifndef VM_STEP_BIG_PARTS
#define VM_STEP_BIG_PARTS 8
#endif
#ifndef VM_CORUN_MIN_WGS
#define VM_CORUN_MIN_WGS 384 // small-level dense workgroups in flight on a device from which 256-thread workgroups pay (1.5 per CU)
#endif
#define VM_MAX_DEVICES_TRACKED 64
#ifndef VM_TILE_LIST_MIN
#define VM_TILE_LIST_MIN 4096 // workgroups of a pruned TILE pass (tiles x pairs) from which the listed form pays
#endif
#ifndef VM_PASS_MAX_GROUPS
#define VM_PASS_MAX_GROUPS 8 // AUTO: PASS instead of STEP while a pass has at most this many tiles (x pairs): one 256-workgroup chunk
#endif
#ifndef VM_SPARSE_TILES
#define VM_SPARSE_TILES 12 // SPARSE takes a pruned level over once <= this many tiles per iteration were active
#endif
// AUTO: STEP / PASS while the previous batch searched at least this many pixels per iteration and pair (below it the
// pruned TILE kernel or SPARSE take over); VM_STEP_MIN_CAND overrides (dev switch)
static double vm_step_min_cand()
{
    static const char *e = getenv("VM_STEP_MIN_CAND");
    static const double v = e ? atof(e) : 200.0;
    return v;
}

static int vm_sparse_tiles()
{
    static const char *e = getenv("VM_SPARSE_TILES"); // dev switch
    static const int v = e ? atoi(e) : VM_SPARSE_TILES;
    return v;
}

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <cerrno>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

static thread_local std::string g_err;

int vm_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char *vm_last_error(void) { return g_err.c_str(); }
extern "C" const char *vm_version(void) { return "vmorph-mi355x 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------
// constant tables (thin-plate stencil per border class, improving-mask window
// bits).  Product-side generator: the stencil row of a pixel is the Hessian
// row of the discrete bending energy  sum (dxx v)^2 + (dyy v)^2 + 2 (dxy v)^2
// over all operator placements that fit in the image -- what
// Algorithm/stencils.cpp:156-261 tabulates per border class.
static void build_tables(uint32_t *tab)
{
    float tps[5][5][5][5];
    memset(tps, 0, sizeof(tps));
    // a 5x5 image realises every border class pair once: pixel (n,m) has class (m,n)
    const int N = 5;
    struct Op { int n; int dx[4], dy[4]; float c[4]; float w; };
    const Op ops[3] = {
        {3, {-1, 0, 1, 0}, {0, 0, 0, 0}, {1, -2, 1, 0}, 1.0f},  // dxx, centred
        {3, {0, 0, 0, 0}, {-1, 0, 1, 0}, {1, -2, 1, 0}, 1.0f},  // dyy, centred
        {4, {0, 1, 0, 1}, {0, 0, 1, 1}, {1, -1, -1, 1}, 2.0f},  // dxy on the cell (x..x+1, y..y+1)
    };
    for (int k = 0; k < 3; ++k)
        for (int cy = 0; cy < N; ++cy)
            for (int cx = 0; cx < N; ++cx) {
                const Op &o = ops[k];
                bool fits = true;
                for (int t = 0; t < o.n; ++t) {
                    int x = cx + o.dx[t], y = cy + o.dy[t];
                    if (x < 0 || x >= N || y < 0 || y >= N) fits = false;
                }
                if (!fits) continue;
                // d/dv_p of w*(sum c_t v_t)^2 = 2 w c_p sum c_t v_t
                for (int p = 0; p < o.n; ++p)
                    for (int t = 0; t < o.n; ++t) {
                        int px = cx + o.dx[p], py = cy + o.dy[p];
                        int qx = cx + o.dx[t], qy = cy + o.dy[t];
                        tps[py][px][qy - py + 2][qx - px + 2] += 2.0f * o.w * o.c[p] * o.c[t];
                    }
            }
    for (int i = 0; i < 625; ++i) {
        float f = (&tps[0][0][0][0])[i];
        memcpy(&tab[VM_TAB_TPS + i], &f, 4);
    }
    // improving-mask bits (stencils.cpp:90-126): for a pixel at (ox,oy) inside
    // its 5x5 block, which bits of the 3x3 neighbouring blocks fall in its window
    for (int oy = 0; oy < 5; ++oy)
        for (int ox = 0; ox < 5; ++ox)
            for (int by = 0; by < 3; ++by)
                for (int bx = 0; bx < 3; ++bx) {
                    uint32_t m = 0;
                    for (int ry = 0; ry < 5; ++ry)
                        for (int rx = 0; rx < 5; ++rx) {
                            int wx = (bx - 1) * 5 + rx - ox, wy = (by - 1) * 5 + ry - oy;
                            if (wx >= -2 && wx <= 2 && wy >= -2 && wy <= 2)
                                m |= 1u << (rx + ry * 5);
                        }
                    tab[VM_TAB_IMP + ((oy * 5 + ox) * 3 + by) * 3 + bx] = m;
                }
}

// ---------------------------------------------------------------------------
static void ctx_free(vm_ctx *c);

// Live contexts.  Destroying a pyramid, video or frame after its context is a caller error; a
// garbage-collected host language can produce that order.  The destroy functions check here and
// then free the object's device buffers without the context (every object remembers its device;
// hipFree needs neither the stream nor the context, after a device-wide synchronise nothing can
// still be using them): no use-after-free and no leak.  Every other entry point refuses an object
// whose context is gone.
#include <set>
static std::mutex g_live_mu;
static std::set<const vm_ctx *> g_live;
bool vm_ctx_alive(const vm_ctx *c)
{
    std::lock_guard<std::mutex> lock(g_live_mu);
    return g_live.count(c) != 0;
}

extern "C" int vm_ctx_create(int device, vm_ctx **out)
{
    if (!out) return vm_fail(VM_E_INVALID, "vm_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return vm_fail(VM_E_DEVICE, "vm_ctx_create: no HIP device (%s); there is no CPU fallback",
                       hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return vm_fail(VM_E_INVALID, "vm_ctx_create: device %d out of range (0..%d)", device, ndev - 1);
    vm_ctx *c = new vm_ctx();
    c->device = device;
    VM_ON_DEVICE(c);
    c->math_mode = VM_MATH_EXACT;
    c->kp = {10.0f, 1e5f, 0.05f, 100.0f, 0.0f, 0.01f, VM_BCOND_NONE}; // UI/MdiEditor.cpp:131-140
    // a context that cannot be completed is torn down again: nothing leaks on the error paths
    uint32_t tab[VM_TAB_WORDS];
    build_tables(tab);
    c->flags_cap = 4096;
    hipError_t e2 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e2 == hipSuccess) e2 = hipEventCreate(&c->ev0);
    if (e2 == hipSuccess) e2 = hipEventCreate(&c->ev1);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming);
    if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&c->xfer_ev, hipEventDisableTiming);
    if (e2 == hipSuccess) e2 = hipEventRecord(c->done_ev, c->stream);
    if (e2 == hipSuccess) e2 = hipMalloc((void **)&c->tables, sizeof(tab));
    if (e2 == hipSuccess) e2 = hipMemcpy(c->tables, tab, sizeof(tab), hipMemcpyHostToDevice);
    if (e2 == hipSuccess) e2 = hipMalloc((void **)&c->flags, c->flags_cap * sizeof(uint32_t));
    if (e2 == hipSuccess) e2 = hipHostMalloc((void **)&c->flags_host, c->flags_cap * sizeof(uint32_t), hipHostMallocDefault);
    if (e2 == hipSuccess) e2 = hipMalloc((void **)&c->stats, c->flags_cap * VM_STAT_WORDS * sizeof(uint32_t));
    if (e2 == hipSuccess) e2 = hipHostMalloc((void **)&c->stats_host, c->flags_cap * VM_STAT_WORDS * sizeof(uint32_t), hipHostMallocDefault);
    if (e2 != hipSuccess) {
        ctx_free(c);
        return vm_fail(VM_E_DEVICE, "vm_ctx_create: %s", hipGetErrorString(e2));
    }
    {
        std::lock_guard<std::mutex> lock(g_live_mu);
        g_live.insert(c);
    }
    *out = c;
    return VM_OK;
}

static void ctx_free(vm_ctx *c)
{
    if (c->stream) hipStreamSynchronize(c->stream);
    hipFree(c->tables);
    hipFree(c->flags);
    hipHostFree(c->flags_host);
    hipFree(c->stats);
    hipHostFree(c->stats_host);
    hipFree(c->step_slots);
    hipFree(c->pass_bar);
    hipFree(c->pass_err);
    hipFree(c->pass_snap);
    hipHostFree(c->pass_err_host);
    hipFree(c->pass_dbg);
    hipFree(c->cons_dev);
    hipFree(c->views);
    hipFree(c->iter_dev);
    hipFree(c->tile_list);
    hipFree(c->mgb_sys);
    hipFree(c->mgb_shared);
    for (auto &g : c->graphs) hipGraphExecDestroy(g.exec);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->done_ev) hipEventDestroy(c->done_ev);
    if (c->xfer_ev) hipEventDestroy(c->xfer_ev);
    if (c->stream) hipStreamDestroy(c->stream);
    (void)hipGetLastError();
    delete c;
}

extern "C" void vm_ctx_destroy(vm_ctx *c)
{
    if (!c || !vm_ctx_alive(c)) return;
    {
        std::lock_guard<std::mutex> lock(g_live_mu);
        g_live.erase(c);
    }
    VM_ON_DEVICE_VOID(c);
    ctx_free(c);
}

extern "C" int vm_ctx_sync(vm_ctx *c)
{
    if (!c) return vm_fail(VM_E_INVALID, "ctx is NULL");
    VM_ON_DEVICE(c);
    VM_HIP(hipStreamSynchronize(c->stream));
    return VM_OK;
}

extern "C" int vm_set_params(vm_ctx *c, const vm_kern_params *p)
{
    if (!c || !p) return vm_fail(VM_E_INVALID, "vm_set_params: NULL argument");
    if (p->bcond < VM_BCOND_NONE || p->bcond > VM_BCOND_BORDER)
        return vm_fail(VM_E_INVALID, "vm_set_params: bcond %d", p->bcond);
    if (!(p->eps > 0)) return vm_fail(VM_E_INVALID, "vm_set_params: eps must be > 0");
    c->kp = *p;
    return VM_OK;
}

extern "C" int vm_get_params(vm_ctx *c, vm_kern_params *p)
{
    if (!c || !p) return vm_fail(VM_E_INVALID, "vm_get_params: NULL argument");
    *p = c->kp;
    return VM_OK;
}

extern "C" int vm_set_math_mode(vm_ctx *c, int mode)
{
    if (!c || mode < VM_MATH_EXACT || mode > VM_MATH_REF_TEX8_TRUNC)
        return vm_fail(VM_E_INVALID, "vm_set_math_mode: bad argument");
    c->math_mode = mode;
    return VM_OK;
}

extern "C" int vm_set_commit_order(vm_ctx *c, int order)
{
    if (!c) return vm_fail(VM_E_INVALID, "vm_set_commit_order: ctx is NULL");
    if (order < 0 || order > 3) return vm_fail(VM_E_INVALID, "vm_set_commit_order: order %d (0 row-major, 1 reversed, 2 column-major, 3 column-major reversed)", order);
    c->commit_order = order;
    return VM_OK;
}

extern "C" int vm_dbg_sparse_resident(vm_ctx *c, int mode)
{
    if (!c) return vm_fail(VM_E_INVALID, "vm_dbg_sparse_resident: ctx is NULL");
    if (mode < 0 || mode > 3) return vm_fail(VM_E_INVALID, "vm_dbg_sparse_resident: mode %d (0 automatic, 1 never, 2 re-centre at every commit, 3 give up at the first commit)", mode);
    c->sparse_resident = mode;
    return VM_OK;
}

extern "C" int vm_dbg_sparse_resident_visits(vm_ctx *c)
{
    if (!c) return vm_fail(VM_E_INVALID, "vm_dbg_sparse_resident_visits: ctx is NULL");
    return (int)std::min<unsigned long long>(c->sparse_resident_visits, 0x7fffffffull);
}

extern "C" int vm_dbg_pass_force_timeout(vm_ctx *c, int on)
{
    if (!c) return vm_fail(VM_E_INVALID, "vm_dbg_pass_force_timeout: ctx is NULL");
    c->pass_test_timeout = on ? 1 : 0;
    // switching the hook off re-admits the context to PASS only if the HOOK latched it off: a latch set by a
    // genuine barrier timeout (masked or shared compute units) stays
    if (!on && c->pass_latched_by_test) {
        c->pass_latched_off = false;
        c->pass_latched_by_test = false;
    }
    return VM_OK;
}

extern "C" int vm_dbg_pass_fallbacks(vm_ctx *c)
{
    return c ? c->pass_fallbacks : -1;
}

extern "C" int vm_dbg_pass_placement(vm_ctx *c, uint8_t *xcc_of_block, int n)
{
    if (!c || !xcc_of_block || n < 1 || n > 2048) return vm_fail(VM_E_INVALID, "vm_dbg_pass_placement: bad argument");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    VM_ON_DEVICE(c);
    if (!c->pass_dbg) { // arm: the next PASS launches record where their workgroups run
        VM_HIP(hipMalloc((void **)&c->pass_dbg, 2048 * sizeof(uint32_t)));
        VM_HIP(hipMemsetAsync(c->pass_dbg, 0xFF, 2048 * sizeof(uint32_t), c->stream));
        VM_HIP(hipStreamSynchronize(c->stream));
        memset(xcc_of_block, 0xFF, (size_t)n);
        return VM_OK;
    }
    std::vector<uint32_t> h(2048);
    VM_HIP(hipStreamSynchronize(c->stream));
    VM_HIP(hipMemcpy(h.data(), c->pass_dbg, 2048 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < n; ++k) xcc_of_block[k] = (uint8_t)(h[k] & 0xFFu);
    return VM_OK;
}

extern "C" int vm_set_tuning(vm_ctx *c, int sweep_mode, int threads, int parts)
{
    if (!c || sweep_mode < VM_SWEEP_AUTO || sweep_mode > VM_SWEEP_PASS || threads < 0 || parts < 0 ||
        (threads && (threads % 64 || threads < 256 || threads > 1024)) || parts > 64)
        return vm_fail(VM_E_INVALID, "vm_set_tuning: bad argument");
    c->sweep_mode = sweep_mode;
    c->sweep_threads = threads;
    c->sweep_parts = parts;
    return VM_OK;
}

extern "C" int vm_device_info(vm_ctx *c, char *name256, int *cus, uint64_t *hbm)
{
    if (!c) return vm_fail(VM_E_INVALID, "ctx is NULL");
    hipDeviceProp_t pr;
    VM_HIP(hipGetDeviceProperties(&pr, c->device));
    if (name256) snprintf(name256, 256, "%s (%s)", pr.name, pr.gcnArchName);
    if (cus) *cus = pr.multiProcessorCount;
    if (hbm) *hbm = (uint64_t)pr.totalGlobalMem;
    return VM_OK;
}

// ---------------------------------------------------------------------------
void vm_level_free(vm_level &l)
{
    hipFree(l.slab);
    hipFree(l.ws);
    hipFree(l.sp_ws);
    l.slab = l.ws = l.sp_ws = nullptr;
    l.has_state = false;
    VmLevelView &V = l.view;
    V.rec_a = V.rec_b = V.rec_a2 = V.rec_b2 = nullptr;
    V.rec_tag = V.rec_tag2 = nullptr;
    V.mean2 = V.var2 = V.tps_b2 = nullptr;
    V.cross2 = V.value2 = nullptr;
    V.impmask2 = nullptr;
    V.temp_ref = nullptr;
    V.temp_mask = nullptr;
    V.sp_wl = V.sp_cnt = V.sp_stamp = nullptr;
}

// one slab per level: every array starts on a 256-byte boundary
int vm_level_alloc(vm_ctx *c, vm_level &l, bool with_images)
{
    size_t n = (size_t)l.rs * l.h;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off = 0;
    size_t o_v = off; off += al(n * 8);
    size_t o_img0 = off, o_img1 = off, o_luma = off, o_mean = off, o_var = off, o_tpsb = off,
           o_uib = off, o_cross = off, o_value = off, o_uiaxy = off, o_imp = off;
    if (with_images) {
        o_img0 = off; off += al(n * 4);
        o_img1 = off; off += al(n * 4);
        o_luma = off; off += al(n * 8);
        o_mean = off; off += al(n * 8);
        o_var = off; off += al(n * 8);
        o_tpsb = off; off += al(n * 8);
        o_uib = off; off += al(n * 8);
        o_cross = off; off += al(n * 4);
        o_value = off; off += al(n * 4);
        o_uiaxy = off; off += al(n * 4);
        o_imp = off; off += al((size_t)l.imp_rs * l.imp_rows * 4);
    }
    VM_HIP(hipMalloc((void **)&l.slab, off));
    // stream-ordered: the context's stream does not synchronise with the null stream
    VM_HIP(hipMemsetAsync(l.slab, 0, off, c->stream));
    l.slab_bytes = off;
    char *b = (char *)l.slab;
    VmLevelView &V = l.view;
    V.w = l.w; V.h = l.h; V.rs = l.rs;
    V.inv_wh = 1.0f / (l.w * l.h);          // pyramid.cu:537
    V.imp_rs = l.imp_rs; V.imp_rows = l.imp_rows;
    V.v = (float2 *)(b + o_v);
    if (with_images) {
        V.img0 = (const float *)(b + o_img0); V.img1 = (const float *)(b + o_img1);
        V.luma = (float2 *)(b + o_luma); V.mean = (float2 *)(b + o_mean);
        V.var = (float2 *)(b + o_var); V.tps_b = (float2 *)(b + o_tpsb);
        V.ui_b = (float2 *)(b + o_uib); V.cross = (float *)(b + o_cross);
        V.value = (float *)(b + o_value); V.ui_axy = (float *)(b + o_uiaxy);
        V.impmask = (uint32_t *)(b + o_imp);
    } else {
        V.img0 = V.img1 = nullptr;
        V.luma = V.mean = V.var = V.tps_b = V.ui_b = nullptr;
        V.cross = V.value = V.ui_axy = nullptr;
        V.impmask = nullptr;
    }
    V.rec_a = V.rec_b = V.rec_a2 = V.rec_b2 = nullptr;
    V.rec_tag = V.rec_tag2 = nullptr;
    V.mean2 = V.var2 = V.tps_b2 = nullptr;
    V.cross2 = V.value2 = nullptr;
    V.impmask2 = nullptr;
    V.temp_ref = nullptr;
    V.temp_mask = nullptr;
    V.factor_d = 1.0f;
    V.sp_wl = V.sp_cnt = V.sp_stamp = nullptr;
    return VM_OK;
}

// The workspace of the SPLIT / STEP schedules (two record sets, the second copy of the sums
// and of the mask: 104 B per pixel against 72 B of solver state) is allocated the first time
// a level is swept with one of them -- in practice the small levels only.
static int level_ensure_ws(vm_ctx *c, vm_level &l)
{
    if (l.ws) return VM_OK;
    const size_t n = (size_t)l.rs * l.h, nimp = (size_t)l.imp_rs * l.imp_rows;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t total = 2 * al(n * 4) + 4 * al(n * 16) + 3 * al(n * 8) + 2 * al(n * 4) + al(nimp * 4);
    VM_HIP(hipMalloc((void **)&l.ws, total));
    VM_HIP(hipMemsetAsync(l.ws, 0, total, c->stream));
    char *b = (char *)l.ws;
    VmLevelView &V = l.view;
    V.rec_tag = (uint32_t *)b; b += al(n * 4);
    V.rec_tag2 = (uint32_t *)b; b += al(n * 4);
    V.rec_a = (float4 *)b; b += al(n * 16);
    V.rec_b = (float4 *)b; b += al(n * 16);
    V.rec_a2 = (float4 *)b; b += al(n * 16);
    V.rec_b2 = (float4 *)b; b += al(n * 16);
    V.mean2 = (float2 *)b; b += al(n * 8);
    V.var2 = (float2 *)b; b += al(n * 8);
    V.tps_b2 = (float2 *)b; b += al(n * 8);
    V.cross2 = (float *)b; b += al(n * 4);
    V.value2 = (float *)b; b += al(n * 4);
    V.impmask2 = (uint32_t *)b;
    return VM_OK;
}

// The workspace of the SPARSE schedule (vm_sweep_kernels.hip): two lists of mask-word indices,
// their lengths and a stamp per word -- 12 B per 5x5 block, allocated on first use.
static int level_ensure_sparse(vm_ctx *c, vm_level &l)
{
    if (l.sp_ws) return VM_OK;
    const size_t nw = (size_t)l.imp_rs * l.imp_rows;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t total = al(2 * nw * 4) + al(nw * 4) + 256;
    VM_HIP(hipMalloc((void **)&l.sp_ws, total));
    VM_HIP(hipMemsetAsync(l.sp_ws, 0, total, c->stream));
    char *b = (char *)l.sp_ws;
    l.view.sp_wl = (uint32_t *)b; b += al(2 * nw * 4);
    l.view.sp_stamp = (uint32_t *)b; b += al(nw * 4);
    l.view.sp_cnt = (uint32_t *)b;
    return VM_OK;
}

extern "C" int vm_pyramid_create(vm_ctx *c, int nlevels, const int *w, const int *h, vm_pyr **out)
{
    if (!c || !w || !h || !out || nlevels < 2)
        return vm_fail(VM_E_INVALID, "vm_pyramid_create: need ctx, sizes and >= 2 levels");
    for (int i = 0; i < nlevels; ++i)
        if (w[i] < 5 || h[i] < 5)
            return vm_fail(VM_E_INVALID, "vm_pyramid_create: level %d is %dx%d (min 5x5)", i, w[i], h[i]);
    VM_ON_DEVICE(c);
    vm_pyr *p = new vm_pyr();
    p->ctx = c;
    p->device = c->device;
    p->lv.resize(nlevels);
    for (int i = 0; i < nlevels; ++i) {
        vm_level &l = p->lv[i];
        l.w = w[i]; l.h = h[i];
        l.rs = (w[i] + 31) / 32 * 32;               // pyramid.cu:535
        l.imp_rs = (w[i] + 4) / 5 + 2;              // pyramid.cu:538
        l.imp_rows = (h[i] + 4) / 5 + 2;            // pyramid.cu:539
        int rc = vm_level_alloc(c, l, i != nlevels - 1);
        if (rc != VM_OK) { vm_pyramid_destroy(p); return rc; }
    }
    *out = p;
    return VM_OK;
}

extern "C" void vm_pyramid_destroy(vm_pyr *p)
{
    if (!p) return;
    if (!vm_ctx_alive(p->ctx)) { // destroyed after its context: the buffers are freed without it
        VmDeviceGuard g(p->device);
        if (g.ok) {
            hipDeviceSynchronize();
            for (auto &l : p->lv) vm_level_free(l);
            (void)hipGetLastError();
        }
        delete p;
        return;
    }
    VM_ON_DEVICE_VOID(p->ctx);
    hipStreamSynchronize(p->ctx->stream);
    for (auto &l : p->lv) vm_level_free(l);
    delete p;
}

extern "C" int vm_pyramid_levels(vm_pyr *p) { return p ? (int)p->lv.size() : 0; }

#define CHECK_LVL(p, lvl)                                                        \
    if (!(p)) return vm_fail(VM_E_INVALID, "%s: pyramid is NULL", __func__);     \
    if ((lvl) < 0 || (lvl) >= (int)(p)->lv.size())                               \
        return vm_fail(VM_E_INVALID, "%s: level %d out of range", __func__, (lvl)); \
    if (!vm_ctx_alive((p)->ctx)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__); \
    VM_ON_DEVICE((p)->ctx);

extern "C" int vm_level_dims(vm_pyr *p, int lvl, int *w, int *h, int *rs)
{
    CHECK_LVL(p, lvl);
    if (w) *w = p->lv[lvl].w;
    if (h) *h = p->lv[lvl].h;
    if (rs) *rs = p->lv[lvl].rs;
    return VM_OK;
}

extern "C" int vm_level_upload_luma(vm_pyr *p, int lvl, const float *img0, const float *img1, int pitch)
{
    CHECK_LVL(p, lvl);
    vm_level &l = p->lv[lvl];
    if (!l.view.img0) return vm_fail(VM_E_STATE, "vm_level_upload_luma: coarsest level holds no images");
    if (!img0 || !img1) return vm_fail(VM_E_INVALID, "vm_level_upload_luma: NULL image");
    if (pitch == 0) pitch = l.w;
    if (pitch < l.w) return vm_fail(VM_E_INVALID, "vm_level_upload_luma: pitch < width");
    hipStream_t s = p->ctx->stream;
    VM_HIP(hipMemcpy2DAsync((void *)l.view.img0, l.rs * 4, img0, (size_t)pitch * 4, (size_t)l.w * 4, l.h, hipMemcpyHostToDevice, s));
    VM_HIP(hipMemcpy2DAsync((void *)l.view.img1, l.rs * 4, img1, (size_t)pitch * 4, (size_t)l.w * 4, l.h, hipMemcpyHostToDevice, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

extern "C" int vm_level_set_v(vm_pyr *p, int lvl, const float *v, int pitch)
{
    CHECK_LVL(p, lvl);
    vm_level &l = p->lv[lvl];
    if (!v) return vm_fail(VM_E_INVALID, "vm_level_set_v: NULL");
    if (pitch == 0) pitch = 2 * l.w;
    if (pitch < 2 * l.w) return vm_fail(VM_E_INVALID, "vm_level_set_v: pitch < 2*width");
    hipStream_t s = p->ctx->stream;
    VM_HIP(hipMemcpy2DAsync(l.view.v, l.rs * 8, v, (size_t)pitch * 4, (size_t)l.w * 8, l.h, hipMemcpyHostToDevice, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

extern "C" int vm_level_get_v(vm_pyr *p, int lvl, float *v, int pitch)
{
    CHECK_LVL(p, lvl);
    vm_level &l = p->lv[lvl];
    if (!v) return vm_fail(VM_E_INVALID, "vm_level_get_v: NULL");
    if (pitch == 0) pitch = 2 * l.w;
    if (pitch < 2 * l.w) return vm_fail(VM_E_INVALID, "vm_level_get_v: pitch < 2*width");
    hipStream_t s = p->ctx->stream;
    VM_HIP(hipMemcpy2DAsync(v, (size_t)pitch * 4, l.view.v, l.rs * 8, (size_t)l.w * 8, l.h, hipMemcpyDeviceToHost, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

extern "C" int vm_level_get_field(vm_pyr *p, int lvl, int field, void *host)
{
    CHECK_LVL(p, lvl);
    return vm_level_read_field(p->ctx, p->lv[lvl], field, host);
}

int vm_level_read_field(vm_ctx *c, vm_level &l, int field, void *host)
{
    if (!host) return vm_fail(VM_E_INVALID, "vm_level_get_field: NULL");
    const VmLevelView &V = l.view;
    hipStream_t s = c->stream;
    const void *src = nullptr;
    int elem = 4;
    switch (field) {
    case VM_F_IMG0: src = V.img0; break;
    case VM_F_IMG1: src = V.img1; break;
    case VM_F_V: src = V.v; elem = 8; break;
    case VM_F_LUMA: src = V.luma; elem = 8; break;
    case VM_F_MEAN: src = V.mean; elem = 8; break;
    case VM_F_VAR: src = V.var; elem = 8; break;
    case VM_F_CROSS: src = V.cross; break;
    case VM_F_VALUE: src = V.value; break;
    case VM_F_TPS_B: src = V.tps_b; elem = 8; break;
    case VM_F_UI_AXY: src = V.ui_axy; break;
    case VM_F_UI_B: src = V.ui_b; elem = 8; break;
    case VM_F_TEMP_REF: src = l.temp_ref_store; elem = 8; break;
    case VM_F_TEMP_MASK: src = l.temp_mask_store; break;
    case VM_F_COUNTER: {
        // pure function of position (morph.cu:225): not stored on the device
        float *o = (float *)host;
        for (int y = 0; y < l.h; ++y)
            for (int x = 0; x < l.w; ++x)
                o[(size_t)y * l.w + x] = (float)((std::min(y, 2) + std::min(l.h - 1 - y, 2) + 1) *
                                                  (std::min(x, 2) + std::min(l.w - 1 - x, 2) + 1));
        return VM_OK;
    }
    case VM_F_TPS_AXY: {
        // tps[B][2][2]/2 (morph.cu:242): not stored on the device
        uint32_t tab[VM_TAB_WORDS];
        build_tables(tab);
        auto cls = [](int q, int dim) { return q < 2 ? q : (q == dim - 2 ? 3 : (q == dim - 1 ? 4 : 2)); };
        float *o = (float *)host;
        for (int y = 0; y < l.h; ++y)
            for (int x = 0; x < l.w; ++x) {
                float f;
                memcpy(&f, &tab[VM_TAB_TPS + (cls(y, l.h) * 5 + cls(x, l.w)) * 25 + 12], 4);
                o[(size_t)y * l.w + x] = f / 2;
            }
        return VM_OK;
    }
    case VM_F_IMPMASK:
        if (!V.impmask) return vm_fail(VM_E_STATE, "level has no state");
        VM_HIP(hipMemcpyAsync(host, V.impmask, (size_t)l.imp_rs * l.imp_rows * 4, hipMemcpyDeviceToHost, s));
        VM_HIP(hipStreamSynchronize(s));
        return VM_OK;
    default:
        return vm_fail(VM_E_INVALID, "vm_level_get_field: unknown field %d", field);
    }
    if (!src) return vm_fail(VM_E_STATE, "vm_level_get_field: the level has no such array");
    VM_HIP(hipMemcpy2DAsync(host, (size_t)l.w * elem, src, (size_t)l.rs * elem, (size_t)l.w * elem, l.h, hipMemcpyDeviceToHost, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

extern "C" int vm_dbg_level_set_mask(vm_pyr *p, int lvl, const uint32_t *words)
{
    CHECK_LVL(p, lvl);
    vm_level &l = p->lv[lvl];
    if (!words) return vm_fail(VM_E_INVALID, "vm_dbg_level_set_mask: NULL");
    if (!l.has_state || !l.view.impmask) return vm_fail(VM_E_STATE, "vm_dbg_level_set_mask: level not initialised");
    hipStream_t s = p->ctx->stream;
    VM_HIP(hipMemcpyAsync(l.view.impmask, words, (size_t)l.imp_rs * l.imp_rows * 4, hipMemcpyHostToDevice, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

extern "C" int vm_level_clear(vm_pyr *p, int lvl)
{
    CHECK_LVL(p, lvl);
    // Morph::clear_level frees the per-level state; the slab stays allocated
    // (288 GB of HBM: reuse beats hipFree/hipMalloc churn), it is only marked stale
    p->lv[lvl].has_state = false;
    return VM_OK;
}

// ---------------------------------------------------------------------------
static int upload_constraints(vm_ctx *c, const vm_constraint *cons, int n)
{
    if (n <= 0) return VM_OK;
    if (n > c->cons_cap) {
        hipFree(c->cons_dev);
        c->cons_dev = nullptr;
        c->cons_cap = 0;
        VM_HIP(hipMalloc((void **)&c->cons_dev, (size_t)n * sizeof(vm_constraint)));
        c->cons_cap = n;
    }
    VM_HIP(hipMemcpyAsync(c->cons_dev, cons, (size_t)n * sizeof(vm_constraint), hipMemcpyHostToDevice, c->stream));
    VM_HIP(hipStreamSynchronize(c->stream)); // the host buffer belongs to the caller
    return VM_OK;
}

extern "C" int vm_coarse_solve(vm_pyr *p, int lvl, int w0, int h0, const vm_constraint *cons, int n)
{
    CHECK_LVL(p, lvl);
    if (n < 0 || (n > 0 && !cons)) return vm_fail(VM_E_INVALID, "vm_coarse_solve: constraints");
    vm_level &l = p->lv[lvl];
    std::vector<float> v((size_t)2 * l.w * l.h, 0.0f);
    int rc = vm_host_coarse_solve(l.w, l.h, w0, h0, p->ctx->kp, cons, n, v.data());
    if (rc != VM_OK) return rc;
    return vm_level_set_v(p, lvl, v.data(), 0);
}

// upsample(PyramidLevel&dest, PyramidLevel&orig) for one page, upsample.cu:260-286
int vm_level_upsample(vm_ctx *c, vm_level &d, const vm_level &s)
{
    if (c->math_mode == VM_MATH_REF_TEX8)          // the reference upsamples through a linear-filtered texture too
        vm_launch_upsample_tex8(d.view.v, d.w, d.h, d.rs, s.view.v, s.w, s.h, s.rs, c->stream);
    else if (c->math_mode == VM_MATH_REF_TEX8_TRUNC)
        vm_launch_upsample_tex8t(d.view.v, d.w, d.h, d.rs, s.view.v, s.w, s.h, s.rs, c->stream);
    else if (c->math_mode != VM_MATH_FAST)
        vm_launch_upsample_exact(d.view.v, d.w, d.h, d.rs, s.view.v, s.w, s.h, s.rs, c->stream);
    else
        vm_launch_upsample_fast(d.view.v, d.w, d.h, d.rs, s.view.v, s.w, s.h, s.rs, c->stream);
    VM_HIP(hipGetLastError());
    return VM_OK;
}

extern "C" int vm_upsample_v(vm_pyr *p, int dst, int src)
{
    CHECK_LVL(p, dst);
    CHECK_LVL(p, src);
    return vm_level_upsample(p->ctx, p->lv[dst], p->lv[src]);
}

extern "C" int vm_init_level(vm_pyr *p, int lvl, int w0, int h0, const vm_constraint *cons, int n)
{
    CHECK_LVL(p, lvl);
    return vm_level_init(p->ctx, p->lv[lvl], w0, h0, cons, n);
}

// Morph::initialize_level for one page, morph.cu:264-390
int vm_level_init(vm_ctx *c, vm_level &l, int w0, int h0, const vm_constraint *cons, int n)
{
    if (!l.view.img0) return vm_fail(VM_E_STATE, "vm_init_level: the coarsest level is solved by vm_coarse_solve");
    if (n < 0 || (n > 0 && !cons)) return vm_fail(VM_E_INVALID, "vm_init_level: constraints");
    int rc = upload_constraints(c, cons, n);
    if (rc != VM_OK) return rc;
    if (c->math_mode == VM_MATH_REF_TEX8) {
        vm_launch_init_level_tex8(l.view, c->kp.ssim_clamp, c->tables, c->stream);
        if (n > 0) vm_launch_splat_exact(l.view, w0, h0, c->cons_dev, n, c->stream);
    } else if (c->math_mode == VM_MATH_REF_TEX8_TRUNC) {
        vm_launch_init_level_tex8t(l.view, c->kp.ssim_clamp, c->tables, c->stream);
        if (n > 0) vm_launch_splat_exact(l.view, w0, h0, c->cons_dev, n, c->stream);
    } else if (c->math_mode != VM_MATH_FAST) {
        vm_launch_init_level_exact(l.view, c->kp.ssim_clamp, c->tables, c->stream);
        if (n > 0) vm_launch_splat_exact(l.view, w0, h0, c->cons_dev, n, c->stream);
    } else {
        vm_launch_init_level_fast(l.view, c->kp.ssim_clamp, c->tables, c->stream);
        if (n > 0) vm_launch_splat_fast(l.view, w0, h0, c->cons_dev, n, c->stream);
    }
    VM_HIP(hipGetLastError());
    l.has_state = true;
    return VM_OK;
}

// How many workgroups the dense sweeps of SMALL levels (<= 32 tiles per pass: the 256-VGPR kernel without the
// interior form, one 512-thread workgroup = 8 waves = all of a CU's registers) have in flight on a device, summed
// over the contexts of this process that are sweeping such a level right now.  A batch of 30 pairs x 8 tiles
// = 240 workgroups fills the chip one per CU; a second stream's 240 then wait for them.  As 256-thread
// workgroups (4 waves: a tile's ~127 candidates of a phase at two lanes each; a full phase in two rounds) two
// fit a CU -- 2 x 75 KB of LDS, 2 x 4 waves x 256 VGPRs -- and the two streams' tiles run side by side, each SIMD
// with two searching waves instead of one: config[2]'s 60 pairs on one GPU 968 -> 890 ms.  It only pays when the
// workgroups in flight exceed the CUs by enough (measured: 2 x 240 and 1 x 840 yes; 1 x 240, 2 x 120 no: -16 %),
// so the rule counts them.  Results do not depend on the workgroup size (the lane fan-out per candidate, which
// orders the FAST sums, is a compile-time constant of the kernel).
static std::atomic<int> g_small_dense_wgs[VM_MAX_DEVICES_TRACKED];
struct SmallDensePresence {
    int dev = -1, wgs = 0;
    void enter(int device, int n_wgs)
    {
        if (dev >= 0 || device < 0 || device >= VM_MAX_DEVICES_TRACKED) return;
        dev = device;
        wgs = n_wgs;
        g_small_dense_wgs[dev].fetch_add(wgs);
    }
    void leave()
    {
        if (dev >= 0) g_small_dense_wgs[dev].fetch_sub(wgs);
        dev = -1;
    }
    int in_flight() const { return dev < 0 ? 0 : g_small_dense_wgs[dev].load(); }
    ~SmallDensePresence() { leave(); }
};

// the sweep launchers of one arithmetic build of vm_sweep_kernels.hip
struct SweepLaunchers {
    decltype(&vm_launch_optimize_exact) optimize;
    decltype(&vm_launch_next_iter_exact) next_iter;
    decltype(&vm_launch_optimize_sparse_exact) sparse;
    decltype(&vm_launch_optimize_split_exact) split;
    decltype(&vm_launch_optimize_step_exact) step;
    decltype(&vm_launch_optimize_pass_exact) pass;
    decltype(&vm_pass_resident_blocks_exact) pass_resident;
};
static const SweepLaunchers &sweep_launchers(int math_mode)
{
    static const SweepLaunchers exact = {vm_launch_optimize_exact, vm_launch_next_iter_exact, vm_launch_optimize_sparse_exact,
                                         vm_launch_optimize_split_exact, vm_launch_optimize_step_exact, vm_launch_optimize_pass_exact,
                                         vm_pass_resident_blocks_exact};
    static const SweepLaunchers fast = {vm_launch_optimize_fast, vm_launch_next_iter_fast, vm_launch_optimize_sparse_fast,
                                        vm_launch_optimize_split_fast, vm_launch_optimize_step_fast, vm_launch_optimize_pass_fast,
                                        vm_pass_resident_blocks_fast};
    // VM_MATH_EXACT_FMA: the EXACT source with -ffp-contract=fast (fused multiply-adds wherever the compiler
    // contracts, IEEE division and square root): what nvcc's default --fmad=true makes of the reference source
    static const SweepLaunchers exactf = {vm_launch_optimize_exactf, vm_launch_next_iter_exactf, vm_launch_optimize_sparse_exactf,
                                          vm_launch_optimize_split_exactf, vm_launch_optimize_step_exactf, vm_launch_optimize_pass_exactf,
                                          vm_pass_resident_blocks_exactf};
    // VM_MATH_REF_FASTMATH: that source as the reference's project file compiles it (--use_fast_math)
    static const SweepLaunchers reffm = {vm_launch_optimize_reffm, vm_launch_next_iter_reffm, vm_launch_optimize_sparse_reffm,
                                         vm_launch_optimize_split_reffm, vm_launch_optimize_step_reffm, vm_launch_optimize_pass_reffm,
                                         vm_pass_resident_blocks_reffm};
    // VM_MATH_REF_TEX8 / _TRUNC: that source, IEEE, with the 8-bit bilinear weights of CUDA's texture filter
    static const SweepLaunchers tex8 = {vm_launch_optimize_tex8, vm_launch_next_iter_tex8, vm_launch_optimize_sparse_tex8,
                                        vm_launch_optimize_split_tex8, vm_launch_optimize_step_tex8, vm_launch_optimize_pass_tex8,
                                        vm_pass_resident_blocks_tex8};
    static const SweepLaunchers tex8t = {vm_launch_optimize_tex8t, vm_launch_next_iter_tex8t, vm_launch_optimize_sparse_tex8t,
                                         vm_launch_optimize_split_tex8t, vm_launch_optimize_step_tex8t, vm_launch_optimize_pass_tex8t,
                                         vm_pass_resident_blocks_tex8t};
    switch (math_mode) {
    case VM_MATH_FAST: return fast;
    case VM_MATH_EXACT_FMA: return exactf;
    case VM_MATH_REF_FASTMATH: return reffm;
    case VM_MATH_REF_TEX8: return tex8;
    case VM_MATH_REF_TEX8_TRUNC: return tex8t;
    default: return exact;
    }
}

// A hipGraph of VM_GRAPH_ITERS TILE-schedule iterations (4 pass launches each, one counter bump)
// for the given geometry, instantiated once per context and replayed: pruned sweeps last 2-3 us
// on the GPU, less than the 4-6 us the host needs per eager launch, so the sweep loop of a
// converged or nearly converged level is launch-bound without it.  The iteration number is not
// a kernel argument there but a device counter.  Returns nullptr when graphs are unavailable
// (VM_NO_GRAPH set, or capture/instantiation failed once): the caller launches eagerly.
#define VM_GRAPH_ITERS 8
static hipGraphExec_t sweep_graph(vm_ctx *c, int math_mode, int n, int w, int h, int cap, int fixed_work, int threads,
                                  int dense, uint32_t *tile_list, const VmKParams &P)
{
    if (c->use_graphs < 0) c->use_graphs = getenv("VM_NO_GRAPH") ? 0 : 1;
    if (!c->use_graphs) return nullptr;
    for (auto &g : c->graphs)
        if (g.math_mode == math_mode && g.n == n && g.w == w && g.h == h && g.cap == cap && g.fixed_work == fixed_work &&
            g.threads == threads && g.dense == dense && g.order == c->commit_order && g.views == c->views && g.flags == c->flags && g.stats == c->stats && g.tile_list == tile_list &&
            memcmp(&g.kp, &c->kp, sizeof(c->kp)) == 0)
            return g.exec;
    if (!c->iter_dev && hipMalloc((void **)&c->iter_dev, sizeof(int)) != hipSuccess) {
        c->use_graphs = 0;
        return nullptr;
    }
    const int offs[4][2] = {{0, 0}, {VM_TILE_W, 0}, {0, VM_TILE_H}, {VM_TILE_W, VM_TILE_H}};
    const SweepLaunchers &SL = sweep_launchers(math_mode);
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        for (int it = 0; it < VM_GRAPH_ITERS; ++it) {
            for (int k = 0; k < 4; ++k) {
                SL.optimize(c->views, n, cap, w, h, P, c->tables, offs[k][0], offs[k][1], c->flags, c->stats, it, fixed_work, threads, c->iter_dev, dense, tile_list, c->stream);
            }
        }
        SL.next_iter(c->iter_dev, 0, VM_GRAPH_ITERS, c->stream);
        ok = hipStreamEndCapture(c->stream, &graph) == hipSuccess && graph;
    }
    if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) hipGraphDestroy(graph);
    (void)hipGetLastError();
    if (!ok) {
        c->use_graphs = 0;
        return nullptr;
    }
    if (c->graphs.size() >= 64) { // plenty for a pyramid's levels; start over rather than grow
        for (auto &g : c->graphs) hipGraphExecDestroy(g.exec);
        c->graphs.clear();
    }
    c->graphs.push_back({math_mode, n, w, h, cap, fixed_work, threads, dense, c->commit_order, c->views, c->flags, c->stats, tile_list, c->kp, exec});
    return exec;
}

// The PASS token of a device.  k_pass spins at tile-local barriers, so the workgroups of all its tile
// groups must become co-resident; two PASS launches at once -- of two contexts, or of two PROCESSES
// sharing the device -- could hold part of the compute units each and starve each other's groups.  One
// holder at a time: inside the process a mutex per device, across processes an advisory flock() on a lock
// file named after the device's PCI bus id (so that HIP_VISIBLE_DEVICES renumbering cannot split it);
// whoever does not get the token runs STEP for that call.  Kernels that do not spin (every other
// schedule, any other program) only delay a PASS launch: they finish and free their compute units.
// VM_LOCK_DIR (default /tmp) holds the files; if one cannot be opened or locked the PASS schedule stays off in this
// process (STEP instead); the bounded barrier wait + the STEP rerun below remain the safety net for everything else.
namespace {
struct PassDevice {
    std::mutex mu;
    int fd = -2; // -2: not opened yet, -3: no usable lock file (PASS stays off in this process), >= 0: the lock file
};
PassDevice g_pass_dev[64];

struct PassToken {
    PassDevice *d = nullptr;
    bool owns = false;
    bool try_acquire(int device)
    {
        d = &g_pass_dev[device & 63];
        if (!d->mu.try_lock()) return false;
        if (d->fd == -2) {
            char bus[64] = "";
            if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
                (void)hipGetLastError();
                snprintf(bus, sizeof(bus), "ordinal%d", device);
            }
            for (char *q = bus; *q; ++q)
                if (*q == ':' || *q == '/' || *q == '.') *q = '_';
            const char *dir = getenv("VM_LOCK_DIR");
            const std::string path = std::string(dir && *dir ? dir : "/tmp") + "/vmorph-pass-" + bus + ".lock";
            // Open an existing file first: with fs.protected_regular (the default of many distributions) another
            // user's O_CREAT open of an existing file in a sticky directory fails with EACCES although a plain open
            // succeeds.  Create it only if it is not there (world-readable is all flock() needs).
            d->fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
            if (d->fd < 0 && errno == ENOENT) {
                d->fd = open(path.c_str(), O_RDONLY | O_CREAT | O_CLOEXEC, 0666);
                if (d->fd >= 0) (void)fchmod(d->fd, 0666); // readable by every user of the device whatever this process' umask
            }
            if (d->fd < 0) {
                // no lock file: exclusivity across processes cannot be had.  Refuse PASS rather than run it on a
                // process-local token -- two processes in PASS at once time out against each other (STEP is the
                // schedule of whoever does not hold the token anyway).  Said once.
                d->fd = -3;
                fprintf(stderr, "vmorph: cannot open %s (%s): the PASS schedule stays off on this device in this process; "
                                "set VM_LOCK_DIR to a directory every user of the device can read\n", path.c_str(), strerror(errno));
            }
        }
        if (d->fd == -3) {
            d->mu.unlock();
            return false;
        }
        if (d->fd >= 0) {
            int rc;
            do rc = flock(d->fd, LOCK_EX | LOCK_NB); while (rc != 0 && errno == EINTR);
            if (rc != 0) {
                if (errno != EWOULDBLOCK && errno != ENOLCK && errno != EOPNOTSUPP && errno != EINVAL) {
                    d->mu.unlock();                 // an error that says nothing about the holder: not this time
                    return false;
                }
                if (errno == EWOULDBLOCK) {         // another process holds the device's token
                    d->mu.unlock();
                    return false;
                }
                close(d->fd);                       // a file system without flock(): same as no lock file
                d->fd = -3;
                d->mu.unlock();
                return false;
            }
        }
        owns = true;
        return true;
    }
    void release()
    {
        if (!owns) return;
        if (d->fd >= 0) (void)flock(d->fd, LOCK_UN);
        d->mu.unlock();
        owns = false;
    }
    ~PassToken() { release(); }
};
} // namespace

// Morph::optimize_level for a BATCH of frame pairs of identical geometry on one context:
// every sweep launch covers the same level of all pairs (grid.z = pair), so a level with
// too few tiles to occupy 256 CUs is filled by the batch instead -- the natural parallelism
// of the path (independent pairs).  Each pair keeps its own convergence
// flags: one that stopped improving turns into mask-pruned no-ops while the others go on.
static int optimize_level_batch(vm_pyr **ps, int n, int lvl, float max_iter, volatile const int *run_flag,
                                int fixed_work, vm_progress *out)
{
    vm_pyr *p0 = ps[0];
    vm_ctx *c = p0->ctx;
    std::vector<vm_level *> lv(n);
    for (int i = 0; i < n; ++i) {
        if (!ps[i] || ps[i]->ctx != c) return vm_fail(VM_E_INVALID, "batch: pyramids must share one context");
        if (lvl < 0 || lvl >= (int)ps[i]->lv.size()) return vm_fail(VM_E_INVALID, "batch: level %d out of range", lvl);
        lv[i] = &ps[i]->lv[lvl];
    }
    return vm_optimize_levels(c, lv.data(), n, max_iter, run_flag, fixed_work, out);
}

// The iteration count of `do { ... iter++; } while (iter < _max_iter && ...)` (morph.cu:1378-1390)
// for the float _max_iter of morph.h:20: max(1, ceil(max_iter)).  Not finite, or beyond 2^20
// iterations, is a caller error (the flag and counter arrays are sized by it).
int vm_iteration_cap(float max_iter, int *cap)
{
    if (!std::isfinite(max_iter)) return vm_fail(VM_E_INVALID, "max_iter must be finite (got %g)", (double)max_iter);
    if (max_iter > (float)(1 << 20)) return vm_fail(VM_E_INVALID, "max_iter %g exceeds the limit of %d iterations per level", (double)max_iter, 1 << 20);
    *cap = std::max(1, (int)std::ceil(max_iter));
    return VM_OK;
}

// The same level (one page) of n frame pairs -- or n pages of a video that do not depend on
// each other -- relaxed by the same launches.
int vm_optimize_levels(vm_ctx *c, vm_level **lv, int n, float max_iter, volatile const int *run_flag,
                       int fixed_work, vm_progress *out)
{
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    VM_ON_DEVICE(c);
    vm_level &l0 = *lv[0];
    for (int i = 0; i < n; ++i) {
        vm_level &l = *lv[i];
        if (l.w != l0.w || l.h != l0.h) return vm_fail(VM_E_INVALID, "batch: pyramids must share their geometry");
        if (!l.has_state) return vm_fail(VM_E_STATE, "vm_optimize_level: level not initialised");
        if ((l.view.temp_mask != nullptr) != (l0.view.temp_mask != nullptr))
            return vm_fail(VM_E_INVALID, "batch: pages with and without the temporal term cannot share a launch");
    }
    VmKParams P = {c->kp.w_ui, c->kp.w_tps, c->kp.w_ssim, c->kp.ssim_clamp, c->kp.eps, c->kp.bcond, c->kp.w_temp,
                   c->commit_order};
    int cap = 1;
    int rc0 = vm_iteration_cap(max_iter, &cap);
    if (rc0 != VM_OK) return rc0;
    const size_t words = (size_t)cap * n;
    if ((size_t)c->flags_cap < words) {
        hipFree(c->flags); hipHostFree(c->flags_host);
        hipFree(c->stats); hipHostFree(c->stats_host);
        c->flags = c->flags_host = c->stats = c->stats_host = nullptr;
        c->flags_cap = 0;
        VM_HIP(hipMalloc((void **)&c->flags, words * 4));
        VM_HIP(hipHostMalloc((void **)&c->flags_host, words * 4, hipHostMallocDefault));
        VM_HIP(hipMalloc((void **)&c->stats, words * 4 * VM_STAT_WORDS));
        VM_HIP(hipHostMalloc((void **)&c->stats_host, words * 4 * VM_STAT_WORDS, hipHostMallocDefault));
        c->flags_cap = (int)words;
    }
    if (n > c->views_cap) {
        hipFree(c->views);
        c->views = nullptr;
        c->views_cap = 0;
        VM_HIP(hipMalloc((void **)&c->views, (size_t)n * sizeof(VmLevelView)));
        c->views_cap = n;
    }
    hipStream_t s = c->stream;
    const int ntiles_lvl = ((l0.w + VM_PITCH_X - 1) / VM_PITCH_X) * ((l0.h + VM_PITCH_Y - 1) / VM_PITCH_Y);
    // SPARSE: one workgroup per pair walks the few active tiles of a pruned level on the device
    const bool may_sparse = (c->sweep_mode == VM_SWEEP_AUTO || c->sweep_mode == VM_SWEEP_SPARSE) && ntiles_lvl <= 8192;
    {
        // levels that may run the SPLIT / STEP schedules need their workspace before the views
        // are copied to the device
        const int tiles0 = ((l0.w + VM_PITCH_X - 1) / VM_PITCH_X) * ((l0.h + VM_PITCH_Y - 1) / VM_PITCH_Y);
        if (c->sweep_mode == VM_SWEEP_SPLIT || c->sweep_mode == VM_SWEEP_STEP || c->sweep_mode == VM_SWEEP_PASS ||
            (c->sweep_mode == VM_SWEEP_AUTO && tiles0 * n <= vm_step_max_tiles()))
            for (int i = 0; i < n; ++i) {
                int rc = level_ensure_ws(c, *lv[i]);
                if (rc != VM_OK) return rc;
            }
        if (may_sparse)
            for (int i = 0; i < n; ++i) {
                int rc = level_ensure_sparse(c, *lv[i]);
                if (rc != VM_OK) return rc;
                // the stamps are epochs of THIS call (iteration * 4 + pass + 1)
                VM_HIP(hipMemsetAsync(lv[i]->view.sp_stamp, 0, (size_t)l0.imp_rs * l0.imp_rows * 4, s));
            }
        std::vector<VmLevelView> hv(n);
        for (int i = 0; i < n; ++i) hv[i] = lv[i]->view;
        VM_HIP(hipMemcpyAsync(c->views, hv.data(), (size_t)n * sizeof(VmLevelView), hipMemcpyHostToDevice, s));
        VM_HIP(hipStreamSynchronize(s)); // hv is a stack object
    }
    VM_HIP(hipMemsetAsync(c->flags, 0, words * 4, s));
    VM_HIP(hipMemsetAsync(c->stats, 0, words * 4 * VM_STAT_WORDS, s));
    const bool exact = c->math_mode != VM_MATH_FAST; // EXACT and its FMA-contracted diagnostic build
    const SweepLaunchers &SL = sweep_launchers(c->math_mode);
    // FAST kernels are built for at most 512 threads (256-VGPR budget: the register-cached
    // window sums must not spill), EXACT ones for up to 1024
    const int threads = std::min(c->sweep_threads ? c->sweep_threads : 512, exact ? 1024 : 512);
    const int tiles_per_pass = ((l0.w + VM_PITCH_X - 1) / VM_PITCH_X) * ((l0.h + VM_PITCH_Y - 1) / VM_PITCH_Y);
    // (see SmallDensePresence) this call's share of the small-level dense workgroups on the device, while it lasts
    // -- registered only while the call's CURRENT batch launches such workgroups (dense TILE sweeps): a call that
    // runs PASS, STEP, SPARSE or pruned lean batches has none in flight and must not make another context believe
    // it has company (measured there: 256-thread workgroups without a partner cost 16 %)
    SmallDensePresence small_dense;
    static const bool no_corun = getenv("VM_NO_CORUN") != nullptr; // dev switch
    const bool small_dense_ok = !exact && !no_corun && tiles_per_pass <= 32 && c->sweep_threads == 0;
    // (k_tile_scan) the listed form of pruned TILE passes: from VM_TILE_LIST_MIN workgroups per pass on, tiles that fit the
    // entries' 16 bits; counters and stamps start from zero in every call (the epochs do)
    static const bool no_list = getenv("VM_NO_TILE_LIST") != nullptr; // dev switch
    const bool listed_ok = !exact && !no_list && (c->sweep_mode == VM_SWEEP_AUTO || c->sweep_mode == VM_SWEEP_TILE) &&
                           (size_t)tiles_per_pass * n >= (size_t)(c->sweep_mode == VM_SWEEP_TILE && c->sweep_parts ? c->sweep_parts : VM_TILE_LIST_MIN) &&
                           tiles_per_pass <= 65535 && n <= 65535;
    if (listed_ok) {
        const size_t need = 4 * (size_t)cap + 2 * (size_t)tiles_per_pass * n; // counters per iteration and pass, stamps, entries
        if (c->tile_list_words < need) {
            VM_HIP(hipStreamSynchronize(s));
            hipFree(c->tile_list);
            c->tile_list = nullptr;
            c->tile_list_words = 0;
            VM_HIP(hipMalloc((void **)&c->tile_list, need * sizeof(uint32_t)));
            c->tile_list_words = need;
        }
        VM_HIP(hipMemsetAsync(c->tile_list, 0, (4 * (size_t)cap + (size_t)tiles_per_pass * n) * sizeof(uint32_t), s));
    }
    // SPLIT / STEP schedules: workgroups per tile (every candidate gets 32 lanes, 16 candidates
    // per 512-thread workgroup)
    // (16 workgroups of 16 candidates per tile while the chip has room for them; 8 of 32 when a
    // phase-step of the batch would otherwise need more than two full waves of workgroups --
    // measured on 8 x 120x68: 232 -> 217 ms per level; 4 x 64 is slower again)
    // 32 on the smallest levels: with <= 8 candidates per workgroup k_step gives every
    // candidate a whole wave and its line search takes two steps per round (decide64) --
    // 120x68: 107 -> 98.5 ms per 500 iterations; 240x135 (28 tiles) is better off at 16.
    const int parts = c->sweep_parts ? c->sweep_parts
                                     : (tiles_per_pass * n * 16 >= 1024 ? (tiles_per_pass * n > 64 ? VM_STEP_BIG_PARTS : 8) : (tiles_per_pass * n <= 12 ? 32 : 16));
    // Schedule, re-decided per batch of iterations (AUTO).  TILE: 4 launches per iteration, a
    // tile's four phases inside one workgroup -- unbeatable when a pass touches nothing (24 us
    // per converged iteration) or when there are enough tiles to fill the chip.  STEP (SPLIT
    // when forced): a tile's line searches spread over `parts` workgroups, 16 (32) launches
    // per iteration -- measured on MI355X (FAST, 1080p pyramid): 120x68, every pixel active,
    // 0.32 (STEP) vs 0.64 ms (TILE) per iteration; 240x135 with 900 line searches per
    // iteration 0.22 vs 0.36; with 90: 0.24 vs 0.23; converged 0.08 vs 0.024.  All schedules
    // work on the same state in HBM, so the choice can change from batch to batch.
    // (r03: restricting levels of more than 12 tiles per pass -- 240x135 -- to single pairs helped two
    // streams x 2 independent pairs, 274 -> 206 ms per job, and cost the coupled 5-frame video, whose
    // chain steps are batches of two pages, 354 -> 417 ms: not done)
    const bool may_split = c->sweep_mode == VM_SWEEP_AUTO && tiles_per_pass * n <= vm_step_max_tiles();
    double cand_prev = 1e9; // line searches per iteration in the previous batch (first batch: dense)
    double tiles_prev = 1e9; // active tile visits per iteration and pair in the previous batch
    if (may_split || c->sweep_mode == VM_SWEEP_SPLIT || c->sweep_mode == VM_SWEEP_STEP || c->sweep_mode == VM_SWEEP_PASS) // epochs restart with every call: forget old records
        for (int i = 0; i < n; ++i)
        {
            VM_HIP(hipMemsetAsync(lv[i]->view.rec_tag, 0, (size_t)l0.rs * l0.h * 4, s));
            VM_HIP(hipMemsetAsync(lv[i]->view.rec_tag2, 0, (size_t)l0.rs * l0.h * 4, s));
        }
    // PASS: the workgroups of a tile group spin at a barrier of their own, so every group of a
    // launch must become resident whatever else runs.  One 256-workgroup chunk (8 groups) always
    // fits an idle MI355X; two PASS launches at once could starve each other's groups, so a
    // device-wide token (PassToken: across contexts AND processes) admits one holder at a time --
    // the others run STEP -- and the barrier's spin is bounded: in AUTO a timeout (a device whose
    // compute units are masked or otherwise not all ours) puts the level back to where the batch
    // started, reruns the batch with STEP and keeps this context off PASS from then on; only a
    // FORCED PASS schedule reports it as VM_E_DEVICE.  VM_NO_PASS=1 (environment) turns PASS off.
    static const bool no_pass = getenv("VM_NO_PASS") != nullptr;
    PassToken pass_token;
    bool want_pass = !no_pass && (c->sweep_mode == VM_SWEEP_PASS ||
                                  (c->sweep_mode == VM_SWEEP_AUTO && !c->pass_latched_off && tiles_per_pass * n <= VM_PASS_MAX_GROUPS));
    // k_pass addresses a level's arrays by 32-bit byte offsets from its slab and from its schedule
    // workspace (72 + 104 B per pixel)
    if ((size_t)l0.rs * l0.h * 128 >= ((size_t)1 << 32)) {
        if (c->sweep_mode == VM_SWEEP_PASS)
            return vm_fail(VM_E_STATE, "vm_optimize_level: the PASS schedule addresses levels of up to 32 Mpixel");
        want_pass = false;
    }
    if (want_pass) { // a 256-workgroup chunk of the launch must fit the device at once
        int &res = c->pass_resident[c->math_mode & 7];
        if (res < 0) res = SL.pass_resident(c->device);
        if (res < 256) {
            if (c->sweep_mode == VM_SWEEP_PASS)
                return vm_fail(VM_E_STATE, "vm_optimize_level: the PASS schedule needs 256 co-resident workgroups, this device holds %d", res);
            want_pass = false;
        }
    }
    if (want_pass && pass_token.try_acquire(c->device) && !c->pass_err) {
        VM_HIP(hipMalloc((void **)&c->pass_err, 256));
        VM_HIP(hipMemsetAsync(c->pass_err, 0, 256, s));
        VM_HIP(hipHostMalloc((void **)&c->pass_err_host, 256, hipHostMallocDefault));
    }
    bool may_pass = want_pass && pass_token.owns;
    // diagnostic forms of a FORCED PASS schedule (vm_set_tuning(VM_SWEEP_PASS, 0, parts)): parts == 1 stores
    // write-through from the start, parts == 2 maps 32 consecutive workgroup ids to a tile group, so that
    // every group spans all XCDs and takes the census -> write-back -> write-through route for real
    const int pass_switches = (c->sweep_mode == VM_SWEEP_PASS && c->sweep_parts == 1 ? 1 : 0) | (c->pass_test_timeout ? 2 : 0) |
                              (c->sweep_mode == VM_SWEEP_PASS && c->sweep_parts == 2 ? 4 : 0);
    // AUTO only: the levels as they stand before a PASS batch, to rerun it with STEP should a barrier time out
    const bool pass_guard = may_pass && c->sweep_mode == VM_SWEEP_AUTO;
    if (pass_guard) {
        size_t need = 0;
        for (int i = 0; i < n; ++i) need += lv[i]->slab_bytes;
        if (c->pass_snap_bytes < need) {
            hipFree(c->pass_snap);
            c->pass_snap = nullptr;
            c->pass_snap_bytes = 0;
            VM_HIP(hipMalloc(&c->pass_snap, need));
            c->pass_snap_bytes = need;
        }
    }
    const int offs[4][2] = {{0, 0}, {VM_TILE_W, 0}, {0, VM_TILE_H}, {VM_TILE_W, VM_TILE_H}}; // morph.cu:1382-1385
    std::vector<int> executed(n, cap), improving(n, 1), stopped(n, 0), live(n, -1);
    std::vector<double> st_tiles(n, 0.0), st_cand(n, 0.0), st_commit(n, 0.0), st_eval(n, 0.0);
    int done = 0, launches = 0;
    bool cancelled = false;
    float ms = 0;
    float sched_ms[5] = {0, 0, 0, 0, 0}; // [0] TILE dense kernel, [1] TILE lean kernel, [2] STEP / SPLIT, [3] SPARSE, [4] PASS
    double clk_shader[2] = {0, 0}, clk_wall[2] = {0, 0}; // in-kernel clock probe: [0] dense TILE kernel, [1] k_pass
    int sched_launches[5] = {0, 0, 0, 0, 0};
    static const bool force_dense = getenv("VM_TILE_DENSE") != nullptr; // dev switch
    // Iterations are enqueued in batches; each sweep kernel of iteration i exits at once (per
    // pair) when iteration i-1 did not improve (device-side flag), so running past convergence
    // inside a batch costs launch latency only, and the host reads the flags once per batch
    // instead of once per iteration.  The HIP events bracket the sweep launches of each batch
    // on the context's stream.
    int batch = 2; // a short first batch: the schedule of the rest depends on what it finds
    while (done < cap) {
        const int nb = std::min(batch, cap - done);
        const bool split = c->sweep_mode == VM_SWEEP_SPLIT || c->sweep_mode == VM_SWEEP_STEP || c->sweep_mode == VM_SWEEP_PASS ||
                           (may_split && cand_prev >= vm_step_min_cand() * n);
        // one launch per pass (PASS) where it is admitted, else one per phase (STEP), unless
        // the two-kernel SPLIT is forced
        const bool pass = split && may_pass;
        const bool step = split && !pass && c->sweep_mode != VM_SWEEP_SPLIT;
        // TILE, FAST arithmetic: the register-light kernel variant once fewer than a tenth of the pixels
        // are searched per iteration (after the first sweep of a level, typically)
        const bool lean_regime = cand_prev < 0.1 * l0.w * l0.h * n;
        // dense sweeps, FAST: the 128-VGPR form of the dense kernel (>= 4 lanes per candidate, two
        // workgroups per CU) on levels of >= 256 tiles per pass (960x540 and up) -- measured on MI355X
        // (r03, tools/dev_dense.py, us per dense pass, 256- vs 128-VGPR kernel, with the taps shared by
        // lane pairs): 1080p x 1 pair 1399 vs 1290; 960x540 x 1 468 vs 466, x 8 2711 vs 2362, x 30 9946 vs
        // 8272; but 480x270 x 1 228 vs 306, x 8 726 vs 798 (x 30 2509 vs 2254), 240x135 x 30 875 vs 960,
        // 120x68 x 30 277 vs 438: with about one workgroup per CU the second round of a 256-candidate
        // phase costs more than the second workgroup hides.  The rule looks at the level only, never
        // at the batch: a pair is solved by the same kernels alone and in a batch (FAST sums are
        // ordered by the lane fan-out).  VM_DENSE128=0 / 1 forces it (dev switch).
        static const char *d128 = getenv("VM_DENSE128");
        const bool dense128 = !exact && (d128 ? atoi(d128) != 0 : tiles_per_pass >= 256);
        const int dense = (exact || force_dense || !lean_regime) ? (dense128 ? 2 : 1) : 0;
        // SPARSE replaces the TILE launches of a pruned level once at most three tiles per pass
        // and pair are still active (measured on MI355X, 1080p: a no-op TILE iteration costs
        // 4 x 3.4 us, a no-op SPARSE iteration 4 x ~0.3 us; with more active tiles than that the
        // one workgroup per pair serialises what the TILE grid runs side by side)
        const bool sparse = may_sparse && !split && lean_regime && !force_dense &&
                            (c->sweep_mode == VM_SWEEP_SPARSE || tiles_prev <= (double)vm_sparse_tiles());
        const int sched = pass ? 4 : (split ? 2 : (sparse ? 3 : (dense ? 0 : 1)));
        const int launches_before = launches;
        VM_HIP(hipEventRecord(c->ev0, s));
        uint32_t last_epoch = 0;
        int sb = 0; // step index inside this batch: parity = which copy of the sums is read
        int slot_iter = -1; // iteration whose counts the previous STEP launch left in its slots
        const int pass_groups = tiles_per_pass * n, pass_blocks = (pass_groups + 7) / 8 * 256;
        if (pass && pass_guard) {
            size_t off = 0;
            for (int i = 0; i < n; ++i) {
                VM_HIP(hipMemcpyAsync((char *)c->pass_snap + off, lv[i]->slab, lv[i]->slab_bytes, hipMemcpyDeviceToDevice, s));
                off += lv[i]->slab_bytes;
            }
        }
        if (pass) { // barrier counters of every launch of the batch, zeroed once
            const size_t need_bar = (size_t)nb * 4 * pass_groups * VM_PASS_SYNC_WORDS;
            if (c->pass_bar_words < need_bar) {
                VM_HIP(hipStreamSynchronize(s));
                hipFree(c->pass_bar);
                c->pass_bar = nullptr;
                c->pass_bar_words = 0;
                VM_HIP(hipMalloc((void **)&c->pass_bar, std::max(need_bar, (size_t)64 * 4 * 8 * VM_PASS_SYNC_WORDS) * sizeof(uint32_t)));
                c->pass_bar_words = std::max(need_bar, (size_t)64 * 4 * 8 * VM_PASS_SYNC_WORDS);
            }
            VM_HIP(hipMemsetAsync(c->pass_bar, 0, need_bar * sizeof(uint32_t), s));
        }
        if (step || pass) {
            // per-workgroup count slots of the last two launches (k_step / k_pass fold them one launch late)
            const int gxs = (l0.w + VM_PITCH_X - 1) / VM_PITCH_X, gys = (l0.h + VM_PITCH_Y - 1) / VM_PITCH_Y;
            const size_t need = pass ? (size_t)pass_blocks * 4 : (size_t)gxs * gys * parts * n * 4;
            if (c->step_slots_words < need) {
                VM_HIP(hipStreamSynchronize(s));
                hipFree(c->step_slots);
                c->step_slots = nullptr;
                c->step_slots_words = 0;
                VM_HIP(hipMalloc((void **)&c->step_slots, 2 * need * sizeof(uint32_t)));
                c->step_slots_words = need;
            }
        }
        int it0 = done;
        if (sparse) {
            for (int i = 0; i < n; ++i)
                VM_HIP(hipMemsetAsync(lv[i]->view.sp_cnt, 0, 8, s));
            // (forced SPARSE schedule with parts given: the LDS capacity of the word list, 0 < parts; parts = 1 is
            // "as good as none": the list then lives in memory from the first pass it holds two words -- tests)
            SL.sparse(
                c->views, n, cap, l0.w, l0.h, P, c->tables, c->flags, c->stats, done, nb, fixed_work, threads, dense,
                c->sweep_mode == VM_SWEEP_SPARSE && c->sweep_parts > 0 ? c->sweep_parts : 1 << 20, c->sparse_resident, s);
            launches += 2;
            it0 = done + nb;
        }
        // dense TILE sweeps of a small level as 256-thread workgroups when enough of them are in flight on the
        // device to pair up on the CUs (SmallDensePresence)
        if (small_dense_ok && dense == 1 && !split && !sparse)
            small_dense.enter(c->device, tiles_per_pass * n);
        else
            small_dense.leave();
        const int tile_threads = small_dense.in_flight() >= VM_CORUN_MIN_WGS ? 256 : threads;
        // pruned TILE passes of a big batch: the listed form (k_tile_scan) -- dispatching tiles x pairs workgroups that
        // find nothing costs ~4.7 ns each, 118 us per pass over 30 1080p pairs
        uint32_t *const tile_list = (listed_ok && dense == 0 && !split && !sparse) ? c->tile_list : nullptr;
        if (!split && !sparse && nb >= VM_GRAPH_ITERS) {
            // TILE batch: whole groups of VM_GRAPH_ITERS iterations are graph replays
            if (hipGraphExec_t ge = sweep_graph(c, c->math_mode, n, l0.w, l0.h, cap, fixed_work, tile_threads, dense, tile_list, P)) {
                SL.next_iter(c->iter_dev, 1, done, s);
                for (; it0 + VM_GRAPH_ITERS <= done + nb; it0 += VM_GRAPH_ITERS) {
                    VM_HIP(hipGraphLaunch(ge, s));
                    launches += 4 * VM_GRAPH_ITERS;
                }
            }
        }
        for (int it = it0; it < done + nb; ++it)
            for (int k = 0; k < 4; ++k) {
                if (pass) {
                    SL.pass(
                        c->views, n, cap, l0.w, l0.h, P, c->tables, offs[k][0], offs[k][1], 1u + (uint32_t)((it * 4 + k) * 4),
                        c->pass_bar + (size_t)((it - done) * 4 + k) * pass_groups * VM_PASS_SYNC_WORDS, c->flags, c->stats, it, fixed_work,
                        c->step_slots + (size_t)(sb & 1) * c->step_slots_words,
                        c->step_slots + (size_t)((sb + 1) & 1) * c->step_slots_words, sb == 0 ? -1 : slot_iter, c->pass_err,
                        c->pass_dbg, 1, pass_switches, s);
                    slot_iter = it;
                    ++sb;
                    ++launches;
                } else if (step) {
                    for (int ph = 0; ph < 4; ++ph, ++sb) {
                        const uint32_t epoch = 1u + (uint32_t)((it * 4 + k) * 4 + ph);
                        SL.step(
                            c->views, n, cap, l0.w, l0.h, P, c->tables, offs[k][0], offs[k][1], ph >> 1, ph & 1, epoch,
                            sb == 0 ? 0u : epoch - 1u, sb & 1, 1, c->flags, c->stats, it, fixed_work, threads, parts,
                            c->step_slots + (size_t)(sb & 1) * c->step_slots_words,
                            c->step_slots + (size_t)((sb + 1) & 1) * c->step_slots_words, sb == 0 ? -1 : slot_iter, s);
                        slot_iter = it;
                        last_epoch = epoch;
                    }
                    launches += 4;
                } else if (split) {
                    SL.split(c->views, n, cap, l0.w, l0.h, P, c->tables, offs[k][0], offs[k][1], k, c->flags, c->stats, it, fixed_work, threads, parts, s);
                    launches += 8;
                } else {
                    SL.optimize(c->views, n, cap, l0.w, l0.h, P, c->tables, offs[k][0], offs[k][1], c->flags, c->stats, it, fixed_work, tile_threads, nullptr, dense, tile_list, s);
                    ++launches;
                }
            }
        if (step) { // fold the last phase's records in place: copy 0 is complete again
            SL.step(
                c->views, n, cap, l0.w, l0.h, P, c->tables, 0, 0, 0, 0, 0u, last_epoch, 2, 0, c->flags, c->stats,
                done + nb - 1, fixed_work, threads, parts, c->step_slots + (size_t)(sb & 1) * c->step_slots_words,
                c->step_slots + (size_t)((sb + 1) & 1) * c->step_slots_words, sb == 0 ? -1 : slot_iter, s);
            ++launches;
        }
        if (pass && sb > 0) { // the counts the last launch left in its slots
            SL.pass(
                c->views, n, cap, l0.w, l0.h, P, c->tables, 0, 0, 0u, nullptr, c->flags, c->stats, done + nb - 1, fixed_work,
                nullptr, c->step_slots + (size_t)((sb + 1) & 1) * c->step_slots_words, slot_iter, c->pass_err, nullptr, 0, pass_switches, s);
            ++launches;
        }
        VM_HIP(hipEventRecord(c->ev1, s));
        VM_HIP(hipGetLastError());
        if (pass)
            VM_HIP(hipMemcpyAsync(c->pass_err_host, c->pass_err, 4, hipMemcpyDeviceToHost, s));
        // (one strided copy per array instead of 2 n small ones was measured: 60 pairs 838 -> 836 ms, 8 pairs 328 -> 332: not kept)
        for (int i = 0; i < n; ++i) {
            VM_HIP(hipMemcpyAsync(c->flags_host + (size_t)i * cap + done, c->flags + (size_t)i * cap + done, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
            VM_HIP(hipMemcpyAsync(c->stats_host + ((size_t)i * cap + done) * VM_STAT_WORDS, c->stats + ((size_t)i * cap + done) * VM_STAT_WORDS,
                                  (size_t)nb * 4 * VM_STAT_WORDS, hipMemcpyDeviceToHost, s));
        }
        VM_HIP(hipStreamSynchronize(s));
        if (pass && c->pass_err_host[0]) {
            VM_HIP(hipMemsetAsync(c->pass_err, 0, 4, s));
            if (!pass_guard)
                return vm_fail(VM_E_DEVICE, "vm_optimize_level: a tile barrier of the PASS schedule timed out (are all of this device's "
                                            "compute units available to this process?  VM_SWEEP_AUTO falls back to the STEP schedule by itself)");
            // AUTO: the batch never happened -- state, records, flags and counters as before it -- and runs again with STEP
            size_t off = 0;
            for (int i = 0; i < n; ++i) {
                VM_HIP(hipMemcpyAsync(lv[i]->slab, (char *)c->pass_snap + off, lv[i]->slab_bytes, hipMemcpyDeviceToDevice, s));
                off += lv[i]->slab_bytes;
                VM_HIP(hipMemsetAsync(lv[i]->view.rec_tag, 0, (size_t)l0.rs * l0.h * 4, s));
                VM_HIP(hipMemsetAsync(lv[i]->view.rec_tag2, 0, (size_t)l0.rs * l0.h * 4, s));
                VM_HIP(hipMemsetAsync(c->flags + (size_t)i * cap + done, 0, (size_t)nb * 4, s));
                VM_HIP(hipMemsetAsync(c->stats + ((size_t)i * cap + done) * VM_STAT_WORDS, 0, (size_t)nb * 4 * VM_STAT_WORDS, s));
            }
            launches = launches_before;
            may_pass = false;
            if (!c->pass_latched_off) c->pass_latched_by_test = c->pass_test_timeout != 0;
            c->pass_latched_off = true;
            ++c->pass_fallbacks;
            pass_token.release();
            continue;
        }
        float bms = 0;
        VM_HIP(hipEventElapsedTime(&bms, c->ev0, c->ev1));
        ms += bms;
        sched_ms[sched] += bms;
        sched_launches[sched] += launches - launches_before;
        bool all_stopped = true;
        double b_cand = 0, b_tiles = 0;
        for (int i = 0; i < n; ++i) {
            const uint32_t *fl = c->flags_host + (size_t)i * cap, *st = c->stats_host + (size_t)i * cap * VM_STAT_WORDS;
            for (int it = done; it < done + nb && !stopped[i]; ++it) {
                // [0] tile visits (TILE schedule), [3] tile-phases with records (SPLIT schedule)
                st_tiles[i] += st[VM_STAT_WORDS * it] + 0.25 * st[VM_STAT_WORDS * it + 3];
                b_tiles += st[VM_STAT_WORDS * it] + 0.25 * st[VM_STAT_WORDS * it + 3];
                st_cand[i] += st[VM_STAT_WORDS * it + 1];
                b_cand += st[VM_STAT_WORDS * it + 1];
                st_commit[i] += st[VM_STAT_WORDS * it + 2];
                st_eval[i] += st[VM_STAT_WORDS * it + 4];
                c->sparse_resident_visits += st[VM_STAT_WORDS * it + 5];
                if (i == 0 && (sched == 0 || sched == 4)) { // in-kernel clock probe of the dense TILE kernel / of k_pass (pair 0 only)
                    clk_shader[sched == 4] += st[VM_STAT_WORDS * it + 6];
                    clk_wall[sched == 4] += st[VM_STAT_WORDS * it + 7];
                }
                improving[i] = fl[it] != 0;
                if (!improving[i] && live[i] < 0) live[i] = it + 1; // the reference's loop ends here (morph.cu:1390)
                if (!improving[i] && !fixed_work) { executed[i] = it + 1; stopped[i] = 1; }
            }
            all_stopped = all_stopped && stopped[i];
        }
        cand_prev = b_cand / nb;
        tiles_prev = b_tiles / nb / n;
        done += nb;
        if (all_stopped) break;
        if (run_flag && !*run_flag) {
            for (int i = 0; i < n; ++i) if (!stopped[i]) executed[i] = done;
            cancelled = true;
            break;
        }
        batch = std::min(batch * 4, 64);
    }
    // everything this call wrote into the levels is enqueued: consumers on other streams wait on this event
    // (vm_frame_set_v_from_level across contexts) instead of draining this stream from the host
    VM_HIP(hipEventRecord(c->done_ev, s));
    for (int i = 0; i < n && out; ++i) {
        out[i].iters = executed[i];
        out[i].iters_live = live[i] < 0 ? executed[i] : std::min(live[i], executed[i]);
        out[i].improving = improving[i];
        out[i].pixel_iters = (double)executed[i] * l0.w * l0.h;
        out[i].elapsed_ms = ms;       // of the batch the pair was solved in
        out[i].launches = launches;   // idem
        out[i].active_tiles = st_tiles[i];
        out[i].candidates = st_cand[i];
        out[i].commits = st_commit[i];
        out[i].evaluations = st_eval[i];
        for (int k = 0; k < 5; ++k) { // of the batch, like elapsed_ms
            out[i].sched_ms[k] = sched_ms[k];
            out[i].sched_launches[k] = sched_launches[k];
        }
        for (int k = 0; k < 2; ++k) {
            out[i].clk_shader_ticks[k] = clk_shader[k];
            out[i].clk_wall_ticks[k] = clk_wall[k];
        }
    }
    return cancelled ? vm_fail(VM_E_CANCELLED, "vm_optimize_level: cancelled by run_flag") : VM_OK;
}

extern "C" int vm_optimize_level(vm_pyr *p, int lvl, float max_iter, volatile const int *run_flag,
                                 int fixed_work, vm_progress *out)
{
    CHECK_LVL(p, lvl);
    return optimize_level_batch(&p, 1, lvl, max_iter, run_flag, fixed_work, out);
}

extern "C" int vm_optimize_level_batch(vm_pyr **pyrs, int n, int lvl, float max_iter,
                                       volatile const int *run_flag, int fixed_work, vm_progress *out)
{
    if (!pyrs || n < 1 || !pyrs[0]) return vm_fail(VM_E_INVALID, "vm_optimize_level_batch: empty batch");
    return optimize_level_batch(pyrs, n, lvl, max_iter, run_flag, fixed_work, out);
}

// Morph::calculate_halfway_parametrization for a batch of pairs in lockstep; cons[i] / ncons[i] = pair i's own
// user constraints (cons == NULL: none anywhere)
extern "C" int vm_solve_batch_cons(vm_pyr **pyrs, int n, float max_iter, float drop, const vm_constraint *const *cons,
                                   const int *ncons, volatile const int *run_flag, int fixed_work, vm_progress *per_level)
{
    if (!pyrs || n < 1 || !pyrs[0]) return vm_fail(VM_E_INVALID, "vm_solve_batch: empty batch");
    if (!(drop > 0)) return vm_fail(VM_E_INVALID, "vm_solve_batch: max_iter_drop_factor must be > 0");
    if (cons && !ncons) return vm_fail(VM_E_INVALID, "vm_solve_batch_cons: constraints without their counts");
    std::lock_guard<std::recursive_mutex> lock(pyrs[0]->ctx->mu);
    const int L = (int)pyrs[0]->lv.size();
    const int w0 = pyrs[0]->lv[0].w, h0 = pyrs[0]->lv[0].h;
    int rc;
    for (int i = 0; i < n; ++i) {
        if (!pyrs[i] || (int)pyrs[i]->lv.size() != L) return vm_fail(VM_E_INVALID, "vm_solve_batch: pyramids must share their geometry");
        if (cons && (ncons[i] < 0 || (ncons[i] > 0 && !cons[i]))) return vm_fail(VM_E_INVALID, "vm_solve_batch_cons: constraints of pair %d", i);
        if ((rc = vm_coarse_solve(pyrs[i], L - 1, w0, h0, cons ? cons[i] : nullptr, cons ? ncons[i] : 0)) != VM_OK) return rc;
    }
    float mi = max_iter;
    std::vector<vm_progress> pr(n);
    for (int el = L - 2; el >= 0; --el) {
        if (run_flag && !*run_flag) return vm_fail(VM_E_CANCELLED, "vm_solve_batch: cancelled by run_flag");
        for (int i = 0; i < n; ++i) {
            if ((rc = vm_upsample_v(pyrs[i], el, el + 1)) != VM_OK) return rc;
            if ((rc = vm_init_level(pyrs[i], el, w0, h0, cons ? cons[i] : nullptr, cons ? ncons[i] : 0)) != VM_OK) return rc;
        }
        if ((rc = optimize_level_batch(pyrs, n, el, mi, run_flag, fixed_work, pr.data())) != VM_OK) return rc;
        if (per_level)
            for (int i = 0; i < n; ++i) per_level[(size_t)i * (L - 1) + el] = pr[i];
        mi /= drop;
    }
    return VM_OK;
}

extern "C" int vm_solve_batch(vm_pyr **pyrs, int n, float max_iter, float drop, volatile const int *run_flag,
                              int fixed_work, vm_progress *per_level)
{
    return vm_solve_batch_cons(pyrs, n, max_iter, drop, nullptr, nullptr, run_flag, fixed_work, per_level);
}

extern "C" int vm_solve(vm_pyr *p, float max_iter, float drop, const vm_constraint *cons, int n,
                        volatile const int *run_flag, int fixed_work, vm_progress *per_level)
{
    if (!p) return vm_fail(VM_E_INVALID, "vm_solve: pyramid is NULL");
    std::lock_guard<std::recursive_mutex> lock(p->ctx->mu);
    if (!(drop > 0)) return vm_fail(VM_E_INVALID, "vm_solve: max_iter_drop_factor must be > 0");
    const int L = (int)p->lv.size();
    const int w0 = p->lv[0].w, h0 = p->lv[0].h;
    // Morph::calculate_halfway_parametrization, morph.cu:150-168
    int rc = vm_coarse_solve(p, L - 1, w0, h0, cons, n);
    if (rc != VM_OK) return rc;
    float mi = max_iter;
    for (int el = L - 2; el >= 0; --el) {
        if (run_flag && !*run_flag) return vm_fail(VM_E_CANCELLED, "vm_solve: cancelled by run_flag");
        if ((rc = vm_upsample_v(p, el, el + 1)) != VM_OK) return rc;
        if ((rc = vm_init_level(p, el, w0, h0, cons, n)) != VM_OK) return rc;
        if ((rc = vm_optimize_level(p, el, mi, run_flag, fixed_work, per_level ? &per_level[el] : nullptr)) != VM_OK) return rc;
        // clear_level: the level's state stays allocated; v is kept for the result
        mi /= drop;
    }
    return VM_OK;
}
