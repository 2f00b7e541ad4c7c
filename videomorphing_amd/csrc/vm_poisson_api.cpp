// vm_poisson_api.cpp -- C-ABI of the Poisson boundary extension
// (CPoissonExt::run body for one side, Algorithm/PoissonExt.cpp:19-41) and the
// RCCL broadcast helper.
#include "vm_host.h"
#include "vm_mgb.h"
#include "vm_poisson.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <utility>
#include <vector>

namespace {

size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// grid sizes: halve (rounding up) down to a grid of at most VM_MGB_COARSEST cells
std::vector<std::pair<int, int>> mg_sizes(int w, int h)
{
    std::vector<std::pair<int, int>> v{{w, h}};
    while ((size_t)v.back().first * v.back().second > VM_MGB_COARSEST && (int)v.size() < VM_MGB_MAXLEV)
        v.push_back({(v.back().first + 1) / 2, (v.back().second + 1) / 2});
    return v;
}

// the first level of the cycle's one-workgroup tail: from there on all iterates fit VM_MGB_TAIL_X cells of LDS and all
// right-hand sides but the first VM_MGB_TAIL_B
int mg_tail_level(const std::vector<std::pair<int, int>> &sz)
{
    size_t below = 0;       // cells of the levels after l
    int l = (int)sz.size() - 1;
    while (l > 0) {
        const size_t here = (size_t)sz[l].first * sz[l].second, up = (size_t)sz[l - 1].first * sz[l - 1].second;
        if (below + here + up > VM_MGB_TAIL_X || below + here > VM_MGB_TAIL_B ||
            (size_t)((sz[l - 1].first + 1) / 2) * sz[l - 1].second > VM_MGB_TAIL_PAIRS)
            break;
        below += here;
        --l;
    }
    return l;
}

// red-black sweeps each way on level l of the cycle (VmMgbLevel::nu), by kind of system: VM_MGB_NU_POISSON / _QPATH
// (vm_mgb.h: measured choices).  VM_MGB_NU = "a,b,c" overrides both for experiments: sweeps per level from level 0 on, the
// last entry repeats; 1 or 2 on the levels the tile kernels sweep (larger values are cut to 2 there), 1 .. 9 inside the
// one-workgroup tail
int mg_nu(int l, bool in_tail, bool qpath)
{
    static const std::vector<int> env = [] {
        std::vector<int> t;
        if (const char *e = getenv("VM_MGB_NU"))
            for (const char *q = e; *q; ++q)
                if (*q >= '1' && *q <= '9') t.push_back(*q - '0');
        return t;
    }();
    static const std::vector<int> poisson{VM_MGB_NU_POISSON}, path{VM_MGB_NU_QPATH};
    const std::vector<int> &table = !env.empty() ? env : qpath ? path : poisson;
    const int nu = table[std::min((size_t)l, table.size() - 1)];
    return in_tail ? nu : std::min(nu, 2);
}

} // namespace

// ---------------------------------------------------------------------------
// The solver: multigrid-preconditioned CG, batched over systems (a system = one side of one frame), swept over the
// ring of unknowns only, with the fused kernels of vm_mgb.hip.

namespace {

struct MgbWork {          // one system's device workspace, carved from f->pws2[side - 1]
    VmMgbSys S;           // host copy of the device descriptor
    uint8_t *type;
    char *xcoarse;        // the x arrays of levels >= 1, contiguous (cleared per extension)
    size_t xcoarse_bytes;
    VmV3 *Xbest;          // optional (mgb_carve qpath): the iterate with the smallest residual seen near the tolerance
    int *counts;          // nblocks per level, then ntiles per level (device)
    int tail;             // first level of the cycle's one-workgroup tail
    VmV3 *r1;             // second buffer of the PCG residual, for ...
    bool fused;           // ... the PCG update riding in the level-0 restriction (the residual ping-pongs between S.R[0] and S.R[1]):
                          // can this system's hierarchy do it (mgb_carve); whether a solve does: mgb_solve
};

size_t mgb_bytes(int w, int h, bool with_best = false)
{
    const auto sz = mg_sizes(w, h);
    const size_t N0 = (size_t)w * h;
    size_t need = 2 * al256(N0) + al256(sizeof(VmMgbScalars)) + al256(2 * VM_MGB_MAXLEV * sizeof(int)) + 5 * al256(N0 * 12);
    for (size_t l = 0; l < sz.size(); ++l) {
        const size_t N = (size_t)sz[l].first * sz[l].second;
        const size_t nb = (size_t)((sz[l].first + 63) / 64) * ((sz[l].second + 3) / 4);
        need += (l ? 4 * al256(N * 4) : 0) + 2 * al256(N * 12) + al256((N + 1) / 2 * 12) + 3 * al256(nb * 4);
    }
    return need + (with_best ? al256(N0 * 12) : 0);
}

void mgb_carve(MgbWork &W, int w, int h, char *b, bool qpath = false)
{
    const auto sz = mg_sizes(w, h);
    const size_t N0 = (size_t)w * h;
    W.type = (uint8_t *)b; b += al256(N0);
    W.S.type = W.type;
    W.S.sc = (VmMgbScalars *)b; b += al256(sizeof(VmMgbScalars));
    W.counts = (int *)b; b += al256(2 * VM_MGB_MAXLEV * sizeof(int));
    W.tail = mg_tail_level(sz);
    W.S.X = (VmV3 *)b; b += al256(N0 * 12);
    W.S.P[0] = (VmV3 *)b; b += al256(N0 * 12);
    W.S.P[1] = (VmV3 *)b; b += al256(N0 * 12);
    W.S.Q = (VmV3 *)b; b += al256(N0 * 12);
    VmV3 *const r1 = (VmV3 *)b; b += al256(N0 * 12);
    W.S.nlev = (int)sz.size();
    for (size_t l = 0; l < sz.size(); ++l) {
        VmMgbLevel &L = W.S.lv[l];
        L.w = sz[l].first; L.h = sz[l].second;
        L.gx = (L.w + 63) / 64; L.gy = (L.h + 3) / 4;
        const size_t N = (size_t)L.w * L.h, nb = (size_t)L.gx * L.gy;
        L.info = nullptr;
        L.we = L.ws = L.dg = L.k = nullptr;
        if (l == 0) {                       // one byte of operator per cell
            L.info = (uint8_t *)b; b += al256(N);
        } else {
            L.we = (float *)b; b += al256(N * 4);
            L.ws = (float *)b; b += al256(N * 4);
            L.dg = (float *)b; b += al256(N * 4);
            L.k = (float *)b; b += al256(N * 4);
        }
        L.b = (VmV3 *)b; b += al256(N * 12);
        L.nu = mg_nu((int)l, (int)l >= W.tail, qpath);
        L.xr = (VmV3 *)b; b += al256((N + 1) / 2 * 12);
        L.flags = (uint32_t *)b; b += al256(nb * 4);
        L.blocks = (uint32_t *)b; b += al256(nb * 4);
        L.nblocks = W.counts + l;
        L.tiles = (uint32_t *)b; b += al256(nb * 4);
        L.ntiles = W.counts + VM_MGB_MAXLEV + l;
    }
    // the x arrays last and together: level 0's (z), then the coarse ones, which are cleared per extension (a
    // fine cell may read the correction of a coarse cell that is no unknown and sits in a block nobody sweeps)
    W.S.lv[0].x = (VmV3 *)b; b += al256(N0 * 12);
    W.xcoarse = b;
    for (size_t l = 1; l < sz.size(); ++l) {
        W.S.lv[l].x = (VmV3 *)b;
        b += al256((size_t)sz[l].first * sz[l].second * 12);
    }
    W.xcoarse_bytes = (size_t)(b - W.xcoarse);
    W.Xbest = qpath ? (VmV3 *)b : nullptr;     // (the quadratic path keeps the best iterate seen: mgb_solve)
    // The PCG update can ride in the level-0 restriction wherever that kernel exists in its one-sweep form: level 0 swept by
    // the tile kernels (not inside the tail) with one sweep each way
    W.fused = W.tail > 0 && W.S.lv[0].nu == 1;
    W.r1 = r1;
    W.S.R[0] = W.S.R[1] = W.S.lv[0].b;
}

// z = M^-1 r of every active system: one V cycle (sweeps per level: VmMgbLevel::nu); iteration k's r.z lands in rz[k & 1].
// nb / nt: blocks / tiles per level (the largest count among the systems).  In two halves, because the residual norm
// of iteration k - 1 comes out of the FIRST kernel of iteration k when the update rides in the level-0 restriction:
//   mgb_iter_head(k): the PCG update of iteration k - 1 (k >= 1) -- inside the level-0 restriction of cycle k (fused), or
//                     k_mgb_update by itself -- after which x, r and r.r of k completed iterations stand in memory;
//   mgb_iter_rest(k): the rest of cycle k and p = z + beta p, q = A p.
void mgb_iter_head(const VmMgbSys *dev, int nsys, bool fused, const std::vector<int> &nb, const std::vector<int> &nt, int k, uint64_t active,
                   hipStream_t s)
{
    if (fused)
        vm_mgb_launch_restrict(dev, nsys, 0, 1, nt[0], k, k > 0, active, s);
    else if (k > 0)
        vm_mgb_launch_update(dev, nsys, nb[0], k - 1, active, s);
}

void mgb_iter_rest(const VmMgbSys *dev, int nsys, const MgbWork &W0, bool fused, const std::vector<int> &nb, const std::vector<int> &nt, int k,
                   uint64_t active, hipStream_t s)
{
    const int tail = W0.tail;       // levels tail .. nlev - 1 run in one workgroup
    for (int l = fused ? 1 : 0; l < tail; ++l)
        vm_mgb_launch_restrict(dev, nsys, l, W0.S.lv[l].nu, nt[l], k, false, active, s);
    vm_mgb_launch_tail(dev, nsys, tail, active, s);
    if (tail == 0)
        vm_mgb_launch_dot_rz(dev, nsys, nb[0], k, active, s);
    for (int l = tail - 1; l >= 0; --l)
        vm_mgb_launch_prolong(dev, nsys, l, W0.S.lv[l].nu, nt[l], k, active, s);
    vm_mgb_launch_dirspmv(dev, nsys, nb[0], k, active, s);
}

double mgb_rel(const VmMgbScalars &h, int par)
{
    double worst = 0;
    for (int c = 0; c < 3; ++c) {
        double bb = 0, rr = 0;
        for (int k = 0; k < VM_MGB_SLOTS; ++k) { bb += h.bb[k][c]; rr += h.rr[par][k][c]; }
        if (!(bb == bb) || !(rr == rr) || std::isinf(bb) || std::isinf(rr)) return -1;
        if (bb > 0) worst = std::max(worst, std::sqrt(rr / bb));
    }
    return worst;
}

} // namespace

// make sure *ws holds the batched solver's workspace of a w x h system
static int mgb_reserve(void **ws, size_t *ws_bytes, int w, int h, size_t at_least = 0)
{
    const size_t need = std::max(mgb_bytes(w, h), at_least);
    if (*ws_bytes < need) {
        hipFree(*ws);
        *ws = nullptr;
        *ws_bytes = 0;
        VM_HIP(hipMalloc(ws, need));
        *ws_bytes = need;
    }
    return VM_OK;
}

// The batched PCG proper: nsys systems of one size whose workspaces are carved, whose type maps, right-hand sides
// (lv[0].b) and initial guesses (X) are enqueued on the context's stream.  Leaves every system's solution in its X
// (the iterate it stopped at; the best one seen near the tolerance if the workspace was carved with room for it), its
// iteration count and relative residual in iters / rels.
static int mgb_solve(vm_ctx *c, std::vector<MgbWork> &W, int nsys, float tol, int max_it, int *iters, double *rels)
{
    hipStream_t s = c->stream;
    const size_t N0 = (size_t)W[0].S.lv[0].w * W[0].S.lv[0].h;
    if (!c->mgb_sys) VM_HIP(hipMalloc((void **)&c->mgb_sys, VM_MGB_MAXSYS * sizeof(VmMgbSys)));
    // the systems' PCG scalars and block / tile counts live side by side in one buffer of the context (the descriptors
    // handed to the kernels point there): one clear per solve, one read-back per residual check for the whole batch
    // instead of one per system
    const size_t cnt_bytes = (size_t)VM_MGB_MAXSYS * 2 * VM_MGB_MAXLEV * sizeof(int);
    if (!c->mgb_shared) VM_HIP(hipMalloc(&c->mgb_shared, VM_MGB_MAXSYS * sizeof(VmMgbScalars) + cnt_bytes));
    VmMgbScalars *sc_dev = (VmMgbScalars *)c->mgb_shared;
    int *cnt_dev = (int *)((char *)c->mgb_shared + VM_MGB_MAXSYS * sizeof(VmMgbScalars));
    VmMgbSys *dev = (VmMgbSys *)c->mgb_sys;
    std::vector<VmMgbSys> hs(nsys);
    const int nlev = W[0].S.nlev;
    // The PCG update rides in the level-0 restriction wherever the hierarchy allows it.  Measured on the 2304 x 1464 canvas
    // (tools/exp/fuse_ab.sh, ms per frame at 1e-5, fused against the separate k_mgb_update): 8 systems per batch 1.61 / 1.70,
    // 4 systems 1.95 / 1.99, 2 systems 2.49 / 2.51, one system 1.78 / 1.81 per side (with the fused kernel's loads issued cell
    // by cell it lost on one and two systems, 2.56 / 2.49: vm_mgb.hip).  Same arithmetic either way.
    // VM_MGB_FUSE_MIN_SYS (dev switch): the smallest batch that fuses (0: never).
    static const int fuse_min = [] { const char *e = getenv("VM_MGB_FUSE_MIN_SYS"); return e ? atoi(e) : 1; }();
    const bool fused = W[0].fused && fuse_min > 0 && nsys >= fuse_min;
    for (int i = 0; i < nsys; ++i) {
        W[i].S.R[1] = fused ? W[i].r1 : W[i].S.R[0];
        hs[i] = W[i].S;
        hs[i].sc = sc_dev + i;
        for (int l = 0; l < nlev; ++l) {
            hs[i].lv[l].nblocks = cnt_dev + (size_t)i * 2 * VM_MGB_MAXLEV + l;
            hs[i].lv[l].ntiles = cnt_dev + (size_t)i * 2 * VM_MGB_MAXLEV + VM_MGB_MAXLEV + l;
        }
    }
    VM_HIP(hipMemcpyAsync(dev, hs.data(), nsys * sizeof(VmMgbSys), hipMemcpyHostToDevice, s));
    VM_HIP(hipMemsetAsync(sc_dev, 0, nsys * sizeof(VmMgbScalars), s));
    for (int i = 0; i < nsys; ++i)
        if (W[i].xcoarse_bytes) VM_HIP(hipMemsetAsync(W[i].xcoarse, 0, W[i].xcoarse_bytes, s));
    // the hierarchy and its block lists (batched)
    vm_mgb_launch_level0(dev, nsys, W[0].S.lv[0].gx, W[0].S.lv[0].gy, s);
    for (int l = 1; l < nlev; ++l)
        vm_mgb_launch_coarsen(dev, nsys, l, W[0].S.lv[l].gx, W[0].S.lv[l].gy, s);
    vm_mgb_launch_compact(dev, nsys, nlev, s);
    VM_HIP(hipGetLastError());
    std::vector<int> cnt((size_t)nsys * 2 * VM_MGB_MAXLEV), nb(nlev, 0), nt(nlev, 0);
    VM_HIP(hipMemcpyAsync(cnt.data(), cnt_dev, (size_t)nsys * 2 * VM_MGB_MAXLEV * sizeof(int), hipMemcpyDeviceToHost, s));
    VM_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < nsys; ++i)
        for (int l = 0; l < nlev; ++l) {
            nb[l] = std::max(nb[l], cnt[(size_t)i * 2 * VM_MGB_MAXLEV + l]);
            nt[l] = std::max(nt[l], cnt[(size_t)i * 2 * VM_MGB_MAXLEV + VM_MGB_MAXLEV + l]);
        }
    if (nb[0] == 0) {                       // no unknown anywhere: nothing to extend
        for (int i = 0; i < nsys; ++i) { iters[i] = 0; rels[i] = 0; }
        return VM_OK;
    }
    uint64_t active = nsys == 64 ? ~0ull : ((1ull << nsys) - 1);
    vm_mgb_launch_init(dev, nsys, nb[0], active, s);
    std::vector<VmMgbScalars> h(nsys);
    std::vector<double> best(nsys, 1e300);
    std::vector<int> best_it(nsys, 0), next_check(nsys, 0), saved(nsys, 0);
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
    std::vector<int> prof_sys;
    int it = 0;
    // A system's residual is looked at every 4 iterations (a read drains the stream) until it is within a factor 30
    // of the tolerance -- the cycle gains a decade in two to three iterations -- and every iteration from there: a solve
    // stops at the iteration that reaches the tolerance instead of up to three later.  The cadence is the SYSTEM's own
    // (next_check): where it stops, and so what it pastes, does not depend on its batch-mates.
    // check(it): the systems due after `it` completed iterations (mgb_iter_head(it) has been enqueued: x, r, r.r are theirs)
    auto check = [&](int it) -> int {
        int lo = nsys, hi = -1;          // the systems looked at now: one read-back of the span that holds them
        for (int i = 0; i < nsys; ++i)
            if (((active >> i) & 1) && next_check[i] == it) { lo = std::min(lo, i); hi = i; }
        if (hi < lo) return VM_OK;
        VM_HIP(hipMemcpyAsync(&h[lo], sc_dev + lo, (size_t)(hi - lo + 1) * sizeof(VmMgbScalars), hipMemcpyDeviceToHost, s));
        VM_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < nsys; ++i) {
            if (!((active >> i) & 1) || next_check[i] != it) continue;
            const double worst = mgb_rel(h[i], (it - 1) & 1);   // it == 0: parity 1, where k_mgb_init left r.r
            if (worst < 0)
                return vm_fail(VM_E_NUMERIC, it == 0 ? "multigrid PCG: the right-hand side is not finite" : "multigrid PCG broke down (NaN)");
            if (worst < best[i]) {
                best[i] = worst;
                best_it[i] = it;
                // A system with room for it (the quadratic path: float32 attains 1e-4 .. 1e-5 there, the recursively updated
                // residual passes below what the stored iterate attains and the iteration then drifts) keeps the best
                // iterate seen at a check near the tolerance
                if (W[i].Xbest && worst <= 30.0 * tol) {
                    VM_HIP(hipMemcpyAsync(W[i].Xbest, W[i].S.X, N0 * sizeof(VmV3), hipMemcpyDeviceToDevice, s));
                    saved[i] = 1;
                }
            }
            // a system stops when it reaches the tolerance -- or gives up: no better residual for 12 iterations, a
            // residual 1000 times the best one seen, max_it.  It then holds its best iterate if it kept one, else its
            // CURRENT iterate, and reports that iterate's residual (the callers turn a residual above the tolerance into
            // VM_E_NUMERIC)
            if (worst <= tol || it >= max_it || it - best_it[i] >= 12 || worst > 1e3 * best[i]) {
                active &= ~(1ull << i);
                if (saved[i] && best_it[i] != it) {
                    VM_HIP(hipMemcpyAsync(W[i].S.X, W[i].Xbest, N0 * sizeof(VmV3), hipMemcpyDeviceToDevice, s));
                } else {
                    best[i] = worst;
                    best_it[i] = it;
                }
            }
            next_check[i] = std::min(max_it, it + (best[i] <= 30.0 * tol ? 1 : 4));
        }
        return VM_OK;
    };
    {
        const int rc = check(0);
        if (rc != VM_OK) return rc;
    }
    while (active) {
        if (c->mgb_prof && it > 0) {     // the probe of vm_dbg_poisson_profile: events around the launch that carries the update
            hipEvent_t e0 = nullptr, e1 = nullptr;
            VM_HIP(hipEventCreate(&e0));
            VM_HIP(hipEventCreate(&e1));
            VM_HIP(hipEventRecord(e0, s));
            mgb_iter_head(dev, nsys, fused, nb, nt, it, active, s);
            VM_HIP(hipEventRecord(e1, s));
            prof_ev.push_back({e0, e1});
            prof_sys.push_back(__builtin_popcountll(active));
        } else {
            mgb_iter_head(dev, nsys, fused, nb, nt, it, active, s);
        }
        if (it > 0) {
            const int rc = check(it);
            if (rc != VM_OK) return rc;
            if (!active) break;
        }
        mgb_iter_rest(dev, nsys, W[0], fused, nb, nt, it, active, s);
        ++it;
        VM_HIP(hipGetLastError());
    }
    for (int i = 0; i < nsys; ++i) {
        iters[i] = best_it[i];
        rels[i] = best[i];
    }
    if (!prof_ev.empty()) {
        VM_HIP(hipStreamSynchronize(s));
        for (size_t k = 0; k < prof_ev.size(); ++k) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, prof_ev[k].first, prof_ev[k].second) == hipSuccess) {
                c->mgb_prof_us += 1e3 * ms;
                c->mgb_prof_launches += 1;
                c->mgb_prof_fused += fused ? 1 : 0;
                c->mgb_prof_unknown_launches += prof_sys[k];        // active systems of that launch (x unknowns per system: the caller's)
            }
            hipEventDestroy(prof_ev[k].first);
            hipEventDestroy(prof_ev[k].second);
        }
    }
    return VM_OK;
}

// Diagnostic (bench.py's roofline of the compositor's HBM-bound kernel, measured live as the contract asks: HIP events on
// the stream the kernel is launched on): on != 0 arms the probe and clears its sums; on == 0 disarms it and returns the
// HIP-event time of the launches that carried the PCG update since (microseconds, summed), their number, the number of
// ACTIVE systems summed over those launches, and how many of them were the level-0 restriction with the update fused in
// (k_mgb_restrict<true, true>: 76 B per unknown of every active system) rather than k_mgb_update by itself (73 B).
extern "C" int vm_dbg_poisson_profile(vm_ctx *c, int on, double *update_us, int *update_launches, double *active_systems, int *fused_launches)
{
    if (!c) return vm_fail(VM_E_INVALID, "vm_dbg_poisson_profile: ctx is NULL");
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (on) {
        c->mgb_prof = true;
        c->mgb_prof_us = c->mgb_prof_unknown_launches = 0;
        c->mgb_prof_launches = c->mgb_prof_fused = 0;
        return VM_OK;
    }
    c->mgb_prof = false;
    if (update_us) *update_us = c->mgb_prof_us;
    if (update_launches) *update_launches = c->mgb_prof_launches;
    if (active_systems) *active_systems = c->mgb_prof_unknown_launches;
    if (fused_launches) *fused_launches = c->mgb_prof_fused;
    return VM_OK;
}

// Poisson extension of nsys systems (frames[i], sides[i]) of one context and one canvas size as ONE batch
static int poisson_solve_batch(vm_ctx *c, vm_frame *const *frames, const int *sides, int nsys, float tol, int max_it,
                               int *iters, double *rels)
{
    hipStream_t s = c->stream;
    const int cw = frames[0]->cw, ch = frames[0]->ch;
    std::vector<MgbWork> W(nsys);
    // classify, fill, right-hand side + initial guess (per system)
    for (int i = 0; i < nsys; ++i) {
        vm_frame *f = frames[i];
        const int side = sides[i];
        int rc = mgb_reserve(&f->pws2[side - 1], &f->pws2_bytes[side - 1], cw, ch);
        if (rc != VM_OK) return rc;
        mgb_carve(W[i], cw, ch, (char *)f->pws2[side - 1]);
        uchar4 *ext = f->ext[side - 1];
        const uchar4 *other = f->crop[side == 1 ? 1 : 0]; // PoissonExt.cpp:54-57
        vm_poisson_launch_prepare(ext, W[i].type, other, f->v, f->w, f->h, f->rs, f->ex, side == 1 ? 1 : -1, s);
        vm_poisson_launch_setup3(ext, W[i].type, W[i].S.lv[0].b, W[i].S.X, cw, ch, s);
    }
    int rc = mgb_solve(c, W, nsys, tol, max_it, iters, rels);
    if (rc != VM_OK) return rc;
    for (int i = 0; i < nsys; ++i)
        vm_poisson_launch_paste3(frames[i]->ext[sides[i] - 1], W[i].type, W[i].S.X, cw, ch, s);
    VM_HIP(hipGetLastError());
    return VM_OK;
}

extern "C" int vm_poisson_extend(vm_frame *f, int side, float tol, int max_it, int *iters,
                                 float *rel_res, float *elapsed_ms)
{
    if (!f || (side != 1 && side != 2) || !(tol > 0) || max_it < 1)
        return vm_fail(VM_E_INVALID, "vm_poisson_extend: bad argument");
    vm_ctx *c = f->ctx;
    VM_ON_DEVICE(c);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    hipStream_t s = c->stream;
    VM_HIP(hipEventRecord(c->ev0, s));
    int total_it = 0;
    double rel = 0;
    int rc = poisson_solve_batch(c, &f, &side, 1, tol, max_it, &total_it, &rel);
    if (rc != VM_OK) return rc;
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

// Both sides of n frames in one batch: CPoissonExt::run's loop body (PoissonExt.cpp:24-36) for n frames at once --
// side 1 samples the ORIGINAL image 2 and side 2 the original image 1 (the crops taken at upload), so the 2 n
// systems are independent.  iters / rel_res: 2 n entries, [2 i] = side 1 of frame i, [2 i + 1] = side 2.
extern "C" int vm_poisson_extend_frames(vm_frame *const *frames, int n, float tol, int max_it, int *iters,
                                        float *rel_res, float *elapsed_ms)
{
    if (!frames || n < 1 || 2 * n > VM_MGB_MAXSYS || !(tol > 0) || max_it < 1)
        return vm_fail(VM_E_INVALID, "vm_poisson_extend_frames: bad argument (1 <= n <= %d)", VM_MGB_MAXSYS / 2);
    for (int i = 0; i < n; ++i) {
        if (!frames[i]) return vm_fail(VM_E_INVALID, "vm_poisson_extend_frames: frame %d is NULL", i);
        if (frames[i]->ctx != frames[0]->ctx) return vm_fail(VM_E_INVALID, "vm_poisson_extend_frames: the frames belong to different contexts");
        if (frames[i]->cw != frames[0]->cw || frames[i]->ch != frames[0]->ch)
            return vm_fail(VM_E_INVALID, "vm_poisson_extend_frames: the frames differ in size");
        for (int j = 0; j < i; ++j)
            if (frames[j] == frames[i]) return vm_fail(VM_E_INVALID, "vm_poisson_extend_frames: frame %d listed twice", i);
    }
    vm_ctx *c = frames[0]->ctx;
    if (!vm_ctx_alive(c)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__);
    VM_ON_DEVICE(c);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    hipStream_t s = c->stream;
    std::vector<vm_frame *> fr(2 * n);
    std::vector<int> sd(2 * n), its(2 * n, 0);
    std::vector<double> rel(2 * n, 0.0);
    for (int i = 0; i < n; ++i) { fr[2 * i] = fr[2 * i + 1] = frames[i]; sd[2 * i] = 1; sd[2 * i + 1] = 2; }
    VM_HIP(hipEventRecord(c->ev0, s));
    int rc = poisson_solve_batch(c, fr.data(), sd.data(), 2 * n, tol, max_it, its.data(), rel.data());
    if (rc != VM_OK) return rc;
    VM_HIP(hipEventRecord(c->ev1, s));
    VM_HIP(hipEventSynchronize(c->ev1));
    float ms = 0;
    VM_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    double worst = 0;
    int at = 0;
    for (int i = 0; i < 2 * n; ++i) {
        if (iters) iters[i] = its[i];
        if (rel_res) rel_res[i] = (float)rel[i];
        if (rel[i] > worst) { worst = rel[i]; at = i; }
    }
    if (worst > tol)
        return vm_fail(VM_E_NUMERIC, "vm_poisson_extend_frames: residual %.3g after %d iterations on side %d of frame %d (tol %.3g)",
                       worst, its[at], at % 2 + 1, at / 2, (double)tol);
    return VM_OK;
}

// CQuadraticPath::optimize for the frame's halfway field (QuadraticPath.cpp:24-223): u goes
// to the frame's quadratic-path buffer, where vm_render_halfway reads it
extern "C" int vm_frame_quadratic_path(vm_frame *f, float tol, int max_it, int *iters, float *rel_res,
                                       float *elapsed_ms)
{
    if (!f || !(tol > 0) || max_it < 1)
        return vm_fail(VM_E_INVALID, "vm_frame_quadratic_path: bad argument");
    if (f->w < 2 || f->h < 2)
        return vm_fail(VM_E_INVALID, "vm_frame_quadratic_path: needs a frame of at least 2x2 pixels");
    vm_ctx *c = f->ctx;
    VM_ON_DEVICE(c);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    hipStream_t s = c->stream;
    int it = 0;
    double rel = 0;
    VM_HIP(hipEventRecord(c->ev0, s));
    {
        // the batched solver on the whole grid: every pixel an unknown without a tie (type 2 everywhere), so the level-0
        // operator is the graph Laplacian of the pixel grid with Neumann ends (QuadraticPath.cpp:137-170); the
        // workspace is side 1's of the Poisson extension (the frame is no larger than its canvas, the two run in turn)
        int rc = mgb_reserve(&f->pws2[0], &f->pws2_bytes[0], std::max(f->w, f->cw), std::max(f->h, f->ch), mgb_bytes(f->w, f->h, true));
        if (rc != VM_OK) return rc;
        std::vector<MgbWork> W(1);
        mgb_carve(W[0], f->w, f->h, (char *)f->pws2[0], true);
        VM_HIP(hipMemsetAsync(W[0].type, 2, (size_t)f->w * f->h, s));
        vm_qpath_launch_rhs3(f->v, f->rs, f->w, f->h, W[0].S.lv[0].b, W[0].S.X, s);
        // project the right-hand side onto the range of the singular operator
        double *sums = &W[0].S.sc->bb[0][0];
        VM_HIP(hipMemsetAsync(W[0].S.sc, 0, sizeof(VmMgbScalars), s));
        vm_qpath_launch_sum3(W[0].S.lv[0].b, f->w, f->h, sums, s);
        vm_qpath_launch_shift3(W[0].S.lv[0].b, f->w, f->h, sums, nullptr, 0, s);
        VM_HIP(hipGetLastError());
        rc = mgb_solve(c, W, 1, tol, max_it, &it, &rel);
        if (rc != VM_OK) return rc;
        VM_HIP(hipMemsetAsync(W[0].S.sc, 0, sizeof(VmMgbScalars), s));
        vm_qpath_launch_sum3(W[0].S.X, f->w, f->h, sums, s);
        vm_qpath_launch_shift3(W[0].S.X, f->w, f->h, sums, f->u, f->rs, s);
    }
    f->u_zero = false;
    VM_HIP(hipGetLastError());
    VM_HIP(hipEventRecord(c->ev1, s));
    VM_HIP(hipEventSynchronize(c->ev1));
    float ms = 0;
    VM_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (iters) *iters = it;
    if (rel_res) *rel_res = (float)rel;
    if (elapsed_ms) *elapsed_ms = ms;
    if (rel > tol)
        return vm_fail(VM_E_NUMERIC, "vm_frame_quadratic_path: residual %.3g after %d iterations (tol %.3g)", rel, it, (double)tol);
    return VM_OK;
}

// the frame's quadratic path, tight (h, w, 2) floats
extern "C" int vm_frame_download_qpath(vm_frame *f, float *u_xy)
{
    if (!f || !u_xy) return vm_fail(VM_E_INVALID, "vm_frame_download_qpath: bad argument");
    if (!vm_ctx_alive(f->ctx)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__);
    VM_ON_DEVICE(f->ctx);
    hipStream_t s = f->ctx->stream;
    VM_HIP(hipMemcpy2DAsync(u_xy, (size_t)f->w * 8, f->u, (size_t)f->rs * 8, (size_t)f->w * 8, f->h, hipMemcpyDeviceToHost, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

// the frame's halfway field as the compositor holds it, tight (h, w, 2) floats (_vector[frame] of
// the reference's Pyramid once update_result has run)
extern "C" int vm_frame_download_v(vm_frame *f, float *v_xy)
{
    if (!f || !v_xy) return vm_fail(VM_E_INVALID, "vm_frame_download_v: bad argument");
    if (!vm_ctx_alive(f->ctx)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__);
    VM_ON_DEVICE(f->ctx);
    hipStream_t s = f->ctx->stream;
    VM_HIP(hipMemcpy2DAsync(v_xy, (size_t)f->w * 8, f->v, (size_t)f->rs * 8, (size_t)f->w * 8, f->h, hipMemcpyDeviceToHost, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

// RCCL is resolved at first use so that the library loads (and the CPU-side
// tests run) on hosts without a usable librccl.
namespace {
struct Rccl {
    typedef int (*bcast_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    typedef int (*initall_fn)(void **, int, const int *);
    typedef int (*destroy_fn)(void *);
    typedef int (*group_fn)(void);
    bcast_fn bcast = nullptr;
    initall_fn init_all = nullptr;
    destroy_fn destroy = nullptr;
    group_fn group_start = nullptr, group_end = nullptr;
    bool tried = false;
};
Rccl &rccl()
{
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!r.tried) {
        r.tried = true;
        // an RCCL the process already holds first (a host framework's own copy: RCCL and the HIP runtime must come
        // from ONE ROCm installation), then the system's
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            r.bcast = (Rccl::bcast_fn)dlsym(h, "ncclBroadcast");
            r.init_all = (Rccl::initall_fn)dlsym(h, "ncclCommInitAll");
            r.destroy = (Rccl::destroy_fn)dlsym(h, "ncclCommDestroy");
            r.group_start = (Rccl::group_fn)dlsym(h, "ncclGroupStart");
            r.group_end = (Rccl::group_fn)dlsym(h, "ncclGroupEnd");
        }
    }
    return r;
}
const int kNcclInt8 = 0; // ncclDataType_t: ncclInt8 / ncclChar
} // namespace

extern "C" int vm_rccl_bcast(vm_ctx *c, void *comm, void *dev_buf, uint64_t bytes, int root)
{
    if (!c || !comm || !dev_buf) return vm_fail(VM_E_INVALID, "vm_rccl_bcast: NULL argument");
    VM_ON_DEVICE(c);
    Rccl &R = rccl();
    if (!R.bcast) return vm_fail(VM_E_DEVICE, "vm_rccl_bcast: cannot load librccl / ncclBroadcast");
    int rc = R.bcast(dev_buf, dev_buf, (size_t)bytes, kNcclInt8, root, comm, c->stream);
    if (rc != 0) return vm_fail(VM_E_DEVICE, "vm_rccl_bcast: ncclBroadcast returned %d", rc);
    return VM_OK;
}

// ncclCommInitAll: one communicator per device of ONE process (the C++ multi-device driver, examples/solve_shard.cpp)
extern "C" int vm_rccl_comm_init_all(int n, const int *devices, void **comms_out)
{
    if (n < 1 || !devices || !comms_out) return vm_fail(VM_E_INVALID, "vm_rccl_comm_init_all: bad argument");
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j])
                return vm_fail(VM_E_INVALID, "vm_rccl_comm_init_all: device %d listed twice (RCCL wants one rank per device; contexts "
                                             "that share a device pass comms = NULL to vm_bcast_params)", devices[i]);
    Rccl &R = rccl();
    if (!R.init_all) return vm_fail(VM_E_DEVICE, "vm_rccl_comm_init_all: cannot load librccl / ncclCommInitAll");
    (void)hipGetLastError();
    int rc = R.init_all(comms_out, n, devices);
    (void)hipGetLastError();           // RCCL leaves stale errors behind when it probes peers
    if (rc != 0) return vm_fail(VM_E_DEVICE, "vm_rccl_comm_init_all: ncclCommInitAll returned %d", rc);
    return VM_OK;
}

extern "C" void vm_rccl_comm_destroy(void *comm)
{
    Rccl &R = rccl();
    if (comm && R.destroy) R.destroy(comm);
}

// One process, n contexts: a byte payload goes root -> every context (SURVEY 8(e)'s "exactly one ncclBroadcast of the
// shared parameter block"; for config[4] the block is followed by the frames' point constraints, so the payload's length
// is the caller's).
//   comms != NULL: comms[i] = the ncclComm_t of ctxs[i] (vm_rccl_comm_init_all); the payload is staged in a device
//                  buffer per context and broadcast inside one ncclGroup over xGMI, each on its context's stream;
//   comms == NULL: TEST MODE for contexts that share a device (RCCL refuses two ranks on one device): the root's
//                  device buffer is copied device-to-device into the others'.
// Every context then reads ITS device copy back into dst_host[i] (bytes each): what travelled, not what was sent.
extern "C" int vm_bcast_bytes(vm_ctx *const *ctxs, void *const *comms, int n, int root, const void *src, uint64_t bytes,
                              void *const *dst_host)
{
    if (!ctxs || n < 1 || root < 0 || root >= n || !src || bytes == 0 || !dst_host) return vm_fail(VM_E_INVALID, "vm_bcast_bytes: bad argument");
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || !dst_host[i] || (comms && !comms[i])) return vm_fail(VM_E_INVALID, "vm_bcast_bytes: context / buffer / communicator %d is NULL", i);
    Rccl &R = rccl();
    if (comms && (!R.bcast || !R.group_start || !R.group_end)) return vm_fail(VM_E_DEVICE, "vm_bcast_bytes: cannot load librccl");
    std::vector<void *> buf(n, nullptr);
    int rc = VM_OK;
    auto fail = [&](int code, const char *what) { rc = vm_fail(code, "vm_bcast_bytes: %s", what); };
    for (int i = 0; i < n && rc == VM_OK; ++i) {
        VmDeviceGuard g(ctxs[i]->device);
        if (!g.ok || hipMalloc(&buf[i], (size_t)bytes) != hipSuccess) { fail(VM_E_DEVICE, "device buffer"); break; }
        // everybody but the root starts from zeros: what it ends up with is what travelled
        hipError_t e = i == root ? hipMemcpyAsync(buf[i], src, (size_t)bytes, hipMemcpyHostToDevice, ctxs[i]->stream)
                                 : hipMemsetAsync(buf[i], 0, (size_t)bytes, ctxs[i]->stream);
        if (e == hipSuccess && i == root) e = hipStreamSynchronize(ctxs[i]->stream);      // src belongs to the caller
        if (e != hipSuccess) fail(VM_E_DEVICE, hipGetErrorString(e));
    }
    if (rc == VM_OK && comms) {
        if (R.group_start() != 0) fail(VM_E_DEVICE, "ncclGroupStart");
        for (int i = 0; i < n && rc == VM_OK; ++i) {
            VmDeviceGuard g(ctxs[i]->device);
            if (!g.ok || R.bcast(buf[i], buf[i], (size_t)bytes, kNcclInt8, root, comms[i], ctxs[i]->stream) != 0)
                fail(VM_E_DEVICE, "ncclBroadcast");
        }
        if (R.group_end() != 0 && rc == VM_OK) fail(VM_E_DEVICE, "ncclGroupEnd");
    } else if (rc == VM_OK) {
        VmDeviceGuard g(ctxs[root]->device);
        if (hipStreamSynchronize(ctxs[root]->stream) != hipSuccess) fail(VM_E_DEVICE, "sync");
        for (int i = 0; i < n && rc == VM_OK; ++i)
            if (i != root && hipMemcpyAsync(buf[i], buf[root], (size_t)bytes, hipMemcpyDeviceToDevice, ctxs[i]->stream) != hipSuccess)
                fail(VM_E_DEVICE, "device-to-device copy");
    }
    for (int i = 0; i < n && rc == VM_OK; ++i) {
        VmDeviceGuard g(ctxs[i]->device);
        if (hipMemcpyAsync(dst_host[i], buf[i], (size_t)bytes, hipMemcpyDeviceToHost, ctxs[i]->stream) != hipSuccess ||
            hipStreamSynchronize(ctxs[i]->stream) != hipSuccess) { fail(VM_E_DEVICE, "read-back"); break; }
    }
    for (int i = 0; i < n; ++i)
        if (buf[i]) {
            VmDeviceGuard g(ctxs[i]->device);
            hipFree(buf[i]);
        }
    (void)hipGetLastError();
    return rc;
}

// The shared parameter block root -> every context (vm_bcast_bytes), and each context adopts what IT received: kernel
// parameters + arithmetic mode.  blocks_out (n entries, may be NULL) receives the block as each context got it.
extern "C" int vm_bcast_params(vm_ctx *const *ctxs, void *const *comms, int n, int root, const vm_param_block *blk,
                               vm_param_block *blocks_out)
{
    if (!ctxs || n < 1 || root < 0 || root >= n || !blk) return vm_fail(VM_E_INVALID, "vm_bcast_params: bad argument");
    std::vector<vm_param_block> got(n);
    std::vector<void *> dst(n);
    for (int i = 0; i < n; ++i) dst[i] = &got[i];
    int rc = vm_bcast_bytes(ctxs, comms, n, root, blk, sizeof(vm_param_block), dst.data());
    for (int i = 0; i < n && rc == VM_OK; ++i) {
        if ((rc = vm_set_params(ctxs[i], &got[i].kp)) != VM_OK) break;
        if ((rc = vm_set_math_mode(ctxs[i], got[i].math_mode)) != VM_OK) break;
        if (blocks_out) blocks_out[i] = got[i];
    }
    return rc;
}
