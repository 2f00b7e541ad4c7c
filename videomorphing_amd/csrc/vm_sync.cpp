// vm_sync.cpp -- host side of the synchronisation stage: CSyncThread (Algorithm/SyncThread.cpp)
// and the layered arrays / renderer of stage 1 (pyramid.cu:57-165, render.cu:203-246,
// UI/RenderWidget.cpp:205-227).  Kernels: vm_sync.hip.
#include "vm_host.h"
#include "vm_sync.h"

#include <array>
#include <cmath>
#include <cstring>
#include <map>

struct vm_sync {
    vm_ctx *ctx = nullptr;
    std::vector<int> w, h, d;                 // per level; [0] = full resolution (placeholder)
    std::vector<vm_sync_constraint> cons;
    std::vector<std::array<float *, 3>> f;    // d_x[el], d_y[el], d_z[el]; null until the level is reached
    void *ws = nullptr;                       // CG workspace of the level being solved (kept between levels)
    size_t ws_bytes = 0;
    // stage-1 renderer: Pyramid::_video0/_video1/_forw0/_forw1 (Pyramid.h:46-47) and _vector[frame]
    uchar4 *video[2] = {nullptr, nullptr};
    float2 *forw[2] = {nullptr, nullptr};
    float4 *vec = nullptr;
    int vec_frame = -1;
    uint8_t *out = nullptr;
};

#define CHECK_SYNC(s)                                                              \
    if (!(s) || !(s)->ctx) return vm_fail(VM_E_INVALID, "%s: null handle", __func__);  \
    if (!vm_ctx_alive((s)->ctx)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__); \
    std::lock_guard<std::recursive_mutex> lock_((s)->ctx->mu);                     \
    VM_ON_DEVICE((s)->ctx)

// Pyramid::build(video0, video1, f0, f1, start_res), pyramid.cu:143-163 -- evaluated in float as
// there (`w /= decres_fa` on an int, the log2 template of pyramid.cu:50-55, ceil(w / 2.0f))
extern "C" int vm_sync_level_table(int w, int h, int d, int start_res, int *lw, int *lh, int *ld, int cap, int *n_out)
{
    if (w < 1 || h < 1 || d < 1 || start_res < 1 || !lw || !lh || !ld || cap < 1 || !n_out)
        return vm_fail(VM_E_INVALID, "vm_sync_level_table: bad arguments");
    if ((double)w * h * d > 2147483647.0) return vm_fail(VM_E_INVALID, "vm_sync_level_table: w * h * d overflows an int");
    int n = 0;
    auto push = [&](int a, int b, int c) {
        if (n < cap) { lw[n] = a; lh[n] = b; ld[n] = c; }
        ++n;
    };
    push(w, h, d);
    float fa = (float)(w * h * d) / (float)4000000; // Max_stage1, pyramid.cu:7
    const float root = std::sqrt(fa);
    fa = root > 1 ? root : 1;
    w = (int)((float)w / fa);
    h = (int)((float)h / fa);
    if (w < 1 || h < 1) return vm_fail(VM_E_INVALID, "vm_sync_level_table: the decimated level is empty");
    const float l2 = std::log(2.0f);
    const int el_t = 1;
    const int el_y = (int)(std::log((float)h) / l2 - std::log((float)start_res) / l2 + 1);
    const int el_x = (int)(std::log((float)w) / l2 - std::log((float)start_res) / l2 + 1);
    const int maxl = std::max(std::max(el_x, el_y), el_t);
    for (int el = 0; el < maxl; ++el) {
        push(w, h, d);
        if (maxl - el <= el_x) w = (int)std::ceil(w / 2.0f);
        if (maxl - el <= el_y) h = (int)std::ceil(h / 2.0f);
        if (maxl - el <= el_t) d = (int)std::ceil(d / 2.0f);
    }
    *n_out = n;
    return VM_OK;
}

extern "C" int vm_sync_create(vm_ctx *ctx, int nlevels, const int *w, const int *h, const int *d, vm_sync **out)
{
    if (!ctx || !out || !w || !h || !d || nlevels < 2) return vm_fail(VM_E_INVALID, "vm_sync_create: bad arguments");
    if (!vm_ctx_alive(ctx)) return vm_fail(VM_E_INVALID, "vm_sync_create: the context was destroyed");
    for (int l = 0; l < nlevels; ++l) {
        if (w[l] < 1 || h[l] < 1 || d[l] < 1) return vm_fail(VM_E_INVALID, "vm_sync_create: level %d is empty", l);
        if ((double)w[l] * h[l] * d[l] > 1.0e9) return vm_fail(VM_E_INVALID, "vm_sync_create: level %d is too large", l);
        if (d[l] != d[0]) return vm_fail(VM_E_INVALID, "vm_sync_create: the sync pyramid keeps every frame (level %d)", l);
    }
    vm_sync *s = new vm_sync();
    s->ctx = ctx;
    s->w.assign(w, w + nlevels);
    s->h.assign(h, h + nlevels);
    s->d.assign(d, d + nlevels);
    s->f.assign(nlevels, std::array<float *, 3>{nullptr, nullptr, nullptr});
    *out = s;
    return VM_OK;
}

// the three components of a level live in ONE allocation (x at the base): large allocations get
// large page fragments, which the plane-strided accesses of the CG kernels need
static void free_field(vm_sync *s, int lvl)
{
    if (s->f[lvl][0]) (void)hipFree(s->f[lvl][0]);
    s->f[lvl] = {nullptr, nullptr, nullptr};
}

extern "C" void vm_sync_destroy(vm_sync *s)
{
    if (!s) return;
    if (s->ctx && vm_ctx_alive(s->ctx)) {
        std::lock_guard<std::recursive_mutex> lock(s->ctx->mu);
        VM_ON_DEVICE_VOID(s->ctx);
        (void)hipStreamSynchronize(s->ctx->stream);
        for (size_t l = 0; l < s->f.size(); ++l) free_field(s, (int)l);
        if (s->ws) (void)hipFree(s->ws);
        for (int k = 0; k < 2; ++k) {
            if (s->video[k]) (void)hipFree(s->video[k]);
            if (s->forw[k]) (void)hipFree(s->forw[k]);
        }
        if (s->vec) (void)hipFree(s->vec);
        if (s->out) (void)hipFree(s->out);
    }
    delete s;
}

extern "C" int vm_sync_set_constraints(vm_sync *s, const vm_sync_constraint *c, int n)
{
    CHECK_SYNC(s);
    if (n < 0 || (n > 0 && !c)) return vm_fail(VM_E_INVALID, "vm_sync_set_constraints: bad arguments");
    s->cons.assign(c, c + n);
    return VM_OK;
}

#define CHECK_SLVL(s, lvl)                                                                          \
    if ((lvl) < 1 || (lvl) >= (int)(s)->w.size())                                                   \
        return vm_fail(VM_E_INVALID, "%s: level %d out of range (1..%d)", __func__, (lvl), (int)(s)->w.size() - 1)

static int alloc_field(vm_sync *s, int lvl)
{
    const size_t N = (size_t)s->w[lvl] * s->h[lvl] * s->d[lvl];
    if (!s->f[lvl][0]) {
        const size_t bytes = (3 * N * sizeof(float) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        float *base = nullptr;
        VM_HIP(hipMalloc((void **)&base, bytes));
        VM_HIP(hipMemsetAsync(base, 0, 3 * N * sizeof(float), s->ctx->stream));
        s->f[lvl] = {base, base + N, base + 2 * N};
    }
    return VM_OK;
}

extern "C" int vm_sync_load_identity(vm_sync *s, int lvl)
{
    CHECK_SYNC(s);
    CHECK_SLVL(s, lvl);
    free_field(s, lvl);
    return alloc_field(s, lvl);
}

// CSyncThread::upsample_level(el, el + 1), SyncThread.cpp:103-128: x and y scale with the size
// ratio, the frame displacement does not; the coarser field is released
extern "C" int vm_sync_upsample_level(vm_sync *s, int lvl)
{
    CHECK_SYNC(s);
    CHECK_SLVL(s, lvl);
    const int pel = lvl + 1;
    if (pel >= (int)s->w.size() || !s->f[pel][0]) return vm_fail(VM_E_STATE, "vm_sync_upsample_level: level %d holds no field", pel);
    free_field(s, lvl);
    if (int rc = alloc_field(s, lvl)) return rc;
    const float ratio[3] = {(float)s->w[lvl] / (float)s->w[pel], (float)s->h[lvl] / (float)s->h[pel], 1.0f};
    for (int c = 0; c < 3; ++c)
        vm_sync_launch_upsample(s->f[lvl][c], s->w[lvl], s->h[lvl], s->f[pel][c], s->w[pel], s->h[pel], ratio[c], s->d[lvl],
                                s->ctx->stream);
    VM_HIP(hipGetLastError());
    VM_HIP(hipStreamSynchronize(s->ctx->stream));
    free_field(s, pel);
    return VM_OK;
}

// The row pattern of genMatrix (SyncThread.cpp:190-262) depends on a coordinate only through
// p > 1, p > 0, p < n - 1, p < n - 2: positions with equal answers share a state (five border
// classes for n > 5, every position its own state for n <= 5).
static int state_rep(int s, int n) { return (n <= 5 || s <= 2) ? s : n - 5 + s; }

// Off-diagonal entries of the row of voxel (x, y, z), in CSR order (25 slots, slot 12 = the
// diagonal, left 0).  Every increment genMatrix adds to one off-diagonal slot has the same
// value, so a slot is that value added `count` times:
//   axis +-2   one second-difference operator      +2w        once
//   in-plane diagonal   one mixed 2x2 cell          +4w        once
//   axis +-1   the second-difference operators and mixed cells that hold both voxels, -4w each
static void offdiag_row(int x, int y, int z, int w, int h, int d, float wt, float *row25)
{
    static const int taps[25][3] = {{-2, 0, 0}, {-1, -1, 0}, {-1, 0, -1}, {-1, 0, 0}, {-1, 0, 1}, {-1, 1, 0}, {0, -2, 0},
                                    {0, -1, -1}, {0, -1, 0}, {0, -1, 1}, {0, 0, -2}, {0, 0, -1}, {0, 0, 0}, {0, 0, 1},
                                    {0, 0, 2}, {0, 1, -1}, {0, 1, 0}, {0, 1, 1}, {0, 2, 0}, {1, -1, 0}, {1, 0, -1},
                                    {1, 0, 0}, {1, 0, 1}, {1, 1, 0}, {2, 0, 0}};
    const int pos[3] = {z, y, x}, dim[3] = {d, h, w};
    const float far2 = 1.0f * 2.0f * wt, near4 = -2.0f * 2.0f * wt, diag4 = 2.0f * 2.0f * wt;
    for (int t = 0; t < 25; ++t) {
        row25[t] = 0.0f;
        if (t == 12) continue;
        int nz = 0, ax[2] = {0, 0};
        for (int a = 0; a < 3; ++a)
            if (taps[t][a]) ax[nz++] = a;
        bool inside = true;
        for (int a = 0; a < 3; ++a) {
            const int q = pos[a] + taps[t][a];
            inside = inside && q >= 0 && q < dim[a];
        }
        if (!inside) continue;
        if (nz == 2) { row25[t] = diag4; continue; }
        const int a = ax[0], o = taps[t][a], p = pos[a], n = dim[a];
        if (o == 2 || o == -2) { row25[t] = far2; continue; }
        // o = +-1: lo = the smaller of the two coordinates on this axis
        const int lo = o < 0 ? p - 1 : p;
        int count = 0;
        if (lo >= 1) ++count;     // the 1 -2 1 operator centred on lo
        if (lo + 2 <= n - 1) ++count; // ... centred on lo + 1
        for (int b = 0; b < 3; ++b) {
            if (b == a) continue;
            if (pos[b] > 0) ++count;           // the mixed cell on the low side of axis b
            if (pos[b] < dim[b] - 1) ++count;  // ... on the high side
        }
        float v = 0.0f;
        for (int i = 0; i < count; ++i) v += near4;
        row25[t] = v;
    }
}

struct SyncWs {
    VmSyncSys S;
    int *idx;
    float *val;
    float *tab;
};

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" int vm_sync_optimize_level(vm_sync *s, int lvl, float max_iter, volatile const int *run_flag,
                                      vm_sync_progress *out)
{
    CHECK_SYNC(s);
    CHECK_SLVL(s, lvl);
    vm_ctx *c = s->ctx;
    if (!s->f[lvl][0]) return vm_fail(VM_E_STATE, "vm_sync_optimize_level: level %d holds no field (load_identity / upsample_level)", lvl);
    if (!std::isfinite(max_iter) || max_iter > 1048576.0f) return vm_fail(VM_E_INVALID, "vm_sync_optimize_level: max_iter %g", (double)max_iter);
    const int passes = max_iter < 0 ? 0 : (int)std::floor(max_iter) + 1; // k = 0; while (k <= max_iter) k++
    VmSyncGrid g;
    g.w = s->w[lvl]; g.h = s->h[lvl]; g.d = s->d[lvl];
    g.nbx = (g.w + VM_SB_X - 1) / VM_SB_X; g.nby = (g.h + VM_SB_Y - 1) / VM_SB_Y; g.nbz = (g.d + VM_SB_Z - 1) / VM_SB_Z;
    g.nb = g.nbx * g.nby * g.nbz;
    g.per_xcd = (g.nb + 7) / 8;
    const size_t N = (size_t)g.w * g.h * g.d;
    // the UI part of genMatrix (:155-187) on the host, per voxel in constraint order
    const float w_ui = c->kp.w_ui, w_tps = c->kp.w_tps;
    std::map<size_t, std::array<float, 4>> ui;
    {
        const float ratio_x = (float)g.w / (float)s->w[0], ratio_y = (float)g.h / (float)s->h[0];
        for (const vm_sync_constraint &q : s->cons) {
            const float x0 = q.lx * ratio_x, y0 = q.ly * ratio_y, z0 = (float)q.lz;
            const float x1 = q.rx * ratio_x, y1 = q.ry * ratio_y, z1 = (float)q.rz;
            const float con_x = (x0 + x1) / 2.0f, con_y = (y0 + y1) / 2.0f, con_z = (z0 + z1) / 2.0f;
            const float vx = (x1 - x0) / 2.0f, vy = (y1 - y0) / 2.0f, vz = (z1 - z0) / 2.0f;
            const int xa = (int)std::floor(con_x), ya = (int)std::floor(con_y), za = (int)std::floor(con_z);
            for (int z = za; z <= za + 1; ++z)
                for (int y = ya; y <= ya + 1; ++y)
                    for (int x = xa; x <= xa + 1; ++x) {
                        if (x < 0 || y < 0 || z < 0 || x >= g.w || y >= g.h || z >= g.d) continue;
                        const float faz = std::fabs(z - con_z), fay = std::fabs(y - con_y), fax = std::fabs(x - con_x);
                        if (!(faz < 1 && fay < 1 && fax < 1)) continue;
                        const float bw = (float)((1.0 - fax) * (1.0 - fay) * (1.0 - faz));
                        std::array<float, 4> &e = ui[((size_t)z * g.h + y) * g.w + x];
                        e[0] += bw * w_ui;
                        e[1] += bw * vx * w_ui;
                        e[2] += bw * vy * w_ui;
                        e[3] += bw * vz * w_ui;
                    }
        }
    }
    const int ne = (int)ui.size();
    // workspace
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    const size_t o_r = take(3 * N * sizeof(float)), o_p = take(6 * N * sizeof(float)), o_om = take(3 * N * sizeof(float));
    const size_t o_diag = take(N * sizeof(float)), o_tab = take(125 * 25 * sizeof(float));
    const size_t o_part = take((size_t)9 * g.nb * sizeof(double)), o_sc = take(VM_SYNC_SC_WORDS * sizeof(float));
    const size_t tk_bytes = (size_t)VM_SYNC_TICKET_WORDS(8 * g.per_xcd) * sizeof(unsigned), o_tk = take(tk_bytes);
    const size_t o_idx = take((size_t)std::max(ne, 1) * sizeof(int)), o_val = take((size_t)std::max(ne, 1) * 4 * sizeof(float));
    if (off > s->ws_bytes) {
        VM_HIP(hipStreamSynchronize(c->stream));
        if (s->ws) (void)hipFree(s->ws);
        s->ws = nullptr;
        s->ws_bytes = 0;
        VM_HIP(hipMalloc(&s->ws, off));
        s->ws_bytes = off;
    }
    char *base = (char *)s->ws;
    VmSyncSys S;
    for (int k = 0; k < 3; ++k) {
        S.x[k] = s->f[lvl][k];
        S.r[k] = (float *)(base + o_r) + (size_t)k * N;
        S.p[0][k] = (float *)(base + o_p) + (size_t)k * N;
        S.p[1][k] = (float *)(base + o_p) + (size_t)(3 + k) * N;
        S.om[k] = (float *)(base + o_om) + (size_t)k * N;
    }
    S.diag = (float *)(base + o_diag);
    S.tab = (float *)(base + o_tab);
    S.part = (double *)(base + o_part);
    S.sc = (float *)(base + o_sc);
    S.ticket = (unsigned *)(base + o_tk);
    int *d_idx = (int *)(base + o_idx);
    float *d_val = (float *)(base + o_val);

    VM_HIP(hipEventRecord(c->ev0, c->stream));
    VM_HIP(hipMemsetAsync(base + o_r, 0, 3 * N * sizeof(float), c->stream));
    VM_HIP(hipMemsetAsync(S.diag, 0, N * sizeof(float), c->stream));
    VM_HIP(hipMemsetAsync(S.sc, 0, VM_SYNC_SC_WORDS * sizeof(float), c->stream));
    VM_HIP(hipMemsetAsync(S.ticket, 0, tk_bytes, c->stream));
    std::vector<float> tab(125 * 25, 0.0f);
    for (int sz = 0; sz < 5; ++sz)
        for (int sy = 0; sy < 5; ++sy)
            for (int sx = 0; sx < 5; ++sx) {
                const int px = state_rep(sx, g.w), py = state_rep(sy, g.h), pz = state_rep(sz, g.d);
                if (px >= g.w || py >= g.h || pz >= g.d) continue;
                offdiag_row(px, py, pz, g.w, g.h, g.d, w_tps, &tab[((sz * 5 + sy) * 5 + sx) * 25]);
            }
    VM_HIP(hipMemcpyAsync((void *)S.tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    std::vector<int> h_idx(std::max(ne, 1));
    std::vector<float> h_val((size_t)std::max(ne, 1) * 4);
    {
        int i = 0;
        for (const auto &e : ui) {
            h_idx[i] = (int)e.first;
            for (int k = 0; k < 4; ++k) h_val[(size_t)k * ne + i] = e.second[k];
            ++i;
        }
    }
    if (ne > 0) {
        VM_HIP(hipMemcpyAsync(d_idx, h_idx.data(), (size_t)ne * sizeof(int), hipMemcpyHostToDevice, c->stream));
        VM_HIP(hipMemcpyAsync(d_val, h_val.data(), (size_t)ne * 4 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        vm_sync_launch_scatter(S.diag, d_idx, d_val, ne, c->stream);
        for (int k = 0; k < 3; ++k) vm_sync_launch_scatter(S.r[k], d_idx, d_val + (size_t)(k + 1) * ne, ne, c->stream);
    }
    vm_sync_launch_diag(g, S.diag, w_tps, c->stream);
    vm_sync_launch_rr(g, S, c->stream);
    VM_HIP(hipGetLastError());
    // the pageable staging vectors above must outlive their copies
    VM_HIP(hipStreamSynchronize(c->stream));
    int k = 0, launches = 1;
    while (k < passes) {
        if (run_flag && (k & 63) == 0 && !*run_flag) break;
        ++k;
        vm_sync_launch_iteration(g, S, k, c->stream);
        launches += 2;
        if ((k & 511) == 0) {
            VM_HIP(hipGetLastError());
            VM_HIP(hipStreamSynchronize(c->stream)); // bounds the queue, so a cancel takes effect soon
        }
    }
    vm_sync_launch_finish(g, S, k, c->stream);
    VM_HIP(hipGetLastError());
    VM_HIP(hipEventRecord(c->ev1, c->stream));
    float sc[VM_SYNC_SC_WORDS];
    VM_HIP(hipMemcpyAsync(sc, S.sc, sizeof(sc), hipMemcpyDeviceToHost, c->stream));
    VM_HIP(hipStreamSynchronize(c->stream));
    if (out) {
        out->iters = k;
        out->launches = launches;
        out->voxel_iters = (double)k * (double)N;
        VM_HIP(hipEventElapsedTime(&out->elapsed_ms, c->ev0, c->ev1));
        for (int q = 0; q < 3; ++q) out->resid[q] = sc[3 + (k & 1) * 3 + q];
    }
    return VM_OK;
}

// CSyncThread::run, SyncThread.cpp:58-84
extern "C" int vm_sync_solve(vm_sync *s, float max_iter, volatile const int *run_flag, vm_sync_progress *out)
{
    CHECK_SYNC(s);
    const int total = (int)s->w.size() - 1;
    float mi = max_iter * 10;
    for (int el = total; el > 0; --el) {
        int rc = el == total ? vm_sync_load_identity(s, el) : vm_sync_upsample_level(s, el);
        if (rc) return rc;
        vm_sync_progress pr{};
        if ((rc = vm_sync_optimize_level(s, el, mi, run_flag, &pr))) return rc;
        if (out) out[el - 1] = pr;
        mi /= 2;
        if (run_flag && !*run_flag) break;
    }
    s->vec_frame = -1;
    return VM_OK;
}

extern "C" int vm_sync_get_field(vm_sync *s, int lvl, float *x, float *y, float *z)
{
    CHECK_SYNC(s);
    CHECK_SLVL(s, lvl);
    if (!s->f[lvl][0]) return vm_fail(VM_E_STATE, "vm_sync_get_field: level %d holds no field", lvl);
    const size_t N = (size_t)s->w[lvl] * s->h[lvl] * s->d[lvl];
    float *dst[3] = {x, y, z};
    for (int c = 0; c < 3; ++c)
        if (dst[c]) VM_HIP(hipMemcpyAsync(dst[c], s->f[lvl][c], N * sizeof(float), hipMemcpyDeviceToHost, s->ctx->stream));
    VM_HIP(hipStreamSynchronize(s->ctx->stream));
    return VM_OK;
}

extern "C" int vm_sync_set_field(vm_sync *s, int lvl, const float *x, const float *y, const float *z)
{
    CHECK_SYNC(s);
    CHECK_SLVL(s, lvl);
    if (int rc = alloc_field(s, lvl)) return rc;
    const size_t N = (size_t)s->w[lvl] * s->h[lvl] * s->d[lvl];
    const float *src[3] = {x, y, z};
    for (int c = 0; c < 3; ++c)
        if (src[c]) VM_HIP(hipMemcpyAsync(s->f[lvl][c], src[c], N * sizeof(float), hipMemcpyHostToDevice, s->ctx->stream));
    VM_HIP(hipStreamSynchronize(s->ctx->stream));
    s->vec_frame = -1;
    return VM_OK;
}

static int ensure_vec(vm_sync *s)
{
    if (!s->vec) VM_HIP(hipMalloc((void **)&s->vec, (size_t)s->w[0] * s->h[0] * sizeof(float4)));
    return VM_OK;
}

static int refresh_vec(vm_sync *s, int lvl, int frame)
{
    if (int rc = ensure_vec(s)) return rc;
    const size_t page = (size_t)s->w[lvl] * s->h[lvl];
    vm_sync_launch_result(s->f[lvl][0] + frame * page, s->f[lvl][1] + frame * page, s->f[lvl][2] + frame * page, s->w[lvl],
                          s->h[lvl], s->w[0], s->h[0], s->vec, s->ctx->stream);
    VM_HIP(hipGetLastError());
    s->vec_frame = frame;
    return VM_OK;
}

extern "C" int vm_sync_result(vm_sync *s, int lvl, int frame, float *vec4)
{
    CHECK_SYNC(s);
    CHECK_SLVL(s, lvl);
    if (frame < 0 || frame >= s->d[lvl]) return vm_fail(VM_E_INVALID, "vm_sync_result: frame %d out of range", frame);
    if (!s->f[lvl][0]) return vm_fail(VM_E_STATE, "vm_sync_result: level %d holds no field", lvl);
    if (int rc = refresh_vec(s, lvl, frame)) return rc;
    if (vec4) VM_HIP(hipMemcpyAsync(vec4, s->vec, (size_t)s->w[0] * s->h[0] * sizeof(float4), hipMemcpyDeviceToHost, s->ctx->stream));
    VM_HIP(hipStreamSynchronize(s->ctx->stream));
    return VM_OK;
}

extern "C" int vm_sync_upload_frame(vm_sync *s, int side, int frame, const uint8_t *rgba, int pitch_bytes)
{
    CHECK_SYNC(s);
    const int w0 = s->w[0], h0 = s->h[0], d0 = s->d[0];
    if (side < 0 || side > 1 || frame < 0 || frame >= d0 || !rgba || pitch_bytes < w0 * 4)
        return vm_fail(VM_E_INVALID, "vm_sync_upload_frame: bad arguments");
    const size_t page = (size_t)w0 * h0;
    if (!s->video[side]) {
        VM_HIP(hipMalloc((void **)&s->video[side], page * d0 * sizeof(uchar4)));
        VM_HIP(hipMemsetAsync(s->video[side], 0, page * d0 * sizeof(uchar4), s->ctx->stream));
    }
    VM_HIP(hipMemcpy2DAsync(s->video[side] + frame * page, (size_t)w0 * 4, rgba, (size_t)pitch_bytes, (size_t)w0 * 4, h0,
                            hipMemcpyHostToDevice, s->ctx->stream));
    VM_HIP(hipStreamSynchronize(s->ctx->stream));
    return VM_OK;
}

extern "C" int vm_sync_upload_flow(vm_sync *s, int side, int frame, const float *flow_xy, int pitch_floats)
{
    CHECK_SYNC(s);
    const int w0 = s->w[0], h0 = s->h[0], d0 = s->d[0];
    if (side < 0 || side > 1 || frame < 0 || frame >= d0 || !flow_xy || pitch_floats < w0 * 2)
        return vm_fail(VM_E_INVALID, "vm_sync_upload_flow: bad arguments");
    const size_t page = (size_t)w0 * h0;
    if (!s->forw[side]) {
        VM_HIP(hipMalloc((void **)&s->forw[side], page * d0 * sizeof(float2)));
        VM_HIP(hipMemsetAsync(s->forw[side], 0, page * d0 * sizeof(float2), s->ctx->stream));
    }
    VM_HIP(hipMemcpy2DAsync(s->forw[side] + frame * page, (size_t)w0 * 8, flow_xy, (size_t)pitch_floats * 4, (size_t)w0 * 8, h0,
                            hipMemcpyHostToDevice, s->ctx->stream));
    VM_HIP(hipStreamSynchronize(s->ctx->stream));
    return VM_OK;
}

static int render_common(vm_sync *s, float fa, int frame)
{
    const int w0 = s->w[0], h0 = s->h[0], d0 = s->d[0];
    if (frame < 0 || frame >= d0) return vm_fail(VM_E_INVALID, "vm_sync_render: frame %d out of range", frame);
    if (!std::isfinite(fa)) return vm_fail(VM_E_INVALID, "vm_sync_render: fa is not finite");
    for (int k = 0; k < 2; ++k)
        if (!s->video[k] || !s->forw[k])
            return vm_fail(VM_E_STATE, "vm_sync_render: video %d or its flow was never uploaded", k);
    if (s->vec_frame != frame) {
        if (s->f[1][0]) {
            if (int rc = refresh_vec(s, 1, frame)) return rc;
        } else { // Pyramid::build leaves _vector zero until the thread delivers (pyramid.cu:66-69)
            if (int rc = ensure_vec(s)) return rc;
            VM_HIP(hipMemsetAsync(s->vec, 0, (size_t)w0 * h0 * sizeof(float4), s->ctx->stream));
            s->vec_frame = frame;
        }
    }
    if (!s->out) VM_HIP(hipMalloc((void **)&s->out, (size_t)w0 * h0 * 3));
    vm_sync_launch_render(s->out, w0 * 3, w0, h0, d0, fa, frame, s->vec, s->video[0], s->video[1], s->forw[0], s->forw[1],
                          s->ctx->stream);
    VM_HIP(hipGetLastError());
    return VM_OK;
}

extern "C" int vm_sync_render(vm_sync *s, float fa, int frame, uint8_t *rgb_out, int pitch_bytes)
{
    CHECK_SYNC(s);
    if (!rgb_out || pitch_bytes < s->w[0] * 3) return vm_fail(VM_E_INVALID, "vm_sync_render: bad output");
    if (int rc = render_common(s, fa, frame)) return rc;
    VM_HIP(hipMemcpy2DAsync(rgb_out, (size_t)pitch_bytes, s->out, (size_t)s->w[0] * 3, (size_t)s->w[0] * 3, s->h[0],
                            hipMemcpyDeviceToHost, s->ctx->stream));
    VM_HIP(hipStreamSynchronize(s->ctx->stream));
    return VM_OK;
}

extern "C" int vm_sync_render_dev(vm_sync *s, float fa, int frame, float *elapsed_ms)
{
    CHECK_SYNC(s);
    VM_HIP(hipEventRecord(s->ctx->ev0, s->ctx->stream));
    if (int rc = render_common(s, fa, frame)) return rc;
    VM_HIP(hipEventRecord(s->ctx->ev1, s->ctx->stream));
    VM_HIP(hipStreamSynchronize(s->ctx->stream));
    if (elapsed_ms) VM_HIP(hipEventElapsedTime(elapsed_ms, s->ctx->ev0, s->ctx->ev1));
    return VM_OK;
}
