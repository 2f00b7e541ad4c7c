// vm_video.cpp -- the stage-2 pyramid of a VIDEO pair and the temporally coupled solve:
// class Pyramid with depth > 1 (Algorithm/Pyramid.h:14-98), upsample() incl. its
// optical-flow half (Algorithm/upsample.cu:260-340), initialize_temp (:214-258) and the
// page schedule of Morph::optimize_level (Algorithm/morph.cu:1353-1441): the middle page
// with flag == false, then two chains outward, every page tied to its already solved
// neighbour by the temporal term (morph.cu:752-759).
//
// A page is a vm_level (the same state, the same sweep kernels as a frame pair).  The two
// chains do not depend on each other -- page mid+k needs mid+k-1, page mid-k needs
// mid-k+1 -- so step k of both runs as ONE batch of two pages (grid.z, vm_optimize_levels):
// the only parallelism the coupled formulation leaves inside a level.
#include "vm_host.h"
#include "vm_temporal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

#define CHECK_VID(v, lvl, page)                                                                        \
    if (!(v)) return vm_fail(VM_E_INVALID, "%s: video is NULL", __func__);                             \
    if ((lvl) < 0 || (lvl) >= (int)(v)->pages.size())                                                  \
        return vm_fail(VM_E_INVALID, "%s: level %d out of range", __func__, (lvl));                    \
    if ((page) < 0 || (page) >= (v)->depth[(lvl)])                                                     \
        return vm_fail(VM_E_INVALID, "%s: page %d out of range (level %d has %d)", __func__, (page), (lvl), (v)->depth[(lvl)]); \
    if (!vm_ctx_alive((v)->ctx)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__); \
    VM_ON_DEVICE((v)->ctx);

extern "C" int vm_video_create(vm_ctx *c, int nlevels, const int *w, const int *h, const int *d,
                               const int *factor_t, int depth0, vm_video **out)
{
    if (!c || !w || !h || !d || !out || nlevels < 2)
        return vm_fail(VM_E_INVALID, "vm_video_create: need ctx, sizes, depths and >= 2 levels");
    for (int i = 0; i < nlevels; ++i) {
        if (w[i] < 5 || h[i] < 5)
            return vm_fail(VM_E_INVALID, "vm_video_create: level %d is %dx%d (min 5x5)", i, w[i], h[i]);
        if (d[i] < 1 || (i > 0 && d[i] > d[i - 1]))
            return vm_fail(VM_E_INVALID, "vm_video_create: depth of level %d is %d (must be >= 1 and not grow towards coarse levels)", i, d[i]);
    }
    if (depth0 < d[0]) return vm_fail(VM_E_INVALID, "vm_video_create: depth0 %d < depth of the finest level %d", depth0, d[0]);
    // a level built with temporal stride 2 takes its page t from page min(2 t, d_prev - 1) of the
    // finer level (pyramid.cu:406-442; the reference's own depths are ceil((d_prev + 1) / 2),
    // pyramid.cu:237): two pages must never name the same source -- the flow pyramid scales the
    // pages it carries over in place
    for (int i = 1; i < nlevels; ++i) {
        const int ft = factor_t ? (factor_t[i] > 1 ? 2 : 1) : (d[i] != d[i - 1] ? 2 : 1);
        if (ft == 2 && (d[i] - 2) * 2 >= d[i - 1] - 1)
            return vm_fail(VM_E_INVALID, "vm_video_create: level %d (temporal stride 2, %d pages) would take two pages from the same page of level %d (%d pages)",
                           i, d[i], i - 1, d[i - 1]);
    }
    VM_ON_DEVICE(c);
    vm_video *v = new vm_video();
    v->ctx = c;
    v->device = c->device;
    v->depth0 = depth0;
    v->depth.assign(d, d + nlevels);
    // temporal stride each level was built with (pyramid.cu:468): given, or 2 where the depth shrank
    v->factor_t.assign(nlevels, 1);
    for (int i = 1; i < nlevels; ++i)
        v->factor_t[i] = factor_t ? (factor_t[i] > 1 ? 2 : 1) : (d[i] != d[i - 1] ? 2 : 1);
    // factor_d, pyramid.cu:470-477: doubles towards finer levels wherever the depth changes
    v->factor_d.assign(nlevels, 1.0f);
    for (int i = nlevels - 2; i >= 0; --i)
        v->factor_d[i] = v->depth[i + 1] != v->depth[i] ? v->factor_d[i + 1] * 2 : v->factor_d[i + 1];
    v->factor_d0 = depth0 != v->depth[0] ? v->factor_d[0] * 2 : v->factor_d[0];
    v->pages.resize(nlevels);
    for (int i = 0; i < nlevels; ++i) {
        v->pages[i].resize(d[i]);
        const bool with_images = i != nlevels - 1;
        for (int t = 0; t < d[i]; ++t) {
            vm_video_page &pg = v->pages[i][t];
            vm_level &l = pg.lv;
            l.w = w[i]; l.h = h[i];
            l.rs = (w[i] + 31) / 32 * 32;
            l.imp_rs = (w[i] + 4) / 5 + 2;
            l.imp_rows = (h[i] + 4) / 5 + 2;
            int rc = vm_level_alloc(c, l, with_images);
            if (rc != VM_OK) { vm_video_destroy(v); return rc; }
            l.view.factor_d = v->factor_d[i];
            if (with_images) {
                const size_t n = (size_t)l.rs * l.h;
                const size_t total = 5 * al256(n * 8) + al256(n * 4);
                if (hipMalloc(&pg.tslab, total) != hipSuccess || hipMemsetAsync(pg.tslab, 0, total, c->stream) != hipSuccess) {
                    vm_video_destroy(v);
                    return vm_fail(VM_E_DEVICE, "vm_video_create: out of device memory");
                }
                char *b = (char *)pg.tslab;
                for (int k = 0; k < 4; ++k) { pg.flow[k] = (float2 *)b; b += al256(n * 8); }
                pg.temp_ref = (float2 *)b; b += al256(n * 8);
                pg.temp_mask = (float *)b;
                l.temp_ref_store = pg.temp_ref;
                l.temp_mask_store = pg.temp_mask;
            }
        }
    }
    {
        const vm_level &l0 = v->pages[0][0].lv;
        const size_t n = (size_t)l0.rs * l0.h;
        hipError_t e = hipMalloc((void **)&v->acc, n * 3 * sizeof(long long));
        if (e == hipSuccess) e = hipMalloc((void **)&v->vcur, n * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&v->weight, n * 4);
        if (e != hipSuccess) { vm_video_destroy(v); return vm_fail(VM_E_DEVICE, "vm_video_create: %s", hipGetErrorString(e)); }
    }
    *out = v;
    return VM_OK;
}

extern "C" void vm_video_destroy(vm_video *v)
{
    if (!v) return;
    for (vm_video_lane &ln : v->lanes) // the pipeline's lanes are contexts of their own
        if (ln.c && vm_ctx_alive(ln.c)) {
            {
                VM_ON_DEVICE_VOID(ln.c);
                hipStreamSynchronize(ln.c->stream);
                hipFree(ln.acc);
            }
            vm_ctx_destroy(ln.c);
        }
    v->lanes.clear();
    const bool alive = vm_ctx_alive(v->ctx); // destroyed after its context: freed without it (vm_api.cpp)
    VmDeviceGuard g(v->device);
    if (g.ok) {
        if (alive) hipStreamSynchronize(v->ctx->stream);
        else hipDeviceSynchronize();
        for (auto &lv : v->pages)
            for (auto &pg : lv) {
                vm_level_free(pg.lv);
                hipFree(pg.tslab);
            }
        hipFree(v->acc); hipFree(v->vcur); hipFree(v->weight); hipFree(v->result_tmp);
        (void)hipGetLastError();
    }
    delete v;
}

extern "C" int vm_video_levels(vm_video *v) { return v ? (int)v->pages.size() : 0; }

extern "C" int vm_video_level_dims(vm_video *v, int lvl, int *w, int *h, int *depth, float *factor_d)
{
    CHECK_VID(v, lvl, 0);
    if (w) *w = v->pages[lvl][0].lv.w;
    if (h) *h = v->pages[lvl][0].lv.h;
    if (depth) *depth = v->depth[lvl];
    if (factor_d) *factor_d = v->factor_d[lvl];
    return VM_OK;
}

extern "C" int vm_video_upload_luma(vm_video *v, int lvl, int page, const float *img0, const float *img1, int pitch)
{
    CHECK_VID(v, lvl, page);
    vm_level &l = v->pages[lvl][page].lv;
    if (!l.view.img0) return vm_fail(VM_E_STATE, "vm_video_upload_luma: the coarsest level holds no images");
    if (!img0 || !img1) return vm_fail(VM_E_INVALID, "vm_video_upload_luma: NULL image");
    if (pitch == 0) pitch = l.w;
    if (pitch < l.w) return vm_fail(VM_E_INVALID, "vm_video_upload_luma: pitch < width");
    hipStream_t s = v->ctx->stream;
    VM_HIP(hipMemcpy2DAsync((void *)l.view.img0, l.rs * 4, img0, (size_t)pitch * 4, (size_t)l.w * 4, l.h, hipMemcpyHostToDevice, s));
    VM_HIP(hipMemcpy2DAsync((void *)l.view.img1, l.rs * 4, img1, (size_t)pitch * 4, (size_t)l.w * 4, l.h, hipMemcpyHostToDevice, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

// the cudaMemcpy2DToArray uploads of lvl.f0/f1/b0/b1, pyramid.cu:323-326, 452-455
extern "C" int vm_video_upload_flows(vm_video *v, int lvl, int page, const float *f0, const float *f1,
                                     const float *b0, const float *b1, int pitch)
{
    CHECK_VID(v, lvl, page);
    vm_video_page &pg = v->pages[lvl][page];
    vm_level &l = pg.lv;
    if (!pg.tslab) return vm_fail(VM_E_STATE, "vm_video_upload_flows: the coarsest level holds no flows");
    if (pitch == 0) pitch = 2 * l.w;
    if (pitch < 2 * l.w) return vm_fail(VM_E_INVALID, "vm_video_upload_flows: pitch < 2*width");
    hipStream_t s = v->ctx->stream;
    const float *src[4] = {f0, f1, b0, b1};
    for (int k = 0; k < 4; ++k)
        if (src[k])
            VM_HIP(hipMemcpy2DAsync(pg.flow[k], l.rs * 8, src[k], (size_t)pitch * 4, (size_t)l.w * 8, l.h, hipMemcpyHostToDevice, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

extern "C" int vm_video_set_v(vm_video *v, int lvl, int page, const float *vxy, int pitch)
{
    CHECK_VID(v, lvl, page);
    vm_level &l = v->pages[lvl][page].lv;
    if (!vxy) return vm_fail(VM_E_INVALID, "vm_video_set_v: NULL");
    if (pitch == 0) pitch = 2 * l.w;
    if (pitch < 2 * l.w) return vm_fail(VM_E_INVALID, "vm_video_set_v: pitch < 2*width");
    hipStream_t s = v->ctx->stream;
    VM_HIP(hipMemcpy2DAsync(l.view.v, l.rs * 8, vxy, (size_t)pitch * 4, (size_t)l.w * 8, l.h, hipMemcpyHostToDevice, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

extern "C" int vm_video_get_v(vm_video *v, int lvl, int page, float *vxy, int pitch)
{
    CHECK_VID(v, lvl, page);
    vm_level &l = v->pages[lvl][page].lv;
    if (!vxy) return vm_fail(VM_E_INVALID, "vm_video_get_v: NULL");
    if (pitch == 0) pitch = 2 * l.w;
    if (pitch < 2 * l.w) return vm_fail(VM_E_INVALID, "vm_video_get_v: pitch < 2*width");
    hipStream_t s = v->ctx->stream;
    VM_HIP(hipMemcpy2DAsync(vxy, (size_t)pitch * 4, l.view.v, l.rs * 8, (size_t)l.w * 8, l.h, hipMemcpyDeviceToHost, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

// CMatchingThread::update_result for one frame of the full-resolution result
// (MatchingThread.cpp:22-84), on the device: `dst` (pitch in float2) receives frame f.  Which
// page(s) a frame comes from follows the reference's two loops: page i goes to frame
// min(i * factor, depth0 - 1) (a later page overwrites an earlier one that lands on the same
// frame), then frame i * factor + k (0 < k < factor, below depth0 - 1) is the blend of frames
// i * factor and min((i + 1) * factor, depth0 - 1).  tmp: two w0 x h0 float2 planes.
static int video_result_frame(vm_video *v, int lvl, int w0, int h0, int f, float2 *dst, int dpitch, float2 *tmp)
{
    hipStream_t s = v->ctx->stream;
    const int d = v->depth[lvl], depth0 = v->depth0;
    const int factor = (int)(v->factor_d0 / v->factor_d[lvl]); // MatchingThread.cpp:27
    // The reference's two loops write frames 0 .. min((d - 1) factor, depth0 - 1); with its own depth tables
    // (d = ceil((d_finer + 1) / 2), pyramid.cu:237) that is every frame.  A table whose last page stops
    // short of the last frame would leave the tail frames with whatever an EARLIER delivery put into
    // Pyramid::_vector there -- content this library does not have: refused instead of returned as zeros.
    if ((d - 1) * factor < depth0 - 1)
        return vm_fail(VM_E_STATE, "vm_video_result: level %d delivers frames 0..%d of %d (%d pages, stride %d): the reference would "
                                   "leave the rest as an earlier update_result wrote them; deliver a level that reaches the last frame",
                       lvl, (d - 1) * factor, depth0, d, factor);
    auto key_page = [&](int frame) {
        int key = -1;
        for (int i = 0; i < d; ++i)
            if (std::min(i * factor, depth0 - 1) == frame) key = i;
        return key;
    };
    auto resize = [&](int page, float2 *out, int pitch) {
        const vm_level &l = v->pages[lvl][page].lv;
        vm_launch_upscale(out, w0, h0, pitch, l.view.v, l.w, l.h, l.rs, s);
    };
    if (factor > 1)
        for (int i = 0; i < d - 1; ++i)
            for (int k = 1; k < factor; ++k) {
                if (i * factor + k >= depth0 - 1 || i * factor + k != f) continue;
                const int beg = i * factor, end = std::min((i + 1) * factor, depth0 - 1);
                const float fa = (float)k / (float)(end - beg);
                const int pb = key_page(beg), pe = key_page(end);
                float2 *ta = tmp, *tb = tmp + (size_t)w0 * h0;
                if (pb >= 0) resize(pb, ta, w0); else VM_HIP(hipMemsetAsync(ta, 0, (size_t)w0 * h0 * 8, s));
                if (pe >= 0) resize(pe, tb, w0); else VM_HIP(hipMemsetAsync(tb, 0, (size_t)w0 * h0 * 8, s));
                vm_launch_blend_v(dst, dpitch, ta, tb, w0, w0, h0, 1 - fa, fa, s);
                VM_HIP(hipGetLastError());
                return VM_OK;
            }
    const int key = key_page(f);
    if (key >= 0) resize(key, dst, dpitch);
    else VM_HIP(hipMemset2DAsync(dst, (size_t)dpitch * 8, 0, (size_t)w0 * 8, h0, s)); // a frame nothing writes
    VM_HIP(hipGetLastError());
    return VM_OK;
}

extern "C" int vm_video_result(vm_video *v, int lvl, int w0, int h0, float *out)
{
    CHECK_VID(v, lvl, 0);
    if (!out || w0 < 1 || h0 < 1) return vm_fail(VM_E_INVALID, "vm_video_result: bad argument");
    std::lock_guard<std::recursive_mutex> lock(v->ctx->mu);
    hipStream_t s = v->ctx->stream;
    const size_t n = (size_t)w0 * h0;
    float2 *buf = nullptr;
    VM_HIP(hipMalloc((void **)&buf, 3 * n * 8));
    int rc = VM_OK;
    for (int f = 0; f < v->depth0 && rc == VM_OK; ++f) {
        rc = video_result_frame(v, lvl, w0, h0, f, buf, w0, buf + n);
        if (rc == VM_OK && hipMemcpyAsync(out + (size_t)f * n * 2, buf, n * 8, hipMemcpyDeviceToHost, s) != hipSuccess)
            rc = vm_fail(VM_E_DEVICE, "vm_video_result: copy failed");
        if (rc == VM_OK && hipStreamSynchronize(s) != hipSuccess) rc = vm_fail(VM_E_DEVICE, "vm_video_result: sync failed");
    }
    hipStreamSynchronize(s);
    hipFree(buf);
    return rc;
}

extern "C" int vm_frame_set_v_from_video(vm_frame *f, vm_video *v, int lvl, int frame)
{
    CHECK_VID(v, lvl, 0);
    if (!f || f->ctx != v->ctx) return vm_fail(VM_E_INVALID, "vm_frame_set_v_from_video: the frame and the video must share a context");
    if (frame < 0 || frame >= v->depth0) return vm_fail(VM_E_INVALID, "vm_frame_set_v_from_video: frame %d out of range (0..%d)", frame, v->depth0 - 1);
    std::lock_guard<std::recursive_mutex> lock(v->ctx->mu);
    const size_t n = (size_t)f->w * f->h;
    if (!v->result_tmp || v->result_tmp_elems < 2 * n) {
        VM_HIP(hipStreamSynchronize(v->ctx->stream));
        hipFree(v->result_tmp);
        v->result_tmp = nullptr;
        v->result_tmp_elems = 0;
        VM_HIP(hipMalloc((void **)&v->result_tmp, 2 * n * 8));
        v->result_tmp_elems = 2 * n;
    }
    return video_result_frame(v, lvl, f->w, f->h, frame, f->v, f->rs, v->result_tmp);
}

extern "C" int vm_video_get_field(vm_video *v, int lvl, int page, int field, void *host)
{
    CHECK_VID(v, lvl, page);
    vm_video_page &pg = v->pages[lvl][page];
    if (field >= VM_F_FLOW_F0 && field <= VM_F_FLOW_B1) {
        if (!host) return vm_fail(VM_E_INVALID, "vm_video_get_field: NULL");
        if (!pg.tslab) return vm_fail(VM_E_STATE, "vm_video_get_field: the coarsest level holds no flows");
        vm_level &l = pg.lv;
        hipStream_t s = v->ctx->stream;
        VM_HIP(hipMemcpy2DAsync(host, (size_t)l.w * 8, pg.flow[field - VM_F_FLOW_F0], (size_t)l.rs * 8, (size_t)l.w * 8, l.h, hipMemcpyDeviceToHost, s));
        VM_HIP(hipStreamSynchronize(s));
        return VM_OK;
    }
    return vm_level_read_field(v->ctx, pg.lv, field, host);
}

// constraints of page z of level lvl: conz = min(z * factor, depth0 - 1), factor =
// lv0.factor_d / lvl.factor_d (morph.cu:351-358, 472-478)
static std::vector<vm_constraint> page_constraints(const vm_video *v, int lvl, int z, const vm_video_constraint *cons, int n)
{
    const int factor = (int)(v->factor_d0 / v->factor_d[lvl]);
    const int conz = std::min(z * factor, v->depth0 - 1);
    std::vector<vm_constraint> out;
    for (int k = 0; k < n; ++k)
        if (cons[k].frame == conz)
            out.push_back({cons[k].lx, cons[k].ly, cons[k].rx, cons[k].ry, cons[k].weight});
    return out;
}

// Morph::cpu_optimize_level for every page of the coarsest level, morph.cu:419-590
extern "C" int vm_video_coarse_solve(vm_video *v, const vm_video_constraint *cons, int n)
{
    if (!v) return vm_fail(VM_E_INVALID, "vm_video_coarse_solve: video is NULL");
    if (n < 0 || (n > 0 && !cons)) return vm_fail(VM_E_INVALID, "vm_video_coarse_solve: constraints");
    VM_ON_DEVICE(v->ctx);
    const int L = (int)v->pages.size() - 1;
    const int w0 = v->pages[0][0].lv.w, h0 = v->pages[0][0].lv.h;
    for (int z = 0; z < v->depth[L]; ++z) {
        vm_level &l = v->pages[L][z].lv;
        std::vector<vm_constraint> pc = page_constraints(v, L, z, cons, n);
        std::vector<float> vv((size_t)2 * l.w * l.h, 0.0f);
        int rc = vm_host_coarse_solve(l.w, l.h, w0, h0, v->ctx->kp, pc.data(), (int)pc.size(), vv.data(), v->depth[L]);
        if (rc != VM_OK) return rc;
        if ((rc = vm_video_set_v(v, L, z, vv.data(), 0)) != VM_OK) return rc;
    }
    return VM_OK;
}

// temp_ref + interpolate_temp_ref of one neighbour page into the scratch accumulators
static int splat_page(vm_video *v, const vm_video_page &src, const float2 *fa, const float2 *fb, bool with_ssim)
{
    const vm_level &l = src.lv;
    vm_temp_launch_splat(l.w, l.h, l.rs, l.view.v, fa, fb, with_ssim ? l.view.value : nullptr, v->acc, v->ctx->stream);
    VM_HIP(hipGetLastError());
    return VM_OK;
}

// upsample(pyr[dst], pyr[dst+1]), upsample.cu:260-340
extern "C" int vm_video_upsample(vm_video *v, int dst)
{
    CHECK_VID(v, dst, 0);
    if (dst + 1 >= (int)v->pages.size()) return vm_fail(VM_E_INVALID, "vm_video_upsample: level %d has no coarser level", dst);
    vm_ctx *c = v->ctx;
    hipStream_t s = c->stream;
    const int d = v->depth[dst], ds = v->depth[dst + 1];
    const int factor = d > ds ? 2 : 1;
    int rc;
    for (int t = 0; t < d; ++t) { // dest.v.fill(0)
        vm_level &l = v->pages[dst][t].lv;
        VM_HIP(hipMemsetAsync(l.view.v, 0, (size_t)l.rs * l.h * 8, s));
    }
    for (int i = 0; i < ds; ++i)
        if ((rc = vm_level_upsample(c, v->pages[dst][std::min(i * factor, d - 1)].lv, v->pages[dst + 1][i].lv)) != VM_OK) return rc;
    if (factor > 1) { // in-between pages from their two neighbours, along the optical flow
        for (int i = 1; i < d; i += factor) {
            if (i == d - 1) continue;
            vm_video_page &pg = v->pages[dst][i], &pp = v->pages[dst][i - 1], &pn = v->pages[dst][i + 1];
            vm_level &l = pg.lv;
            const size_t n = (size_t)l.rs * l.h;
            VM_HIP(hipMemsetAsync(v->acc, 0, n * 3 * sizeof(long long), s));
            if ((rc = splat_page(v, pp, pp.flow[0], pp.flow[1], false)) != VM_OK) return rc; // f0, f1 of page i-1
            if ((rc = splat_page(v, pn, pn.flow[2], pn.flow[3], false)) != VM_OK) return rc; // b0, b1 of page i+1
            vm_temp_launch_finish(l.w, l.h, l.rs, v->acc, v->vcur, v->weight, 0, s);
            VM_HIP(hipMemsetAsync(l.view.v, 0, n * 8, s));
            vm_temp_launch_smooth(l.w, l.h, l.rs, l.view.v, v->vcur, v->weight, s);
            vm_temp_launch_fill_zeros_x(l.w, l.h, l.rs, l.view.v, v->weight, s);
            VM_HIP(hipGetLastError());
        }
    }
    return VM_OK;
}

// Morph::initialize_level for every page, morph.cu:264-390
extern "C" int vm_video_init_level(vm_video *v, int lvl, const vm_video_constraint *cons, int n)
{
    CHECK_VID(v, lvl, 0);
    if (n < 0 || (n > 0 && !cons)) return vm_fail(VM_E_INVALID, "vm_video_init_level: constraints");
    const int w0 = v->pages[0][0].lv.w, h0 = v->pages[0][0].lv.h;
    for (int z = 0; z < v->depth[lvl]; ++z) {
        vm_video_page &pg = v->pages[lvl][z];
        if (!pg.tslab) return vm_fail(VM_E_STATE, "vm_video_init_level: the coarsest level is solved by vm_video_coarse_solve");
        std::vector<vm_constraint> pc = page_constraints(v, lvl, z, cons, n);
        int rc = vm_level_init(v->ctx, pg.lv, w0, h0, pc.data(), (int)pc.size());
        if (rc != VM_OK) return rc;
        // lvl.temp.ref.fill(0); lvl.temp.mask.fill(0), morph.cu:313-314; flag == false until
        // initialize_temp ties the page to a neighbour
        const size_t np = (size_t)pg.lv.rs * pg.lv.h;
        VM_HIP(hipMemsetAsync(pg.temp_ref, 0, np * 8, v->ctx->stream));
        VM_HIP(hipMemsetAsync(pg.temp_mask, 0, np * 4, v->ctx->stream));
        pg.lv.view.temp_ref = nullptr;
        pg.lv.view.temp_mask = nullptr;
        pg.lv.view.factor_d = v->factor_d[lvl];
    }
    return VM_OK;
}

// initialize_temp(lvl, i, dir), upsample.cu:214-258
extern "C" int vm_video_initialize_temp(vm_video *v, int lvl, int page, int dir)
{
    CHECK_VID(v, lvl, page);
    if (dir != 1 && dir != -1) return vm_fail(VM_E_INVALID, "vm_video_initialize_temp: dir must be +1 or -1");
    const int j = page + dir;
    if (j < 0 || j >= v->depth[lvl]) return vm_fail(VM_E_INVALID, "vm_video_initialize_temp: page %d has no neighbour in direction %d", page, dir);
    vm_video_page &pg = v->pages[lvl][page], &src = v->pages[lvl][j];
    if (!pg.tslab) return vm_fail(VM_E_STATE, "vm_video_initialize_temp: the coarsest level has no temporal state");
    if (!src.lv.has_state) return vm_fail(VM_E_STATE, "vm_video_initialize_temp: neighbour page not initialised");
    vm_level &l = pg.lv;
    hipStream_t s = v->ctx->stream;
    VM_HIP(hipMemsetAsync(v->acc, 0, (size_t)l.rs * l.h * 3 * sizeof(long long), s));
    // dir < 0: the neighbour's forward flows carry it here; dir > 0: its backward flows
    int rc = splat_page(v, src, dir < 0 ? src.flow[0] : src.flow[2], dir < 0 ? src.flow[1] : src.flow[3], true);
    if (rc != VM_OK) return rc;
    vm_temp_launch_finish(l.w, l.h, l.rs, v->acc, pg.temp_ref, pg.temp_mask, 1, s);
    VM_HIP(hipGetLastError());
    l.view.temp_ref = pg.temp_ref;   // flag == true for this page from now on
    l.view.temp_mask = pg.temp_mask;
    return VM_OK;
}

// Morph::optimize_level, morph.cu:1353-1441.  out (may be NULL): depth[lvl] entries, index = page.
extern "C" int vm_video_optimize_level(vm_video *v, int lvl, float max_iter, volatile const int *run_flag,
                                       int fixed_work, vm_progress *out)
{
    CHECK_VID(v, lvl, 0);
    vm_ctx *c = v->ctx;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    const int d = v->depth[lvl], mid = d / 2;
    int rc;
    {
        vm_level *one = &v->pages[lvl][mid].lv;
        one->view.temp_ref = nullptr; // the middle page: flag == false
        one->view.temp_mask = nullptr;
        if ((rc = vm_optimize_levels(c, &one, 1, max_iter, run_flag, fixed_work, out ? &out[mid] : nullptr)) != VM_OK) return rc;
    }
    for (int k = 1; mid + k < d || mid - k >= 0; ++k) {
        if (run_flag && !*run_flag) return vm_fail(VM_E_CANCELLED, "vm_video_optimize_level: cancelled by run_flag");
        vm_level *pair[2];
        int idx[2], m = 0;
        if (mid + k < d) {
            if ((rc = vm_video_initialize_temp(v, lvl, mid + k, -1)) != VM_OK) return rc;
            idx[m] = mid + k;
            pair[m++] = &v->pages[lvl][mid + k].lv;
        }
        if (mid - k >= 0) {
            if ((rc = vm_video_initialize_temp(v, lvl, mid - k, +1)) != VM_OK) return rc;
            idx[m] = mid - k;
            pair[m++] = &v->pages[lvl][mid - k].lv;
        }
        vm_progress pr[2] = {};
        if ((rc = vm_optimize_levels(c, pair, m, max_iter, run_flag, fixed_work, pr)) != VM_OK) return rc;
        if (out)
            for (int i = 0; i < m; ++i) {
                out[idx[i]] = pr[i];
                if (i > 0) { // time and launches belong to the batch: count them once
                    out[idx[i]].elapsed_ms = 0;
                    out[idx[i]].launches = 0;
                    for (int q = 0; q < 5; ++q) { out[idx[i]].sched_ms[q] = 0; out[idx[i]].sched_launches[q] = 0; }
                }
            }
    }
    return VM_OK;
}

// ---------------------------------------------------------------------------------------------
// Level pipeline.  The reference finishes a level (middle page, then the chain outward) before it
// touches the next finer one.  But page p of level l needs only (i) page p of level l + 1 -- its
// own coarser solution, when both levels hold the same frames -- and (ii) its chain neighbour on
// level l.  So the tasks T(l, k) = "pages mid +- k of level l" form a grid whose anti-diagonals are
// independent: while the chain of the coarsest level is still walking outward (its 8000 dependent
// launch-bound steps per page are what a video solve waits for), the pages it has finished are
// already being solved on the finer levels, on other streams.  Same arithmetic per page, hence the
// same bits; about (levels + chain - 1) task slots instead of levels x chain.

static int ensure_lanes(vm_video *v, int n)
{
    vm_ctx *p = v->ctx;
    const vm_level &l0 = v->pages[0][0].lv;
    const size_t np = (size_t)l0.rs * l0.h;
    while ((int)v->lanes.size() < n) {
        vm_video_lane ln;
        int rc = vm_ctx_create(p->device, &ln.c);
        if (rc != VM_OK) return rc;
        // A lane's stream must run SIDE BY SIDE with the other lanes': the runtime deals a new stream to one of its hardware
        // queues as it likes, and two streams on one queue take turns (measured on the frame-pair solver's streams: + 58 %,
        // profiles/r06_notes.md section 4).  A candidate that shares a queue with a lane already chosen is kept alive while
        // the next one is tried (so that it lands elsewhere) and destroyed afterwards; after six tries the last one is kept.
        {
            std::vector<vm_ctx *> parked;
            for (int attempt = 0; attempt < 6; ++attempt) {
                bool beside = true;
                for (const vm_video_lane &o : v->lanes) {
                    int ov = 1;
                    if (vm_dbg_streams_overlap(ln.c, o.c, &ov) == VM_OK && !ov) { beside = false; break; }
                }
                if (beside) break;
                parked.push_back(ln.c);
                ln.c = nullptr;
                rc = vm_ctx_create(p->device, &ln.c);
                if (rc != VM_OK) break;
            }
            for (vm_ctx *q : parked) vm_ctx_destroy(q);
            if (rc != VM_OK) return rc;
        }
        if (hipMalloc((void **)&ln.acc, np * 3 * sizeof(long long)) != hipSuccess) {
            vm_ctx_destroy(ln.c);
            return vm_fail(VM_E_DEVICE, "vm_video_solve: out of device memory (pipeline lane)");
        }
        // Lanes run on plain streams.  Rounds 2-4 gave them PRIORITY streams (lane 0, which runs the coarsest level in
        // flight -- the chain of launch-bound steps the whole solve waits for -- the highest): the solve does not
        // care (404-405 ms either way), but a priority stream takes a hardware queue of its own which the runtime
        // does not hand back after hipStreamDestroy, and every later multi-stream job of the process then runs with
        // streams sharing queues (config[2]'s three streams +17 % behind the graded scheme; the six streams of the
        // config[4] pipeline +19 % behind even one high-priority lane: profiles/r05_notes.md 7).
        // VM_LANE_PRIORITY (development): 0 = none (default), 1 = lane 0 high, 2 = graded (lane j at greatest + j)
        {
            static const int mode = getenv("VM_LANE_PRIORITY") ? atoi(getenv("VM_LANE_PRIORITY")) : 0;
            int least = 0, greatest = 0;
            hipStream_t ps = nullptr;
            const int j = (int)v->lanes.size();
            if ((mode == 2 || (mode == 1 && j == 0)) &&
                hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest &&
                hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, std::min(greatest + j, least)) == hipSuccess) {
                (void)hipStreamDestroy(ln.c->stream);
                ln.c->stream = ps;
            }
        }
        v->lanes.push_back(ln);
    }
    for (vm_video_lane &ln : v->lanes) { // the solver settings of the video's context, as they are now
        ln.c->kp = p->kp;
        ln.c->math_mode = p->math_mode;
        ln.c->sweep_threads = p->sweep_threads;
        ln.c->sweep_mode = p->sweep_mode;
        ln.c->sweep_parts = p->sweep_parts;
        ln.c->commit_order = p->commit_order;
        ln.c->sparse_resident = p->sparse_resident;       // the debug switches too: a hook set on the
        ln.c->pass_test_timeout = p->pass_test_timeout;   // video's context acts on the lanes that do its work
    }
    return VM_OK;
}

// T(el, k): upsample + initialize_level of the task's pages, initialize_temp against their chain
// neighbours (k > 0), the sweeps.  Runs on the lane's stream; everything is complete on return.
static int video_task(vm_video *v, vm_video_lane &ln, int el, int k, float max_iter, const vm_video_constraint *cons,
                      int n, volatile const int *run_flag, int fixed_work, vm_progress *out)
{
    vm_ctx *c = ln.c;
    VM_ON_DEVICE(c);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    hipStream_t s = c->stream;
    const int d = v->depth[el], mid = d / 2;
    const int w0 = v->pages[0][0].lv.w, h0 = v->pages[0][0].lv.h;
    int idx[2], dir[2], m = 0;
    if (k == 0) {
        idx[m] = mid; dir[m++] = 0;
    } else {
        if (mid + k < d) { idx[m] = mid + k; dir[m++] = -1; }
        if (mid - k >= 0) { idx[m] = mid - k; dir[m++] = +1; }
    }
    if (m == 0) return VM_OK;
    vm_level *pair[2];
    int rc;
    for (int i = 0; i < m; ++i) {
        vm_video_page &pg = v->pages[el][idx[i]];
        vm_level &l = pg.lv;
        const size_t np = (size_t)l.rs * l.h;
        VM_HIP(hipMemsetAsync(l.view.v, 0, np * 8, s)); // upsample(): dest.v.fill(0)
        if ((rc = vm_level_upsample(c, l, v->pages[el + 1][idx[i]].lv)) != VM_OK) return rc;
        std::vector<vm_constraint> pc = page_constraints(v, el, idx[i], cons, n);
        if ((rc = vm_level_init(c, l, w0, h0, pc.data(), (int)pc.size())) != VM_OK) return rc;
        VM_HIP(hipMemsetAsync(pg.temp_ref, 0, np * 8, s));
        VM_HIP(hipMemsetAsync(pg.temp_mask, 0, np * 4, s));
        l.view.temp_ref = nullptr;
        l.view.temp_mask = nullptr;
        l.view.factor_d = v->factor_d[el];
        if (dir[i] != 0) { // initialize_temp(lvl, page, dir), upsample.cu:214-258
            const vm_video_page &src = v->pages[el][idx[i] + dir[i]];
            VM_HIP(hipMemsetAsync(ln.acc, 0, np * 3 * sizeof(long long), s));
            vm_temp_launch_splat(src.lv.w, src.lv.h, src.lv.rs, src.lv.view.v, dir[i] < 0 ? src.flow[0] : src.flow[2],
                                 dir[i] < 0 ? src.flow[1] : src.flow[3], src.lv.view.value, ln.acc, s);
            vm_temp_launch_finish(l.w, l.h, l.rs, ln.acc, pg.temp_ref, pg.temp_mask, 1, s);
            VM_HIP(hipGetLastError());
            l.view.temp_ref = pg.temp_ref;
            l.view.temp_mask = pg.temp_mask;
        }
        pair[i] = &l;
    }
    vm_progress pr[2] = {};
    if ((rc = vm_optimize_levels(c, pair, m, max_iter, run_flag, fixed_work, pr)) != VM_OK) return rc;
    VM_HIP(hipStreamSynchronize(s));
    if (out)
        for (int i = 0; i < m; ++i) {
            out[idx[i]] = pr[i];
            if (i > 0) { // time and launches belong to the batch: count them once
                out[idx[i]].elapsed_ms = 0;
                out[idx[i]].launches = 0;
                for (int q = 0; q < 5; ++q) { out[idx[i]].sched_ms[q] = 0; out[idx[i]].sched_launches[q] = 0; }
            }
        }
    return VM_OK;
}

// Morph::calculate_halfway_parametrization, morph.cu:150-168.  per_page (may be NULL):
// sum over the levels with images of depth[l] entries, level-major (finest first), page-minor.
extern "C" int vm_video_solve(vm_video *v, float max_iter, float drop, const vm_video_constraint *cons, int n,
                              volatile const int *run_flag, int fixed_work, vm_progress *per_page)
{
    if (!v) return vm_fail(VM_E_INVALID, "vm_video_solve: video is NULL");
    if (!(drop > 0)) return vm_fail(VM_E_INVALID, "vm_video_solve: max_iter_drop_factor must be > 0");
    std::lock_guard<std::recursive_mutex> lock(v->ctx->mu);
    VM_ON_DEVICE(v->ctx);
    const int L = (int)v->pages.size();
    int rc = vm_video_coarse_solve(v, cons, n);
    if (rc != VM_OK) return rc;
    std::vector<size_t> off(L, 0);
    for (int l = 1; l < L; ++l) off[l] = off[l - 1] + v->depth[l - 1];
    // the finest levels that hold the same frames as the level above them can be pipelined:
    // E = the coarsest of them (levels E..0); -1 = none
    int E = -1;
    if (!getenv("VM_NO_VIDEO_PIPELINE") && v->depth[0] > 1)
        while (E + 1 <= L - 2 && v->depth[E + 1] == v->depth[E + 2]) ++E;
    if (E < 1) E = -1; // a single level has nothing to overlap with
    float mi = max_iter;
    std::vector<float> mi_of(L, max_iter);
    for (int el = L - 2; el >= 0; --el) {
        mi_of[el] = mi;
        mi /= drop;
    }
    for (int el = L - 2; el > E; --el) { // level by level, as the reference walks them
        if (run_flag && !*run_flag) return vm_fail(VM_E_CANCELLED, "vm_video_solve: cancelled by run_flag");
        if ((rc = vm_video_upsample(v, el)) != VM_OK) return rc;
        if ((rc = vm_video_init_level(v, el, cons, n)) != VM_OK) return rc;
        if ((rc = vm_video_optimize_level(v, el, mi_of[el], run_flag, fixed_work, per_page ? per_page + off[el] : nullptr)) != VM_OK) return rc;
    }
    if (E < 0) return VM_OK;
    VM_HIP(hipStreamSynchronize(v->ctx->stream));
    const int A = E + 1, d = v->depth[0], mid = d / 2, K = std::max(mid, d - 1 - mid);
    if ((rc = ensure_lanes(v, std::min(A, K + 1))) != VM_OK) return rc;
    // Dataflow over the task grid: T(a, k) (a = 0: the coarsest pipelined level, k = chain step) is
    // ready once T(a - 1, k) and T(a, k - 1) are done.  One worker per lane; lane 0 (highest stream
    // priority) is reserved for the coarsest level's chain, the critical path; the others take the
    // ready task of the coarsest level first.
    struct Task { int deps; bool done; };
    std::vector<Task> tasks((size_t)A * (K + 1));
    auto T = [&](int a_, int k_) -> Task & { return tasks[(size_t)a_ * (K + 1) + k_]; };
    for (int a_ = 0; a_ < A; ++a_)
        for (int k_ = 0; k_ <= K; ++k_) T(a_, k_) = {(a_ > 0) + (k_ > 0), false};
    std::mutex mu;
    std::condition_variable cv;
    int remaining = A * (K + 1), failed_rc = VM_OK;
    std::string failed_msg;
    const int nl = std::min(A, K + 1);
    auto worker = [&](int lane) {
        // lane 0 takes only the coarsest level, the other lanes everything else (one lane: all of it)
        const int a_lo = (lane == 0 || nl == 1) ? 0 : 1, a_hi = (lane == 0 && nl > 1) ? 1 : A;
        for (;;) {
            int ta = -1, tk = -1;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (remaining == 0 || failed_rc != VM_OK) return;
                    bool mine_left = false; // anything this lane could still take, now or later?
                    for (int a_ = a_lo; a_ < a_hi && ta < 0; ++a_)
                        for (int k_ = 0; k_ <= K && ta < 0; ++k_) {
                            mine_left = mine_left || !T(a_, k_).done;
                            if (!T(a_, k_).done && T(a_, k_).deps == 0) { ta = a_; tk = k_; }
                        }
                    if (ta >= 0) break;
                    if (!mine_left) return;
                    cv.wait(lk);
                }
                T(ta, tk).deps = -1; // taken
            }
            const int el = E - ta;
            int trc = VM_OK;
            std::string msg;
            if (run_flag && !*run_flag)
                trc = VM_E_CANCELLED, msg = "vm_video_solve: cancelled by run_flag";
            else {
                trc = video_task(v, v->lanes[lane], el, tk, mi_of[el], cons, n, run_flag, fixed_work, per_page ? per_page + off[el] : nullptr);
                if (trc != VM_OK) msg = vm_last_error();
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                T(ta, tk).done = true;
                --remaining;
                if (trc != VM_OK && failed_rc == VM_OK) { failed_rc = trc; failed_msg = msg; }
                if (ta + 1 < A) --T(ta + 1, tk).deps;
                if (tk + 1 <= K) --T(ta, tk + 1).deps;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (int lane = 0; lane < nl; ++lane) th.emplace_back(worker, lane);
    for (std::thread &t : th) t.join();
    if (failed_rc != VM_OK) return vm_fail(failed_rc, "%s", failed_msg.c_str());
    return VM_OK;
}
