// vm_frame.cpp -- device-resident frame objects of the compositor: the C-ABI
// around vm_render.hip (render_halfway_image, Algorithm/render.cu:62-96) and
// the result upscale (CMatchingThread::update_result, MatchingThread.cpp:22-100).
#include "vm_host.h"
#include "vm_poisson.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

extern "C" int vm_frame_create(vm_ctx *c, int w, int h, int ex, vm_frame **out)
{
    if (!c || !out || w < 1 || h < 1 || ex < 0)
        return vm_fail(VM_E_INVALID, "vm_frame_create: bad argument");
    VM_ON_DEVICE(c);
    vm_frame *f = new vm_frame();
    f->ctx = c;
    f->device = c->device;
    f->w = w; f->h = h; f->ex = ex;
    f->cw = w + 2 * ex; f->ch = h + 2 * ex;
    f->rs = (w + 31) / 32 * 32; // UI/RenderWidget.cpp:235
    size_t nc = (size_t)f->cw * f->ch, nv = (size_t)f->rs * h;
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipMalloc((void **)&f->ext[k], nc * 4);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipMalloc((void **)&f->crop[k], (size_t)w * h * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&f->v, nv * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&f->u, nv * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&f->out, (size_t)w * h * 3);
    if (e == hipSuccess) e = hipMemsetAsync(f->v, 0, nv * 8, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(f->u, 0, nv * 8, c->stream);
    if (e != hipSuccess) {
        vm_frame_destroy(f);
        return vm_fail(VM_E_DEVICE, "vm_frame_create: %s", hipGetErrorString(e));
    }
    *out = f;
    return VM_OK;
}

extern "C" void vm_frame_destroy(vm_frame *f)
{
    if (!f) return;
    const bool alive = vm_ctx_alive(f->ctx); // destroyed after its context: freed without it (vm_api.cpp)
    VmDeviceGuard g(f->device);
    if (g.ok) {
        if (alive) hipStreamSynchronize(f->ctx->stream);
        else hipDeviceSynchronize();
        hipFree(f->ext[0]); hipFree(f->ext[1]);
        hipFree(f->crop[0]); hipFree(f->crop[1]);
        hipFree(f->v); hipFree(f->u); hipFree(f->out); hipFree(f->rgb_stage); hipFree(f->pws2[0]); hipFree(f->pws2[1]);
        (void)hipGetLastError();
    }
    delete f;
}

extern "C" int vm_frame_upload(vm_frame *f, const uint8_t *e0, const uint8_t *e1, const float *v,
                               const float *q)
{
    if (!f) return vm_fail(VM_E_INVALID, "vm_frame_upload: frame is NULL");
    if (!vm_ctx_alive(f->ctx)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__);
    VM_ON_DEVICE(f->ctx);
    hipStream_t s = f->ctx->stream;
    size_t nc = (size_t)f->cw * f->ch * 4;
    const uint8_t *e[2] = {e0, e1};
    for (int k = 0; k < 2; ++k)
        if (e[k]) {
            VM_HIP(hipMemcpyAsync(f->ext[k], e[k], nc, hipMemcpyHostToDevice, s));
            // the originals both sides' fills sample from (cloned before any solve)
            vm_poisson_launch_crop(f->crop[k], f->ext[k], f->w, f->h, f->ex, s);
        }
    if (v) VM_HIP(hipMemcpy2DAsync(f->v, (size_t)f->rs * 8, v, (size_t)f->w * 8, (size_t)f->w * 8, f->h, hipMemcpyHostToDevice, s));
    if (q) VM_HIP(hipMemcpy2DAsync(f->u, (size_t)f->rs * 8, q, (size_t)f->w * 8, (size_t)f->w * 8, f->h, hipMemcpyHostToDevice, s));
    else if (!f->u_zero) VM_HIP(hipMemsetAsync(f->u, 0, (size_t)f->rs * f->h * 8, s));
    f->u_zero = q == nullptr;
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

// The two frames as RGB8 (h rows of pitch_bytes; 0 = tight): the canvases are built on the device the way Pyramid::build
// builds them (pyramid.cu:186-200) -- 12 MB over the link per 1080p frame pair instead of the 27 MB of finished canvases.
// v and the quadratic path stay what they are.
extern "C" int vm_frame_upload_rgb(vm_frame *f, const uint8_t *rgb0, const uint8_t *rgb1, int pitch_bytes)
{
    if (!f || !rgb0 || !rgb1) return vm_fail(VM_E_INVALID, "vm_frame_upload_rgb: NULL argument");
    if (!vm_ctx_alive(f->ctx)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__);
    if (pitch_bytes == 0) pitch_bytes = 3 * f->w;
    if (pitch_bytes < 3 * f->w) return vm_fail(VM_E_INVALID, "vm_frame_upload_rgb: pitch < 3 * w");
    VM_ON_DEVICE(f->ctx);
    hipStream_t s = f->ctx->stream;
    const size_t one = (size_t)f->w * f->h * 3;
    if (!f->rgb_stage) VM_HIP(hipMalloc((void **)&f->rgb_stage, 2 * one));
    const uint8_t *src[2] = {rgb0, rgb1};
    for (int k = 0; k < 2; ++k) {
        VM_HIP(hipMemcpy2DAsync(f->rgb_stage + k * one, (size_t)3 * f->w, src[k], (size_t)pitch_bytes, (size_t)3 * f->w, f->h, hipMemcpyHostToDevice, s));
        vm_poisson_launch_canvas(f->ext[k], f->crop[k], f->rgb_stage + k * one, f->w, f->h, f->ex, s);
    }
    VM_HIP(hipGetLastError());
    VM_HIP(hipStreamSynchronize(s));        // the host buffers belong to the caller
    return VM_OK;
}

// (vm_render.hip) `workgroups` single-wave workgroups that each do nothing for `ticks` periods of the constant 100 MHz counter
void vm_launch_spin(unsigned long long ticks, int workgroups, hipStream_t s);

// Do the streams of two contexts of one device run SIDE BY SIDE?  The HIP runtime multiplexes a process's streams onto
// GPU_MAX_HW_QUEUES hardware queues; two streams that land on the same queue run their kernels one after the other,
// and which queue a new stream gets is not the caller's to choose.  A host that relies on two contexts overlapping (two
// compositor lanes, the solver streams of a batch job) asks here -- a 100 us do-nothing kernel on each stream, wall time
// of the pair against wall time of one -- and, when the answer is no, creates another context and
// asks again (keeping the rejected one alive until it has what it wants, so that the next stream gets another queue).
extern "C" int vm_dbg_streams_overlap(vm_ctx *a, vm_ctx *b, int *overlap)
{
    if (!a || !b || !overlap || a == b) return vm_fail(VM_E_INVALID, "vm_dbg_streams_overlap: bad argument");
    if (!vm_ctx_alive(a) || !vm_ctx_alive(b)) return vm_fail(VM_E_INVALID, "%s: a context was destroyed", __func__);
    if (a->device != b->device) return vm_fail(VM_E_INVALID, "vm_dbg_streams_overlap: the contexts live on different devices");
    VM_ON_DEVICE(a);
    std::lock_guard<std::recursive_mutex> la(a < b ? a->mu : b->mu), lb(a < b ? b->mu : a->mu);
    const unsigned long long ticks = 10000;          // 100 us at 100 MHz
    // host wall time of one kernel on a, and of one on each stream started back to back (the smallest of four tries each;
    // the first launch is also the warm-up of the code object): side by side the pair takes what one takes, on a shared
    // queue twice that
    double one = 1e30, both = 1e30;
    for (int rep = 0; rep < 8; ++rep) {
        const bool pair = (rep & 1) != 0;
        VM_HIP(hipStreamSynchronize(a->stream));
        VM_HIP(hipStreamSynchronize(b->stream));
        const auto t0 = std::chrono::steady_clock::now();
        vm_launch_spin(ticks, 1, a->stream);
        if (pair) vm_launch_spin(ticks, 1, b->stream);
        VM_HIP(hipStreamSynchronize(a->stream));
        VM_HIP(hipStreamSynchronize(b->stream));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rep >= 2) (pair ? both : one) = std::min(pair ? both : one, us);
    }
    // ... and DISPATCH side by side?  Two queues can overlap single workgroups and still take turns at dispatching (queues
    // of one pipe of the command processor): a small kernel on b then waits until a's big grid has been handed out.  A grid
    // of 64 K short workgroups on a (~ 8 rounds of the chip), one short workgroup on b right behind it: b's finishes early
    // if the two dispatch concurrently, with a's if they take turns.
    double frac = 0;
    for (int rep = 0; rep < 3; ++rep) {
        VM_HIP(hipStreamSynchronize(a->stream));
        VM_HIP(hipStreamSynchronize(b->stream));
        const auto t0 = std::chrono::steady_clock::now();
        vm_launch_spin(500, 65536, a->stream);
        vm_launch_spin(100, 1, b->stream);
        VM_HIP(hipStreamSynchronize(b->stream));
        const double tb = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        VM_HIP(hipStreamSynchronize(a->stream));
        const double ta = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        frac = std::max(frac, tb / ta);
    }
    VM_HIP(hipGetLastError());
    if (getenv("VM_DBG_OVERLAP")) fprintf(stderr, "vm_dbg_streams_overlap: one %.0f us, pair %.0f us, small-behind-big finishes at %.2f of the big one\n", one, both, frac);
    // (measured on MI355X: streams that run side by side 1.07 x and 0.30-0.37; the pairs that slow the compositor's two
    //  lanes from 2.4 to 3.2 ms per frame 1.43 x and 0.70)
    *overlap = (both < 1.25 * one && frac < 0.5) ? 1 : 0;
    return VM_OK;
}

// Page-lock a caller's host buffer (the reference keeps its frames in cv::Mat / QImage memory): uploads from it then
// run at the link's rate instead of through the runtime's staging copies (27 MB of canvases per 1080p frame).
extern "C" int vm_host_register(void *ptr, uint64_t bytes)
{
    if (!ptr || bytes == 0) return vm_fail(VM_E_INVALID, "vm_host_register: bad argument");
    (void)hipGetLastError();
    hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return VM_OK; }
    if (e != hipSuccess) { (void)hipGetLastError(); return vm_fail(VM_E_DEVICE, "vm_host_register: %s", hipGetErrorString(e)); }
    return VM_OK;
}

extern "C" int vm_host_unregister(void *ptr)
{
    if (!ptr) return vm_fail(VM_E_INVALID, "vm_host_unregister: NULL");
    hipError_t e = hipHostUnregister(ptr);
    (void)hipGetLastError();
    if (e != hipSuccess && e != hipErrorHostMemoryNotRegistered) return vm_fail(VM_E_DEVICE, "vm_host_unregister: %s", hipGetErrorString(e));
    return VM_OK;
}

extern "C" int vm_frame_download_ext(vm_frame *f, int side, uint8_t *ext)
{
    if (!f || !ext || (side != 1 && side != 2))
        return vm_fail(VM_E_INVALID, "vm_frame_download_ext: bad argument");
    if (!vm_ctx_alive(f->ctx)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__);
    VM_ON_DEVICE(f->ctx);
    hipStream_t s = f->ctx->stream;
    VM_HIP(hipMemcpyAsync(ext, f->ext[side - 1], (size_t)f->cw * f->ch * 4, hipMemcpyDeviceToHost, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}

extern "C" int vm_frame_set_v_from_level(vm_frame *f, vm_pyr *p, int lvl)
{
    if (!f || !p || lvl < 0 || lvl >= (int)p->lv.size())
        return vm_fail(VM_E_INVALID, "vm_frame_set_v_from_level: bad argument");
    if (!vm_ctx_alive(f->ctx) || !vm_ctx_alive(p->ctx)) return vm_fail(VM_E_INVALID, "%s: a context was destroyed", __func__);
    // a pyramid solved on ANOTHER context of the same device (a solver stream beside the compositor's): the frame's
    // stream waits for an EVENT of the solver's stream, the host never drains that stream -- a hipStreamSynchronize from
    // here could land inside the solver thread's graph capture (hipErrorStreamCaptureUnsupported) and would make the
    // compositor wait for whatever else that solver has queued since.  If nobody is inside a call on the pyramid's
    // context, a fresh event covers everything enqueued there so far; if a solver call is running (the NEXT job, on
    // other pyramids: a pyramid must not be read while it is being solved), the event its last completed call
    // recorded -- after the last write of this pyramid's levels -- is the one to wait for.  Another device is refused.
    if (p->ctx != f->ctx && p->ctx->device != f->ctx->device)
        return vm_fail(VM_E_INVALID, "frame and pyramid live on different devices (%d, %d)", f->ctx->device, p->ctx->device);
    VM_ON_DEVICE(f->ctx);
    if (p->ctx != f->ctx) {
        vm_ctx *pc = p->ctx;
        if (pc->mu.try_lock()) {
            hipError_t e = hipEventRecord(pc->xfer_ev, pc->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(f->ctx->stream, pc->xfer_ev, 0);
            pc->mu.unlock();
            if (e != hipSuccess) return vm_fail(VM_E_DEVICE, "vm_frame_set_v_from_level: %s", hipGetErrorString(e));
        } else {
            VM_HIP(hipStreamWaitEvent(f->ctx->stream, pc->done_ev, 0));
        }
    }
    vm_level &l = p->lv[lvl];
    vm_launch_upscale(f->v, f->w, f->h, f->rs, l.view.v, l.w, l.h, l.rs, f->ctx->stream);
    VM_HIP(hipGetLastError());
    return VM_OK;
}

extern "C" int vm_upscale_result(vm_pyr *p, int lvl, int w0, int h0, float *out, int pitch)
{
    if (!p || lvl < 0 || lvl >= (int)p->lv.size() || !out || w0 < 1 || h0 < 1)
        return vm_fail(VM_E_INVALID, "vm_upscale_result: bad argument");
    if (pitch == 0) pitch = 2 * w0;
    if (pitch < 2 * w0) return vm_fail(VM_E_INVALID, "vm_upscale_result: pitch < 2*w0");
    VM_ON_DEVICE(p->ctx);
    vm_level &l = p->lv[lvl];
    hipStream_t s = p->ctx->stream;
    float2 *tmp = nullptr;
    VM_HIP(hipMalloc((void **)&tmp, (size_t)w0 * h0 * 8));
    vm_launch_upscale(tmp, w0, h0, w0, l.view.v, l.w, l.h, l.rs, s);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(out, (size_t)pitch * 4, tmp, (size_t)w0 * 8, (size_t)w0 * 8, h0, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(tmp);
    if (e != hipSuccess) return vm_fail(VM_E_DEVICE, "vm_upscale_result: %s", hipGetErrorString(e));
    return VM_OK;
}

static int render_dev(vm_frame *f, float color_fa, float geo_fa, int color_from, float *ms)
{
    if (!f) return vm_fail(VM_E_INVALID, "vm_render_halfway: frame is NULL");
    if (color_from < 0 || color_from > 2) return vm_fail(VM_E_INVALID, "vm_render_halfway: color_from %d", color_from);
    vm_ctx *c = f->ctx;
    VM_ON_DEVICE(c);
    if (ms) VM_HIP(hipEventRecord(c->ev0, c->stream));
    vm_launch_render(f->out, f->w * 3, f->w, f->h, f->rs, f->ex, color_fa, geo_fa, color_from,
                     f->ext[0], f->ext[1], f->v, f->u_zero ? nullptr : f->u, c->stream);
    VM_HIP(hipGetLastError());
    if (ms) {
        VM_HIP(hipEventRecord(c->ev1, c->stream));
        VM_HIP(hipEventSynchronize(c->ev1));
        VM_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
    }
    return VM_OK;
}

extern "C" int vm_render_halfway_dev(vm_frame *f, float color_fa, float geo_fa, int color_from, float *ms)
{
    return render_dev(f, color_fa, geo_fa, color_from, ms);
}

extern "C" int vm_render_halfway(vm_frame *f, float color_fa, float geo_fa, int color_from,
                                 uint8_t *rgb, int pitch)
{
    if (!rgb) return vm_fail(VM_E_INVALID, "vm_render_halfway: output is NULL");
    int rc = render_dev(f, color_fa, geo_fa, color_from, nullptr);
    if (rc != VM_OK) return rc;
    if (pitch == 0) pitch = f->w * 3;
    if (pitch < f->w * 3) return vm_fail(VM_E_INVALID, "vm_render_halfway: pitch < 3*w");
    if (!vm_ctx_alive(f->ctx)) return vm_fail(VM_E_INVALID, "%s: the context was destroyed", __func__);
    VM_ON_DEVICE(f->ctx);
    hipStream_t s = f->ctx->stream;
    VM_HIP(hipMemcpy2DAsync(rgb, pitch, f->out, (size_t)f->w * 3, (size_t)f->w * 3, f->h, hipMemcpyDeviceToHost, s));
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}
