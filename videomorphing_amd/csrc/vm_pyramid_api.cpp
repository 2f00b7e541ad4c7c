// vm_pyramid_api.cpp -- C-ABI of the device-side pyramid builder: the image half of
// Pyramid::build (Algorithm/pyramid.cu:166-485) for one frame pair, driving
// vm_pyramid.hip the way pyramid.cu drives include/resample's scale().
#include "vm_host.h"
#include "vm_pyramid.h"
#include "vm_temporal.h"

#include <cmath>
#include <map>
#include <vector>

namespace {

float bspline3(float r)
{
    r = std::fabs(r);
    if (r < 1.f) return (4.f + r * r * (-6.f + 3.f * r)) / 6.f;
    if (r < 2.f) return (8.f + r * (-12.f + (6.f - r) * r)) / 6.f;
    return 0.f;
}
int ext_mirror(int i, int n)
{
    const int m = 2 * n;
    i = i >= 0 ? i % m : (m - 1) - ((-i - 1) % m);
    return i >= n ? m - i - 1 : i;
}

// the factored inverse of the sampled cubic B-spline with mirror boundary
// (dlti.cpp:232-257 assembly, :66-93 LU without pivoting, float), as three arrays:
// [0,n) upper A(j-1? no: A(i,i+1) stored at column i+1), [n,2n) inverse pivots, [2n,3n) lower A(i+1,i)
std::vector<float> tri_factor(int n)
{
    std::vector<float> A((size_t)3 * n, 0.f);
    auto at = [&](int i, int j) -> float & { return A[(size_t)(i - j + 1) * n + j]; };
    const float kern[3] = {bspline3(1.f), bspline3(0.f), bspline3(-1.f)};
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            at(i, ext_mirror(i + k - 1, n)) += kern[k];
    for (int p = 0; p < n; ++p) {
        float inv_p = (at(p, p) = 1.f / at(p, p));
        if (p + 1 < n) {
            float m = (at(p + 1, p) *= inv_p);
            at(p + 1, p + 1) -= m * at(p, p + 1);
        }
    }
    return A;
}

struct Builder {
    vm_ctx *c;
    std::map<int, float *> factors; // line length -> device factors
    float *img = nullptr, *tmp = nullptr;
    uint8_t *rgb = nullptr;
    ~Builder()
    {
        for (auto &kv : factors) hipFree(kv.second);
        hipFree(img); hipFree(tmp); hipFree(rgb);
    }
    int factor(int n, float **out)
    {
        auto it = factors.find(n);
        if (it == factors.end()) {
            std::vector<float> A = tri_factor(n);
            float *d = nullptr;
            VM_HIP(hipMalloc((void **)&d, A.size() * 4));
            VM_HIP(hipMemcpyAsync(d, A.data(), A.size() * 4, hipMemcpyHostToDevice, c->stream));
            VM_HIP(hipStreamSynchronize(c->stream));
            it = factors.emplace(n, d).first;
        }
        *out = it->second;
        return VM_OK;
    }
    // one axis of scale(): src (w x h, 3 planes) -> dst; returns the new size in w, h
    int axis(float *&src, float *&dst, int &w, int &h, int nout, int ax)
    {
        const int nin = ax == 0 ? w : h;
        const int wout = ax == 0 ? nout : w, hout = ax == 0 ? h : nout;
        float *A = nullptr;
        hipStream_t s = c->stream;
        if (nout < nin) {
            vm_pyr_launch_down(src, dst, w, h, nout, ax, s);
            int rc = factor(nout, &A);
            if (rc != VM_OK) return rc;
            vm_pyr_launch_tri_solve(dst, A, wout, hout, ax, s);
        } else {
            vm_pyr_launch_curve(src, (size_t)3 * w * h, 1, s);
            int rc = factor(nin, &A);
            if (rc != VM_OK) return rc;
            vm_pyr_launch_tri_solve(src, A, w, h, ax, s);
            vm_pyr_launch_up(src, dst, w, h, nout, ax, s);
            vm_pyr_launch_curve(dst, (size_t)3 * wout * hout, 0, s);
        }
        std::swap(src, dst);
        w = wout;
        h = hout;
        return VM_OK;
    }
    // scale(), scale.cpp:225-272
    int scale(float *&a, float *&b, int &w, int &h, int wout, int hout)
    {
        int rc;
        if (hout * w < wout * h) {
            if ((rc = axis(a, b, w, h, hout, 1)) != VM_OK) return rc;
            return axis(a, b, w, h, wout, 0);
        }
        if ((rc = axis(a, b, w, h, wout, 0)) != VM_OK) return rc;
        return axis(a, b, w, h, hout, 1);
    }
};

} // namespace

extern "C" int vm_pyramid_build_rgb(vm_pyr *p, const uint8_t *rgb0, const uint8_t *rgb1, int pitch)
{
    if (!p || !rgb0 || !rgb1) return vm_fail(VM_E_INVALID, "vm_pyramid_build_rgb: NULL argument");
    std::lock_guard<std::recursive_mutex> lock(p->ctx->mu);
    vm_ctx *c = p->ctx;
    VM_ON_DEVICE(c);
    const int w0 = p->lv[0].w, h0 = p->lv[0].h, L = (int)p->lv.size();
    if (pitch == 0) pitch = 3 * w0;
    if (pitch < 3 * w0) return vm_fail(VM_E_INVALID, "vm_pyramid_build_rgb: pitch < 3*width");
    Builder B;
    B.c = c;
    const size_t n0 = (size_t)w0 * h0;
    VM_HIP(hipMalloc((void **)&B.img, n0 * 12));
    VM_HIP(hipMalloc((void **)&B.tmp, n0 * 12));
    VM_HIP(hipMalloc((void **)&B.rgb, (size_t)pitch * h0));
    hipStream_t s = c->stream;
    const uint8_t *src[2] = {rgb0, rgb1};
    for (int k = 0; k < 2; ++k) {
        VM_HIP(hipMemcpyAsync(B.rgb, src[k], (size_t)pitch * h0, hipMemcpyHostToDevice, s));
        float *a = B.img, *b = B.tmp;
        int w = w0, h = h0;
        vm_pyr_launch_load(B.rgb, pitch, a, w, h, s);
        for (int el = 0; el < L - 1; ++el) { // the coarsest level holds no images (pyramid.cu:329)
            const vm_level &lv = p->lv[el];
            int rc = B.scale(a, b, w, h, lv.w, lv.h); // el == 0: same size (pyramid.cu:270-273)
            if (rc != VM_OK) return rc;
            vm_pyr_launch_store_gray(a, (float *)(k == 0 ? lv.view.img0 : lv.view.img1), lv.w, lv.h, lv.rs, s);
        }
        VM_HIP(hipGetLastError());
        VM_HIP(hipStreamSynchronize(s)); // B.rgb is reused for the second frame
    }
    return VM_OK;
}


// frame of the video a page shows: page t of level l is scaled from page
// min(t * factor_t, prev_d - 1) of level l-1 (pyramid.cu:363-364)
static int page_frame(const vm_video *v, int lvl, int t)
{
    for (int l = lvl; l > 0; --l)
        t = std::min(t * v->factor_t[l], v->depth[l - 1] - 1);
    return t;
}

// The image half of Pyramid::build for ONE frame of a video pair: the frame's luma pyramid
// goes to every page that shows this frame (a page of a coarser level of the temporal
// pyramid stands for the frame min(t * factor_t, prev_d - 1) of the level above).
extern "C" int vm_video_build_rgb(vm_video *v, int frame, const uint8_t *rgb0, const uint8_t *rgb1, int pitch)
{
    if (!v || !rgb0 || !rgb1) return vm_fail(VM_E_INVALID, "vm_video_build_rgb: NULL argument");
    if (frame < 0 || frame >= v->depth[0]) return vm_fail(VM_E_INVALID, "vm_video_build_rgb: frame %d out of range", frame);
    vm_ctx *c = v->ctx;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    VM_ON_DEVICE(c);
    const int w0 = v->pages[0][0].lv.w, h0 = v->pages[0][0].lv.h, L = (int)v->pages.size();
    if (pitch == 0) pitch = 3 * w0;
    if (pitch < 3 * w0) return vm_fail(VM_E_INVALID, "vm_video_build_rgb: pitch < 3*width");
    Builder B;
    B.c = c;
    const size_t n0 = (size_t)w0 * h0;
    VM_HIP(hipMalloc((void **)&B.img, n0 * 12));
    VM_HIP(hipMalloc((void **)&B.tmp, n0 * 12));
    VM_HIP(hipMalloc((void **)&B.rgb, (size_t)pitch * h0));
    hipStream_t s = c->stream;
    const uint8_t *src[2] = {rgb0, rgb1};
    for (int k = 0; k < 2; ++k) {
        VM_HIP(hipMemcpyAsync(B.rgb, src[k], (size_t)pitch * h0, hipMemcpyHostToDevice, s));
        float *a = B.img, *b = B.tmp;
        int w = w0, h = h0;
        vm_pyr_launch_load(B.rgb, pitch, a, w, h, s);
        for (int el = 0; el < L - 1; ++el) {
            const vm_level &l0 = v->pages[el][0].lv;
            int rc = B.scale(a, b, w, h, l0.w, l0.h);
            if (rc != VM_OK) return rc;
            for (int t = 0; t < v->depth[el]; ++t)
                if (page_frame(v, el, t) == frame) {
                    const vm_level &lv = v->pages[el][t].lv;
                    vm_pyr_launch_store_gray(a, (float *)(k == 0 ? lv.view.img0 : lv.view.img1), lv.w, lv.h, lv.rs, s);
                }
        }
        VM_HIP(hipGetLastError());
        VM_HIP(hipStreamSynchronize(s));
    }
    return VM_OK;
}

// The flow half of Pyramid::build (pyramid.cu:284-326, 375-456) on the device: the four flow
// families of all depth0 frames (full resolution, tight (h0, w0, 2) floats) are scaled level by
// level through load(-50, 50) -> scale() -> store, rescaled by the size ratio, concatenated in
// time where the temporal pyramid halves the depth, and left in the pages' flow arrays.
extern "C" int vm_video_build_flows(vm_video *v, const float *const *f0, const float *const *f1,
                                    const float *const *b0, const float *const *b1)
{
    if (!v || !f0 || !f1 || !b0 || !b1) return vm_fail(VM_E_INVALID, "vm_video_build_flows: NULL argument");
    vm_ctx *c = v->ctx;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    VM_ON_DEVICE(c);
    hipStream_t s = c->stream;
    const int L = (int)v->pages.size(), d0 = v->depth[0];
    const int w0 = v->pages[0][0].lv.w, h0 = v->pages[0][0].lv.h;
    const size_t n0 = (size_t)w0 * h0;
    Builder B;
    B.c = c;
    VM_HIP(hipMalloc((void **)&B.img, n0 * 12));
    VM_HIP(hipMalloc((void **)&B.tmp, n0 * 12));
    // working set: the current level's flows of every frame of the previous level, tight float2
    struct Free { std::vector<void *> p; ~Free() { for (void *q : p) hipFree(q); } } guard;
    std::vector<float2 *> cur[4];
    const float *const *src[4] = {f0, f1, b0, b1};
    for (int k = 0; k < 4; ++k) {
        cur[k].resize(d0);
        for (int t = 0; t < d0; ++t) {
            if (!src[k][t]) return vm_fail(VM_E_INVALID, "vm_video_build_flows: flow %d of frame %d is NULL", k, t);
            VM_HIP(hipMalloc((void **)&cur[k][t], n0 * 8));
            guard.p.push_back(cur[k][t]);
            VM_HIP(hipMemcpyAsync(cur[k][t], src[k][t], n0 * 8, hipMemcpyHostToDevice, s));
        }
    }
    VM_HIP(hipStreamSynchronize(s)); // the host arrays belong to the caller
    int pw = w0, ph = h0, prev_d = d0;
    for (int el = 0; el < L - 1; ++el) { // the coarsest level holds no flows (pyramid.cu:329)
        const int w = v->pages[el][0].lv.w, h = v->pages[el][0].lv.h, d = v->depth[el], ft = v->factor_t[el];
        const float ratiox = (float)w / (float)pw, ratioy = (float)h / (float)ph;
        for (int k = 0; k < 4; ++k)
            for (int t = 0; t < prev_d; ++t) {
                float *a = B.img, *b = B.tmp;
                int cw = pw, ch = ph;
                vm_flow_launch_load(cur[k][t], pw, a, pw, ph, s);
                int rc = B.scale(a, b, cw, ch, w, h);
                if (rc != VM_OK) return rc;
                vm_flow_launch_store(a, cur[k][t], w, w, h, ratiox, ratioy, s); // tight at the new size
            }
        if (el > 0 && ft > 1) {
            for (int t = 0; t < d; ++t) {
                if (t * ft > prev_d - 1) continue;
                if (t * ft + 1 < prev_d)
                    for (int k = 0; k < 2; ++k) vm_flow_launch_concat(cur[k][t * ft], cur[k][t * ft + 1], w, w, h, s);
                if (t > 0)
                    for (int k = 2; k < 4; ++k) vm_flow_launch_concat(cur[k][t * ft], cur[k][t * ft - 1], w, w, h, s);
            }
            for (int k = 0; k < 4; ++k)
                for (int t = 0; t < d; ++t) // forw0[t] = forw0[min(t*factor_t, prev_d-1)] (source index >= t)
                    cur[k][t] = cur[k][std::min(t * ft, prev_d - 1)];
        }
        for (int k = 0; k < 4; ++k)
            for (int t = 0; t < d; ++t) {
                const vm_video_page &pg = v->pages[el][t];
                VM_HIP(hipMemcpy2DAsync(pg.flow[k], (size_t)pg.lv.rs * 8, cur[k][t], (size_t)w * 8, (size_t)w * 8, h, hipMemcpyDeviceToDevice, s));
            }
        VM_HIP(hipGetLastError());
        pw = w; ph = h; prev_d = d;
    }
    VM_HIP(hipStreamSynchronize(s));
    return VM_OK;
}
