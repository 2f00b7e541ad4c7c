// vmorph/render.hpp -- C++ host mirror of the compositor entry points:
// render_halfway_image (UI/RenderWidget.h:52-57, Algorithm/render.cu:62-96) and
// CPoissonExt (Algorithm/PoissonExt.h), reference repository, on device-resident
// frames.
#ifndef VMORPH_RENDER_HPP
#define VMORPH_RENDER_HPP

#include "pyramid.hpp"

namespace vmorph {

// extended RGBA8 canvas of Pyramid::build, pyramid.cu:186-200
inline std::vector<unsigned char> make_extended(const unsigned char *rgb, int w, int h, int ex)
{
    int cw = w + 2 * ex, ch = h + 2 * ex;
    std::vector<unsigned char> can((size_t)cw * ch * 4, 255);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            unsigned char *d = &can[((size_t)(y + ex) * cw + x + ex) * 4];
            const unsigned char *s = &rgb[((size_t)y * w + x) * 3];
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 0;
        }
    return can;
}

class Frame {
public:
    Frame(Context &ctx, int w, int h, int ex) : w_(w), h_(h), ex_(ex) { check(vm_frame_create(ctx.handle(), w, h, ex, &f_)); }
    ~Frame() { vm_frame_destroy(f_); }
    Frame(const Frame &) = delete;
    Frame &operator=(const Frame &) = delete;

    void upload(const unsigned char *ext0, const unsigned char *ext1, const float *v, const float *qpath)
    {
        check(vm_frame_upload(f_, ext0, ext1, v, qpath));
    }
    // the canvases built on the device from the two RGB8 frames (Pyramid::build, pyramid.cu:186-200)
    void upload_rgb(const unsigned char *rgb0, const unsigned char *rgb1, int pitch_bytes = 0) { check(vm_frame_upload_rgb(f_, rgb0, rgb1, pitch_bytes)); }
    void set_v_from_level(Pyramid &pyr, int el) { check(vm_frame_set_v_from_level(f_, pyr.handle(), el - 1)); }

    // CPoissonExt::run body for one side, PoissonExt.cpp:19-41
    int poisson_extend(int side, float tol = 1e-5f, int max_it = 20000)
    {
        int it = 0;
        check(vm_poisson_extend(f_, side, tol, max_it, &it, nullptr, nullptr));
        return it;
    }
    // CQuadraticPath::optimize for this frame (QuadraticPath.cpp:24-223): u stays in the frame
    int quadratic_path(float tol = 1e-4f, int max_it = 200)
    {
        int it = 0;
        check(vm_frame_quadratic_path(f_, tol, max_it, &it, nullptr, nullptr));
        return it;
    }
    std::vector<float> download_qpath()
    {
        std::vector<float> out((size_t)w_ * h_ * 2);
        check(vm_frame_download_qpath(f_, out.data()));
        return out;
    }
    std::vector<unsigned char> download_ext(int side)
    {
        std::vector<unsigned char> out((size_t)(w_ + 2 * ex_) * (h_ + 2 * ex_) * 4);
        check(vm_frame_download_ext(f_, side, out.data()));
        return out;
    }
    // render_halfway_image(out, rowstride, w, h, ex, color_fa, geo_fa, color_from, ...)
    std::vector<unsigned char> render_halfway_image(float color_fa, float geo_fa, int color_from)
    {
        std::vector<unsigned char> out((size_t)w_ * h_ * 3);
        check(vm_render_halfway(f_, color_fa, geo_fa, color_from, out.data(), 0));
        return out;
    }

    vm_frame *handle() const { return f_; }

private:
    vm_frame *f_ = nullptr;
    int w_, h_, ex_;
};

// CPoissonExt::run's loop body (PoissonExt.cpp:24-36) for several frames of one context at once: both sides of every
// frame are independent systems and share every launch (vm_poisson_extend_frames).  Returns the PCG iteration counts,
// [2 i] = side 1 of frames[i], [2 i + 1] = side 2.
inline std::vector<int> poisson_extend_frames(const std::vector<Frame *> &frames, float tol = 1e-5f, int max_it = 20000)
{
    std::vector<vm_frame *> h;
    for (Frame *f : frames) h.push_back(f->handle());
    std::vector<int> its(2 * h.size(), 0);
    check(vm_poisson_extend_frames(h.data(), (int)h.size(), tol, max_it, its.data(), nullptr, nullptr));
    return its;
}

} // namespace vmorph
#endif
