// vmorph/pyramid.hpp -- C++ host mirror of Algorithm/Pyramid.h (reference
// repository): class Pyramid / struct PyramidLevel with the reference's public
// members, device storage behind the C-ABI.  One Pyramid = one frame pair
// (depth 1).  pyramid[0] is the full-resolution placeholder (pyramid.cu:220),
// pyramid[1..size()-1] run finest to coarsest, the last level holds no images.
#ifndef VMORPH_PYRAMID_HPP
#define VMORPH_PYRAMID_HPP

#include <memory>
#include <vector>

#include "parameters.hpp"

namespace vmorph {

class Context {
public:
    explicit Context(int device = 0, int math_mode = VM_MATH_EXACT)
    {
        check(vm_ctx_create(device, &h_));
        check(vm_set_math_mode(h_, math_mode));
    }
    ~Context() { vm_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    vm_ctx *handle() const { return h_; }
    void set_math_mode(int m) { check(vm_set_math_mode(h_, m)); }
    void sync() { check(vm_ctx_sync(h_)); }
private:
    vm_ctx *h_ = nullptr;
};

class Pyramid;

// struct PyramidLevel, Pyramid.h:52-98 (ctor: pyramid.cu:531-543)
struct PyramidLevel {
    PyramidLevel(Pyramid *pyr, int el, int w, int h)
        : width(w), height(h), depth(1), pyr_(pyr), el_(el)
    {
        rowstride = (w + 31) / 32 * 32;
        pagestride = rowstride * h;
        inv_wh = 1.0f / (w * h);
        impmask_rowstride = (w + 4) / 5 + 2;
        impmask_pagestride = impmask_rowstride * ((h + 4) / 5 + 2);
        factor_d = 1.0f;
    }
    int rowstride, pagestride, impmask_rowstride, impmask_pagestride;
    int width, height, depth;
    float factor_d, inv_wh;

    // the device array `v` of the level, tight (h, w, 2)
    std::vector<float> get_v() const;
    void set_v(const std::vector<float> &v);
    // any state array (VM_F_*), tight rows
    std::vector<float> field(int id) const;

private:
    Pyramid *pyr_;
    int el_;
};

class Pyramid {
public:
    explicit Pyramid(Context &ctx) : ctx_(ctx) {}
    ~Pyramid() { clear(); }
    Pyramid(const Pyramid &) = delete;
    Pyramid &operator=(const Pyramid &) = delete;

    PyramidLevel &operator[](int idx) { return *m_data[idx]; }
    const PyramidLevel &operator[](int idx) const { return *m_data[idx]; }
    PyramidLevel &back() { return *m_data.back(); }
    size_t size() const { return m_data.size(); }

    // Pyramid::clear, pyramid.cu:19-49
    void clear()
    {
        if (h_) { vm_pyramid_destroy(h_); h_ = nullptr; }
        m_data.clear();
        _vector.clear(); _qpath.clear(); _extends1.clear(); _extends2.clear();
    }

    // level count of pyramid.cu:230-240 in integer arithmetic
    static int num_levels(int w, int h, int start_res)
    {
        auto el = [&](int dim) { int n = 1; while (dim / 2 >= start_res) { dim /= 2; ++n; } return n; };
        return std::max(el(w), el(h));
    }

    // Pyramid::build(video0, video1, ..., start_res), pyramid.cu:166-485, for one pair
    // of float luma images (w*h, [0,255]).  Geometry: ceil halving (pyramid.cu:466-467);
    // coarser images: 2x2 box filter (the reference's Nehab-Hoppe scale() is a "next" row).
    void build(const float *img0, const float *img1, int w, int h, int start_res, int nlevels = 0)
    {
        clear();
        int n = nlevels > 0 ? nlevels : num_levels(w, h, start_res);
        n = std::max(n, 2);
        std::vector<int> ws(n), hs(n);
        ws[0] = w; hs[0] = h;
        for (int k = 1; k < n; ++k) { ws[k] = (ws[k - 1] + 1) / 2; hs[k] = (hs[k - 1] + 1) / 2; }
        check(vm_pyramid_create(ctx_.handle(), n, ws.data(), hs.data(), &h_));
        m_data.emplace_back(new PyramidLevel(this, 0, w, h));
        for (int k = 0; k < n; ++k) m_data.emplace_back(new PyramidLevel(this, k + 1, ws[k], hs[k]));
        std::vector<float> a(img0, img0 + (size_t)w * h), b(img1, img1 + (size_t)w * h);
        for (int k = 0; k < n - 1; ++k) {
            check(vm_level_upload_luma(h_, k, a.data(), b.data(), 0));
            if (k + 1 < n - 1) { a = down2(a, ws[k], hs[k]); b = down2(b, ws[k], hs[k]); }
        }
        _vector.assign(1, std::vector<float>((size_t)w * h * 2, 0.0f));
        _qpath.assign(1, std::vector<float>((size_t)w * h * 2, 0.0f));
    }

    vm_pyr *handle() const { return h_; }
    Context &context() const { return ctx_; }

    // Pyramid.h:41-47 (cv::Mat replaced by flat arrays): full-resolution results
    std::vector<std::vector<float>> _vector, _qpath;
    std::vector<std::vector<unsigned char>> _extends1, _extends2, _results;

private:
    static std::vector<float> down2(const std::vector<float> &s, int w, int h)
    {
        int W = (w + 1) / 2, H = (h + 1) / 2;
        std::vector<float> d((size_t)W * H);
        auto at = [&](int x, int y) { return s[(size_t)std::min(y, h - 1) * w + std::min(x, w - 1)]; };
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                d[(size_t)y * W + x] = 0.25f * (at(2 * x, 2 * y) + at(2 * x, 2 * y + 1) + at(2 * x + 1, 2 * y) + at(2 * x + 1, 2 * y + 1));
        return d;
    }
    Context &ctx_;
    vm_pyr *h_ = nullptr;
    std::vector<std::unique_ptr<PyramidLevel>> m_data;
};

inline std::vector<float> PyramidLevel::get_v() const
{
    std::vector<float> v((size_t)width * height * 2);
    check(vm_level_get_v(pyr_->handle(), el_ - 1, v.data(), 0));
    return v;
}
inline void PyramidLevel::set_v(const std::vector<float> &v)
{
    check(vm_level_set_v(pyr_->handle(), el_ - 1, v.data(), 0));
}
inline std::vector<float> PyramidLevel::field(int id) const
{
    bool two = id == VM_F_V || id == VM_F_LUMA || id == VM_F_MEAN || id == VM_F_VAR || id == VM_F_TPS_B || id == VM_F_UI_B;
    std::vector<float> out((size_t)width * height * (two ? 2 : 1));
    check(vm_level_get_field(pyr_->handle(), el_ - 1, id, out.data()));
    return out;
}

} // namespace vmorph
#endif
