// vmorph/video.hpp -- C++ host mirror of the reference's class Pyramid with depth > 1
// (Algorithm/Pyramid.h:14-98: per level `depth` pages plus the flow fields f0/f1/b0/b1 of every
// page) and of class Morph over it (Algorithm/morph.h:10-31): the temporally coupled solve of a
// video pair.  Device storage and the page schedule live behind the C-ABI (vm_video_*).
#ifndef VMORPH_VIDEO_HPP
#define VMORPH_VIDEO_HPP

#include <cmath>
#include <vector>

#include "pyramid.hpp"

namespace vmorph {

// one level of the table Pyramid::build walks (pyramid.cu:238-469)
struct VideoLevel { int width, height, depth, factor_t; };

class VideoPyramid {
public:
    explicit VideoPyramid(Context &ctx) : ctx_(ctx) {}
    ~VideoPyramid() { clear(); }
    VideoPyramid(const VideoPyramid &) = delete;
    VideoPyramid &operator=(const VideoPyramid &) = delete;

    void clear()
    {
        if (h_) { vm_video_destroy(h_); h_ = nullptr; }
        levels.clear();
    }

    // The level table of the stage-2 Pyramid::build (pyramid.cu:223-240, 462-477): w, h halve
    // (ceil) on every level, the depth halves (ceil((d+1)/2)) on the coarsest el_t levels.
    // Level counts in integer arithmetic (the reference truncates float32 logarithms); the
    // 14 Mvoxel decimation in float32 as written there.
    static std::vector<VideoLevel> level_table(int w, int h, int d, int start_res, float max_voxels = 14e6f)
    {
        float fa = std::sqrt((float)(w * h * d) / max_voxels);
        if (fa < 1.0f) fa = 1.0f;
        w = (int)((float)w / fa);
        h = (int)((float)h / fa);
        auto el = [&](int dim) { int n = 0; while (dim >= start_res) { dim /= 2; ++n; } return n; };
        const int el_t = el(d), el_xy = std::max(el(w), el(h)), maxl = std::max(el_xy, el_t);
        std::vector<VideoLevel> out;
        int factor_t = 1;
        for (int k = 0; k < maxl; ++k) {
            out.push_back({w, h, d, factor_t});
            if (maxl - k <= el_xy) { w = (w + 1) / 2; h = (h + 1) / 2; }
            if (maxl - k <= el_t) { d = (d + 2) / 2; factor_t = 2; } else factor_t = 1;
        }
        return out;
    }

    // Pyramid::append_new for every level
    void create(const std::vector<VideoLevel> &table, int depth0)
    {
        clear();
        std::vector<int> w, h, d, ft;
        for (const VideoLevel &l : table) { w.push_back(l.width); h.push_back(l.height); d.push_back(l.depth); ft.push_back(l.factor_t); }
        check(vm_video_create(ctx_.handle(), (int)table.size(), w.data(), h.data(), d.data(), ft.data(), depth0, &h_));
        levels = table;
    }

    // Pyramid::build(video0, video1, f0, f1, b0, b1, start_res), pyramid.cu:166-485: RGB8 frames
    // (h*w*3 bytes each) and full-resolution flows (h*w*2 floats each) of every frame
    void build(const std::vector<const unsigned char *> &video0, const std::vector<const unsigned char *> &video1,
               const std::vector<const float *> &f0, const std::vector<const float *> &f1,
               const std::vector<const float *> &b0, const std::vector<const float *> &b1, int w, int h, int start_res)
    {
        const int d = (int)video0.size();
        create(level_table(w, h, d, start_res), d);
        for (int t = 0; t < d; ++t)
            check(vm_video_build_rgb(h_, t, video0[t], video1[t], 0));
        check(vm_video_build_flows(h_, f0.data(), f1.data(), b0.data(), b1.data()));
    }

    std::vector<float> get_v(int lvl, int page) const
    {
        std::vector<float> v((size_t)levels[lvl].width * levels[lvl].height * 2);
        check(vm_video_get_v(h_, lvl, page, v.data(), 0));
        return v;
    }

    vm_video *handle() const { return h_; }
    Context &context() const { return ctx_; }
    std::vector<VideoLevel> levels; // level 0 finest ... back(): coarsest (v only)

private:
    Context &ctx_;
    vm_video *h_ = nullptr;
};

// class Morph over a video pair: middle page first, then both chains with the temporal term
class VideoMorph {
public:
    VideoMorph(Parameters &params, VideoPyramid &pyramid, volatile int &run_flag, bool fixed_work = false)
        : m_cb(run_flag), m_pyramid(pyramid), m_params(params), fixed_work_(fixed_work) {}

    // morph.cu:150-168 with the page schedule of optimize_level (:1353-1441)
    bool calculate_halfway_parametrization()
    {
        KernParameters kp(m_params);
        check(vm_set_params(m_pyramid.context().handle(), &kp));
        std::vector<vm_video_constraint> cons; // morph.cu:354-366, every frame
        for (const auto &row : m_params.cnt)
            for (const Connect &c : row) {
                const Conp &l = m_params.lp.at(c.li.x).at(c.li.y);
                const Conp &r = m_params.rp.at(c.ri.x).at(c.ri.y);
                cons.push_back(vm_video_constraint{(float)l.p.x, (float)l.p.y, (float)r.p.x, (float)r.p.y,
                                                   std::min(l.weight, r.weight), l.p.z});
            }
        size_t total = 0;
        for (size_t l = 0; l + 1 < m_pyramid.levels.size(); ++l) total += m_pyramid.levels[l].depth;
        progress.assign(total, vm_progress{});
        int rc = vm_video_solve(m_pyramid.handle(), (float)m_params.max_iter, m_params.max_iter_drop_factor,
                                cons.data(), (int)cons.size(), &m_cb, fixed_work_ ? 1 : 0, progress.data());
        if (rc != VM_E_CANCELLED) check(rc);
        return true;
    }

    std::vector<vm_progress> progress; // levels with images finest first, pages within a level

private:
    volatile int &m_cb;
    VideoPyramid &m_pyramid;
    Parameters &m_params;
    bool fixed_work_;
};

} // namespace vmorph
#endif
