// vmorph/video.hpp -- C++ host mirror of the reference's class Pyramid with depth > 1
// (Algorithm/Pyramid.h:14-98: per level `depth` pages plus the flow fields f0/f1/b0/b1 of every
// page) and of class Morph over it (Algorithm/morph.h:10-31): the temporally coupled solve of a
// video pair.  Device storage and the page schedule live behind the C-ABI (vm_video_*).
#ifndef VMORPH_VIDEO_HPP
#define VMORPH_VIDEO_HPP

#include <chrono>
#include <cmath>
#include <exception>
#include <thread>
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
        depth0_ = depth0;
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
    int depth0() const { return depth0_; }
    std::vector<VideoLevel> levels; // level 0 finest ... back(): coarsest (v only)
    // Pyramid::_vector (Pyramid.h:41): one full-resolution field per frame of the video, written by
    // VideoMatchingThread::update_result
    std::vector<std::vector<float>> _vector;

private:
    Context &ctx_;
    vm_video *h_ = nullptr;
    int depth0_ = 1;
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

// class CMatchingThread (MatchingThread.h:7-37) over a video pair: the coupled solve on a worker
// thread (QThread -> std::thread), then update_result(): pyramid._vector[frame] for EVERY frame of the
// video at w0 x h0 -- pages scaled and resized, the frames the temporal pyramid skipped blended from
// their neighbours (MatchingThread.cpp:22-84)
class VideoMatchingThread {
public:
    VideoMatchingThread(Parameters &parameters, VideoPyramid &pyramids, int w0 = 0, int h0 = 0)
        : runflag(1), _pyramids(pyramids), _parameters(parameters), gpu_morph(parameters, pyramids, runflag),
          w0_(w0 ? w0 : pyramids.levels.at(0).width), h0_(h0 ? h0 : pyramids.levels.at(0).height) {}
    ~VideoMatchingThread()
    {
        if (thread_.joinable()) thread_.join(); // a stored exception is dropped: wait() shows it
    }

    void run() // MatchingThread.cpp:138-150
    {
        auto t0 = std::chrono::steady_clock::now();
        gpu_morph.calculate_halfway_parametrization();
        run_time = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
        update_result();
    }
    void start() { thread_ = std::thread([this] { try { run(); } catch (...) { error_ = std::current_exception(); } }); }
    void wait()
    {
        if (thread_.joinable()) thread_.join();
        if (error_) { auto e = error_; error_ = nullptr; std::rethrow_exception(e); }
    }

    // MatchingThread.cpp:22-84 from level `lvl` (after run(): the finest)
    void update_result(int lvl = 0)
    {
        const size_t n = (size_t)w0_ * h0_ * 2;
        std::vector<float> all(n * _pyramids.depth0());
        check(vm_video_result(_pyramids.handle(), lvl, w0_, h0_, all.data()));
        _pyramids._vector.assign(_pyramids.depth0(), std::vector<float>());
        for (int f = 0; f < _pyramids.depth0(); ++f)
            _pyramids._vector[f].assign(all.begin() + f * n, all.begin() + (f + 1) * n);
        percentage = 100.0f;
    }

    float percentage = 0.0f;
    float run_time = 0.0f;
    volatile int runflag; // the reference's `bool runflag`, written by the UI thread

private:
    VideoPyramid &_pyramids;
    Parameters &_parameters;
public:
    VideoMorph gpu_morph;
private:
    int w0_, h0_;
    std::thread thread_;
    std::exception_ptr error_;
};

} // namespace vmorph
#endif
