// vmorph/sync.hpp -- C++ host mirror of the reference's synchronisation stage: what class Pyramid
// holds after build(video0, video1, f0, f1, start_res) (Algorithm/pyramid.cu:57-165: the level
// table, the layered video / forward-flow arrays, _vector) and class CSyncThread
// (Algorithm/SyncThread.h:7-39; QThread -> std::thread): runflag, percentage, run_time, run(),
// load_identity / upsample_level / optimize_level, update_result().  The CG solves, the level
// transfer and the stage-1 renderer (render_resample_image, render.cu:99-246) live behind the
// C-ABI (vm_sync_*).
#ifndef VMORPH_SYNC_HPP
#define VMORPH_SYNC_HPP

#include <chrono>
#include <map>
#include <thread>
#include <vector>

#include "pyramid.hpp"

namespace vmorph {

struct SyncLevel { int width, height, depth; };

class SyncPyramid {
public:
    explicit SyncPyramid(Context &ctx) : ctx_(ctx) {}
    ~SyncPyramid() { clear(); }
    SyncPyramid(const SyncPyramid &) = delete;
    SyncPyramid &operator=(const SyncPyramid &) = delete;

    void clear()
    {
        if (h_) { vm_sync_destroy(h_); h_ = nullptr; }
        levels.clear();
        _vector.clear();
    }

    // pyramid.cu:143-163; entry 0 = the full-resolution placeholder (append_new(w, h, d), :141)
    static std::vector<SyncLevel> level_table(int w, int h, int d, int start_res)
    {
        int lw[64], lh[64], ld[64], n = 0;
        check(vm_sync_level_table(w, h, d, start_res, lw, lh, ld, 64, &n));
        std::vector<SyncLevel> out;
        for (int i = 0; i < n && i < 64; ++i) out.push_back({lw[i], lh[i], ld[i]});
        return out;
    }

    void create(const std::vector<SyncLevel> &table)
    {
        clear();
        std::vector<int> w, h, d;
        for (const SyncLevel &l : table) { w.push_back(l.width); h.push_back(l.height); d.push_back(l.depth); }
        check(vm_sync_create(ctx_.handle(), (int)table.size(), w.data(), h.data(), d.data(), &h_));
        levels = table;
        _vector.assign(table[0].depth, std::vector<float>((size_t)table[0].width * table[0].height * 4, 0.0f));
    }

    // Pyramid::build(video0, video1, f0, f1, start_res): RGBA8 frames (alpha ignored) and the
    // forward flows of both videos, tight rows
    void build(const std::vector<const unsigned char *> &video0, const std::vector<const unsigned char *> &video1,
               const std::vector<const float *> &f0, const std::vector<const float *> &f1, int w, int h, int start_res)
    {
        const int d = (int)video0.size();
        create(level_table(w, h, d, start_res));
        for (int t = 0; t < d; ++t) {
            check(vm_sync_upload_frame(h_, 0, t, video0[t], w * 4));
            check(vm_sync_upload_frame(h_, 1, t, video1[t], w * 4));
            check(vm_sync_upload_flow(h_, 0, t, f0[t], w * 2));
            check(vm_sync_upload_flow(h_, 1, t, f1[t], w * 2));
        }
    }

    size_t size() const { return levels.size(); }
    const SyncLevel &operator[](int el) const { return levels[el]; }
    vm_sync *handle() const { return h_; }
    Context &context() const { return ctx_; }

    // render_resample_image via RenderWidget::RenderStage1 (UI/RenderWidget.cpp:205-227): h x w RGB8
    std::vector<unsigned char> render_resample(float fa, int frame)
    {
        std::vector<unsigned char> out((size_t)levels[0].width * levels[0].height * 3);
        check(vm_sync_render(h_, fa, frame, out.data(), levels[0].width * 3));
        return out;
    }

    std::vector<SyncLevel> levels;
    std::vector<std::vector<float>> _vector; // Pyramid::_vector: per frame h0 x w0 float4 (x, y, frame shift, 0)

private:
    Context &ctx_;
    vm_sync *h_ = nullptr;
};

// class CSyncThread, SyncThread.h:7-39
class SyncThread {
public:
    // SyncThread.cpp:6-38
    SyncThread(Parameters &parameters, SyncPyramid &pyramids) : runflag(1), _pyramids(pyramids), _parameters(parameters)
    {
        _total_l = (int)pyramids.size() - 1;
        _current_l = _total_l;
        _total_iter = _current_iter = 0;
        _max_iter = (float)(parameters.max_iter * 10);
        int iter_num = parameters.max_iter * 10;
        for (int el = _total_l; el >= 0; el--)
            if (el > 0) {
                _total_iter += (float)iter_num * pyramids[el].width * pyramids[el].height * pyramids[el].depth;
                iter_num = (int)(iter_num / parameters.max_iter_drop_factor);
            }
    }
    // joins the worker; a stored exception is dropped (a destructor must not throw: wait() shows it)
    ~SyncThread()
    {
        if (thread_.joinable()) thread_.join();
    }

    void load_identity(int el) { check(vm_sync_load_identity(_pyramids.handle(), el)); }
    void upsample_level(int el, int /*pel*/) { check(vm_sync_upsample_level(_pyramids.handle(), el)); }

    // SyncThread.cpp:290-480
    void optimize_level(int el)
    {
        KernParameters kp(_parameters);
        check(vm_set_params(_pyramids.context().handle(), &kp));
        std::vector<vm_sync_constraint> cons;
        for (const auto &row : _parameters.cnt)
            for (const Connect &c : row) {
                const Conp &l = _parameters.lp.at(c.li.x).at(c.li.y), &r = _parameters.rp.at(c.ri.x).at(c.ri.y);
                cons.push_back(vm_sync_constraint{l.p.x, l.p.y, l.p.z, r.p.x, r.p.y, r.p.z});
            }
        check(vm_sync_set_constraints(_pyramids.handle(), cons.data(), (int)cons.size()));
        vm_sync_progress pr{};
        check(vm_sync_optimize_level(_pyramids.handle(), el, _max_iter, &runflag, &pr));
        progress[el] = pr;
        _current_iter += (float)pr.voxel_iters;
    }

    // SyncThread.cpp:58-84
    void run()
    {
        auto t0 = std::chrono::steady_clock::now();
        for (_current_l = _total_l; _current_l > 0; _current_l--) {
            const int el = _current_l;
            if (el == _total_l) load_identity(el);
            else upsample_level(el, el + 1);
            optimize_level(el);
            _max_iter /= 2;
            if (!runflag) break;
        }
        run_time = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
        update_result();
    }
    void start() { thread_ = std::thread([this] { try { run(); } catch (...) { error_ = std::current_exception(); } }); }
    void wait()
    {
        if (thread_.joinable()) thread_.join();
        if (error_) { auto e = error_; error_ = nullptr; std::rethrow_exception(e); }
    }

    // SyncThread.cpp:482-521
    void update_result()
    {
        const int el = std::max(_current_l, 1);
        for (int z = 0; z < _pyramids[el].depth; ++z) check(vm_sync_result(_pyramids.handle(), el, z, _pyramids._vector[z].data()));
        percentage = _total_iter > 0 ? _current_iter / _total_iter * 100.0f : 100.0f;
    }

    volatile int runflag; // the reference's `bool runflag`, written by the UI thread
    float percentage = 0.0f;
    float run_time = 0.0f;
    std::map<int, vm_sync_progress> progress;
    int _total_l, _current_l;
    float _total_iter, _current_iter, _max_iter;

private:
    SyncPyramid &_pyramids;
    Parameters &_parameters;
    std::thread thread_;
    std::exception_ptr error_;
};

} // namespace vmorph
#endif
