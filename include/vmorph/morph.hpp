// vmorph/morph.hpp -- C++ host mirror of Algorithm/morph.h (class Morph) and
// Algorithm/MatchingThread.h (class CMatchingThread, QThread -> std::thread),
// reference repository.  Same constructor arguments, same public progress
// members, same cancellation through a caller-owned flag.
#ifndef VMORPH_MORPH_HPP
#define VMORPH_MORPH_HPP

#include <atomic>
#include <chrono>
#include <map>
#include <thread>

#include "pyramid.hpp"

namespace vmorph {

class Morph {
public:
    // Morph(Parameters&, Pyramid&, bool& run_flag), morph.h:13 / morph.cu:122-141.
    // The flag is an int the caller may clear from another thread.
    Morph(Parameters &params, Pyramid &pyramid, volatile int &run_flag, bool fixed_work = false)
        : m_cb(run_flag), m_pyramid(pyramid), m_params(params), fixed_work_(fixed_work)
    {
        _total_l = (int)pyramid.size() - 1;
        _current_l = _total_l;
        _total_iter = _current_iter = 0;
        _max_iter = (float)params.max_iter;
        int iter_num = params.max_iter;
        for (int el = _total_l - 1; el >= 0; el--)
            if (el > 0) {
                _total_iter += (float)iter_num * pyramid[el].width * pyramid[el].height * pyramid[el].depth;
                iter_num = (int)(iter_num / params.max_iter_drop_factor);
            }
    }

    const Parameters &params() { return m_params; }

    // morph.cu:150-168.  Returns true like the reference; device errors throw
    // std::runtime_error (rod::check_cuda_error does in the reference).
    bool calculate_halfway_parametrization()
    {
        vm_pyr *p = m_pyramid.handle();
        KernParameters kp(m_params);
        check(vm_set_params(m_pyramid.context().handle(), &kp));
        std::vector<vm_constraint> cons = m_params.constraints(0);
        const int w0 = m_pyramid[0].width, h0 = m_pyramid[0].height;
        check(vm_coarse_solve(p, _total_l - 1, w0, h0, cons.data(), (int)cons.size()));
        for (_current_l = _total_l - 1; _current_l > 0; _current_l--) {
            if (m_cb) {
                int el = _current_l;
                check(vm_upsample_v(p, el - 1, el));
                check(vm_init_level(p, el - 1, w0, h0, cons.data(), (int)cons.size()));
                vm_progress pr{};
                int rc = vm_optimize_level(p, el - 1, _max_iter, &m_cb, fixed_work_ ? 1 : 0, &pr);
                if (rc != VM_E_CANCELLED) check(rc);
                progress[el] = pr;
                // morph.cu:1389-1391: a finished level is accounted as max_iter sweeps
                _current_iter += (float)m_pyramid[el].width * m_pyramid[el].height * _max_iter;
                check(vm_level_clear(p, el - 1));
                _max_iter /= m_params.max_iter_drop_factor;
            }
        }
        return true;
    }

    int _total_l, _current_l;
    float _total_iter, _current_iter, _max_iter;
    std::map<int, vm_progress> progress; // per level: what actually ran

private:
    volatile int &m_cb;
    Pyramid &m_pyramid;
    Parameters &m_params;
    bool fixed_work_;
};

// class CMatchingThread, MatchingThread.h:7-37
class MatchingThread {
public:
    MatchingThread(Parameters &parameters, Pyramid &pyramids)
        : runflag(1), _pyramids(pyramids), _parameters(parameters), gpu_morph(parameters, pyramids, runflag) {}
    // joins the worker; an exception it stored is dropped here (a destructor must not throw:
    // call wait() first to see it)
    ~MatchingThread()
    {
        if (thread_.joinable()) thread_.join();
    }

    // MatchingThread.cpp:138-150
    void run()
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

    // MatchingThread.cpp:22-84: v of the current level scaled and resized to full
    // resolution into pyramid._vector[0]
    void update_result()
    {
        int el = std::max(gpu_morph._current_l, 1);
        int w0 = _pyramids[0].width, h0 = _pyramids[0].height;
        std::vector<float> out((size_t)w0 * h0 * 2);
        check(vm_upscale_result(_pyramids.handle(), el - 1, w0, h0, out.data(), 0));
        _pyramids._vector.assign(1, out);
        percentage = gpu_morph._total_iter > 0 ? gpu_morph._current_iter / gpu_morph._total_iter * 100.0f : 100.0f;
    }

    float percentage = 0.0f;
    float run_time = 0.0f;
    volatile int runflag; // the reference's `bool runflag`, written by the UI thread

private:
    Pyramid &_pyramids;
    Parameters &_parameters;
public:
    Morph gpu_morph;
private:
    std::thread thread_;
    std::exception_ptr error_;
};

} // namespace vmorph
#endif
