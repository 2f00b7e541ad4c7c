// examples/solve_shard.cpp -- config[2]'s job from ONE C++ process driving G devices ("one
// process, 8 devices"; 8(e): independent frame pairs, static block partition, exactly one RCCL broadcast of the
// shared parameter block; the reference itself is single-GPU, UI/MdiEditor.cpp:54-75).
//
//   solve_shard G W H N frames.f32 out_v.f32 [max_iter] [start_res] [exact|fast] [--one-device]
//   solve_shard --plan G N                          (prints the partition only; touches no GPU)
//
// frames.f32: N pairs, each img0 then img1, (H, W) float32 luma.  out_v.f32: N full-resolution halfway fields
// (H, W, 2), in pair order.  Rank r = one host thread + one vm_ctx on device r (--one-device: every context on
// device 0, the switch a one-GPU box tests this with; RCCL wants one rank per device, so the block then goes
// device-to-device instead: vm_bcast_params' test mode).  Pair k belongs to rank floor(k G / N) -- the rule of
// videomorphing_amd/dist.py:shard_pairs, which the N-process form (bench.py under torch.distributed) uses.
// Nothing but the parameter block crosses a link.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <thread>
#include <vector>

#include "vmorph/morph.hpp"

static std::vector<int> shard_pairs(int n_pairs, int world, int rank)
{
    std::vector<int> mine;
    for (int k = 0; k < n_pairs; ++k)
        if ((int)(((long long)k * world) / n_pairs) == rank) mine.push_back(k);
    return mine;
}

int main(int argc, char **argv)
{
    if (argc >= 4 && !strcmp(argv[1], "--plan")) {
        const int G = atoi(argv[2]), N = atoi(argv[3]);
        for (int r = 0; r < G; ++r) {
            printf("rank %d:", r);
            for (int k : shard_pairs(N, G, r)) printf(" %d", k);
            printf("\n");
        }
        return 0;
    }
    if (argc < 7) {
        fprintf(stderr, "usage: %s G W H N frames.f32 out_v.f32 [max_iter] [start_res] [exact|fast] [--one-device]\n", argv[0]);
        return 2;
    }
    const int G = atoi(argv[1]), w = atoi(argv[2]), h = atoi(argv[3]), N = atoi(argv[4]);
    bool one_device = false;
    for (int a = 7; a < argc; ++a) one_device = one_device || !strcmp(argv[a], "--one-device");
    const size_t npx = (size_t)w * h;
    std::vector<float> frames((size_t)N * 2 * npx), out((size_t)N * 2 * npx);
    {
        FILE *f = fopen(argv[5], "rb");
        if (!f || fread(frames.data(), 4, frames.size(), f) != frames.size()) { fprintf(stderr, "cannot read %s\n", argv[5]); return 2; }
        fclose(f);
    }
    try {
        // ---- rank 0 decides the shared block; everybody starts from defaults and adopts what the broadcast brings
        vmorph::Parameters params;
        params.max_iter = argc > 7 && argv[7][0] != '-' ? atoi(argv[7]) : 100;
        params.start_res = argc > 8 && argv[8][0] != '-' ? atoi(argv[8]) : 32;
        params.max_iter_drop_factor = 1.0f;
        vm_param_block blk{};
        blk.kp = vmorph::KernParameters(params);
        blk.max_iter = (float)params.max_iter;
        blk.max_iter_drop_factor = params.max_iter_drop_factor;
        blk.start_res = params.start_res;
        blk.math_mode = argc > 9 && !strcmp(argv[9], "fast") ? VM_MATH_FAST : VM_MATH_EXACT;
        std::vector<std::unique_ptr<vmorph::Context>> ctxs;
        std::vector<vm_ctx *> handles;
        std::vector<int> devices;
        for (int r = 0; r < G; ++r) {
            devices.push_back(one_device ? 0 : r);
            ctxs.emplace_back(new vmorph::Context(devices.back(), VM_MATH_EXACT));
            handles.push_back(ctxs.back()->handle());
        }
        std::vector<void *> comms(G, nullptr);
        if (!one_device) vmorph::check(vm_rccl_comm_init_all(G, devices.data(), comms.data()));
        std::vector<vm_param_block> got(G);
        vmorph::check(vm_bcast_params(handles.data(), one_device ? nullptr : comms.data(), G, 0, &blk, got.data()));
        for (void *c : comms) vm_rccl_comm_destroy(c);
        // ---- every rank solves its shard as one batch (vm_solve_batch: all of its pairs behind the same launches)
        std::vector<std::exception_ptr> errs(G);
        std::vector<std::thread> workers;
        for (int r = 0; r < G; ++r)
            workers.emplace_back([&, r] {
                try {
                    const vm_param_block &b = got[r];                 // what THIS rank received
                    const std::vector<int> mine = shard_pairs(N, G, r);
                    if (mine.empty()) return;
                    std::vector<std::unique_ptr<vmorph::Pyramid>> pyrs;
                    std::vector<vm_pyr *> ph;
                    for (int k : mine) {
                        pyrs.emplace_back(new vmorph::Pyramid(*ctxs[r]));
                        pyrs.back()->build(&frames[(size_t)k * 2 * npx], &frames[(size_t)k * 2 * npx + npx], w, h, b.start_res);
                        ph.push_back(pyrs.back()->handle());
                    }
                    vmorph::check(vm_solve_batch(ph.data(), (int)ph.size(), b.max_iter, b.max_iter_drop_factor, nullptr, 0, nullptr));
                    for (size_t i = 0; i < mine.size(); ++i)       // CMatchingThread::update_result: full-resolution field to the host
                        vmorph::check(vm_upscale_result(ph[i], 0, w, h, &out[(size_t)mine[i] * 2 * npx], 0));
                } catch (...) {
                    errs[r] = std::current_exception();
                }
            });
        for (auto &t : workers) t.join();
        for (auto &e : errs)
            if (e) std::rethrow_exception(e);
        FILE *f = fopen(argv[6], "wb");
        if (!f || fwrite(out.data(), 4, out.size(), f) != out.size()) { fprintf(stderr, "cannot write %s\n", argv[6]); return 2; }
        fclose(f);
        for (int r = 0; r < G; ++r)
            printf("rank %d on device %d: %zu pairs, block as received: max_iter %g start_res %d math %d w_ssim %g\n", r, devices[r],
                   shard_pairs(N, G, r).size(), got[r].max_iter, got[r].start_res, got[r].math_mode, got[r].kp.w_ssim);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
