// examples/pipeline_shard.cpp -- BASELINE config[4] from ONE C++ process driving G devices: frame pairs with user point
// constraints and BCOND_BORDER, block-sharded over the ranks ("for config 5: 30 pairs/4 GPUs"); every rank
// solves ITS pairs as one batch (vm_solve_batch_cons), then runs the compositor for them -- canvases up, halfway field
// upscaled on the device, Poisson extension of both sides of up to four frames per batch, one rendered in-between
// frame per pair.  The parameter block AND the constraints travel in exactly one broadcast (vm_bcast_bytes: RCCL over
// xGMI; the reference is single-GPU, UI/MdiEditor.cpp:54-75).  Sibling of examples/solve_shard.cpp (config[2]: solve only).
//
//   pipeline_shard G W H N EX frames.f32 rgb.u8 cons.f32 out_v.f32 out_rgb.u8 [max_iter] [start_res] [exact|fast] [--one-device]
//
// frames.f32: N pairs, img0 then img1, (H, W) float32 luma.  rgb.u8: N pairs, RGB8 (H, W, 3).  cons.f32: M rows of
// (lx, ly, rx, ry, weight), shared by every pair (config[4]'s synthetic frames).  out_v.f32: N fields (H, W, 2);
// out_rgb.u8: N frames (H, W, 3) rendered at t = 0.5, color_from 1.  Rank r = one host thread + one vm_ctx on device r
// (--one-device: all on device 0; the broadcast then goes device-to-device, vm_bcast_bytes' test mode).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <thread>
#include <vector>

#include "vmorph/morph.hpp"
#include "vmorph/render.hpp"

static std::vector<int> shard_pairs(int n_pairs, int world, int rank)
{
    std::vector<int> mine;
    for (int k = 0; k < n_pairs; ++k)
        if ((int)(((long long)k * world) / n_pairs) == rank) mine.push_back(k);
    return mine;
}

template <class T> static bool read_all(const char *path, std::vector<T> &v)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(T));
    const bool ok = fread(v.data(), sizeof(T), v.size(), f) == v.size();
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 11) {
        fprintf(stderr, "usage: %s G W H N EX frames.f32 rgb.u8 cons.f32 out_v.f32 out_rgb.u8 [max_iter] [start_res] [exact|fast] [--one-device]\n", argv[0]);
        return 2;
    }
    const int G = atoi(argv[1]), w = atoi(argv[2]), h = atoi(argv[3]), N = atoi(argv[4]), ex = atoi(argv[5]);
    bool one_device = false;
    for (int a = 11; a < argc; ++a) one_device = one_device || !strcmp(argv[a], "--one-device");
    const size_t npx = (size_t)w * h;
    std::vector<float> frames, cons_in;
    std::vector<unsigned char> rgb;
    if (!read_all(argv[6], frames) || frames.size() != (size_t)N * 2 * npx || !read_all(argv[7], rgb) || rgb.size() != (size_t)N * 2 * npx * 3 ||
        !read_all(argv[8], cons_in) || cons_in.size() % 5) {
        fprintf(stderr, "cannot read the inputs (sizes?)\n");
        return 2;
    }
    std::vector<float> out_v((size_t)N * 2 * npx);
    std::vector<unsigned char> out_rgb((size_t)N * npx * 3);
    try {
        // ---- rank 0's payload: the parameter block followed by the constraints (the layout of dist.py:pack_block)
        vmorph::Parameters params;
        params.max_iter = argc > 11 && argv[11][0] != '-' ? atoi(argv[11]) : 100;
        params.start_res = argc > 12 && argv[12][0] != '-' ? atoi(argv[12]) : 32;
        params.max_iter_drop_factor = 1.0f;
        params.bcond = vmorph::BCOND_BORDER;
        vm_param_block blk{};
        blk.kp = vmorph::KernParameters(params);
        blk.max_iter = (float)params.max_iter;
        blk.max_iter_drop_factor = params.max_iter_drop_factor;
        blk.start_res = params.start_res;
        blk.math_mode = argc > 13 && !strcmp(argv[13], "fast") ? VM_MATH_FAST : VM_MATH_EXACT;
        blk.n_constraints = (int)(cons_in.size() / 5);
        static_assert(sizeof(vm_constraint) == 20, "a constraint is five floats");
        std::vector<char> payload(sizeof(blk) + cons_in.size() * 4);
        memcpy(payload.data(), &blk, sizeof(blk));
        memcpy(payload.data() + sizeof(blk), cons_in.data(), cons_in.size() * 4);

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
        std::vector<std::vector<char>> got(G, std::vector<char>(payload.size()));
        std::vector<void *> dst(G);
        for (int r = 0; r < G; ++r) dst[r] = got[r].data();
        vmorph::check(vm_bcast_bytes(handles.data(), one_device ? nullptr : comms.data(), G, 0, payload.data(), payload.size(), dst.data()));
        for (void *c : comms) vm_rccl_comm_destroy(c);

        std::vector<std::exception_ptr> errs(G);
        std::vector<std::thread> workers;
        for (int r = 0; r < G; ++r)
            workers.emplace_back([&, r] {
                try {
                    // ---- what THIS rank received: block, then constraints
                    vm_param_block b;
                    memcpy(&b, got[r].data(), sizeof(b));
                    std::vector<vm_constraint> cons(b.n_constraints);
                    memcpy(cons.data(), got[r].data() + sizeof(b), cons.size() * sizeof(vm_constraint));
                    vmorph::check(vm_set_params(handles[r], &b.kp));
                    vmorph::check(vm_set_math_mode(handles[r], b.math_mode));
                    const std::vector<int> mine = shard_pairs(N, G, r);
                    if (mine.empty()) return;
                    // ---- solve: the rank's pairs as one batch, every pair with the constraints
                    std::vector<std::unique_ptr<vmorph::Pyramid>> pyrs;
                    std::vector<vm_pyr *> ph;
                    for (int k : mine) {
                        pyrs.emplace_back(new vmorph::Pyramid(*ctxs[r]));
                        pyrs.back()->build(&frames[(size_t)k * 2 * npx], &frames[(size_t)k * 2 * npx + npx], w, h, b.start_res);
                        ph.push_back(pyrs.back()->handle());
                    }
                    std::vector<const vm_constraint *> cp(mine.size(), cons.data());
                    std::vector<int> cn(mine.size(), (int)cons.size());
                    vmorph::check(vm_solve_batch_cons(ph.data(), (int)ph.size(), b.max_iter, b.max_iter_drop_factor, cp.data(), cn.data(), nullptr, 0, nullptr));
                    // ---- compositor: up to four frames (eight systems) per Poisson batch
                    const int per_batch = 4;
                    std::vector<std::unique_ptr<vmorph::Frame>> frs;
                    for (int j = 0; j < per_batch; ++j) frs.emplace_back(new vmorph::Frame(*ctxs[r], w, h, ex));
                    for (size_t g0 = 0; g0 < mine.size(); g0 += per_batch) {
                        std::vector<vmorph::Frame *> batch;
                        for (size_t j = g0; j < mine.size() && j < g0 + per_batch; ++j) {
                            const int k = mine[j];
                            const std::vector<unsigned char> e0 = vmorph::make_extended(&rgb[(size_t)k * 2 * npx * 3], w, h, ex),
                                                             e1 = vmorph::make_extended(&rgb[(size_t)k * 2 * npx * 3 + npx * 3], w, h, ex);
                            vmorph::Frame *f = frs[j - g0].get();
                            f->upload(e0.data(), e1.data(), nullptr, nullptr);
                            f->set_v_from_level(*pyrs[j], 1);
                            batch.push_back(f);
                        }
                        vmorph::poisson_extend_frames(batch, 1e-5f);
                        for (size_t j = 0; j < batch.size(); ++j) {
                            const int k = mine[g0 + j];
                            const std::vector<unsigned char> img = batch[j]->render_halfway_image(0.5f, 0.5f, 1);
                            memcpy(&out_rgb[(size_t)k * npx * 3], img.data(), img.size());
                            vmorph::check(vm_upscale_result(ph[g0 + j], 0, w, h, &out_v[(size_t)k * 2 * npx], 0));
                        }
                    }
                } catch (...) {
                    errs[r] = std::current_exception();
                }
            });
        for (auto &t : workers) t.join();
        for (auto &e : errs)
            if (e) std::rethrow_exception(e);
        FILE *f = fopen(argv[9], "wb");
        if (!f || fwrite(out_v.data(), 4, out_v.size(), f) != out_v.size()) { fprintf(stderr, "cannot write %s\n", argv[9]); return 2; }
        fclose(f);
        f = fopen(argv[10], "wb");
        if (!f || fwrite(out_rgb.data(), 1, out_rgb.size(), f) != out_rgb.size()) { fprintf(stderr, "cannot write %s\n", argv[10]); return 2; }
        fclose(f);
        for (int r = 0; r < G; ++r) {
            vm_param_block b;
            memcpy(&b, got[r].data(), sizeof(b));
            printf("rank %d on device %d: %zu pairs, as received: max_iter %g bcond %d constraints %d\n", r, devices[r], shard_pairs(N, G, r).size(),
                   b.max_iter, b.kp.bcond, b.n_constraints);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
