// examples/solve_sync.cpp -- the synchronisation stage through the C++ facade: Pyramid::build for
// stage 1 (frames + forward flows), CSyncThread, then both videos re-timed by the stage-1 renderer.
//   solve_sync W H D frames.u8 flows.f32 cons.i32 NCONS out_field.f32 out_frames.u8 [max_iter] [start_res]
// frames.u8: D x 2 RGBA8 frames (video 0 frame t, video 1 frame t, ...); flows.f32: D x 2 float2
// fields (forward flow of video 0, of video 1); cons.i32: NCONS x (lx ly lz rx ry rz);
// out_field: level 1's (X, Y, Z); out_frames: D x 2 RGB8 (video 0 re-timed, video 1 re-timed).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vmorph/sync.hpp"

template <class T> static std::vector<T> read_all(const char *path, size_t n)
{
    std::vector<T> v(n);
    FILE *f = fopen(path, "rb");
    if (!f || fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 10) { fprintf(stderr, "usage: %s W H D frames.u8 flows.f32 cons.i32 NCONS out_field.f32 out_frames.u8 [max_iter] [start_res]\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), d = atoi(argv[3]), ncons = atoi(argv[7]);
    try {
        vmorph::Context ctx(0);
        vmorph::Parameters params;
        params.w_ui = 100.0f;
        params.w_tps = 0.001f;
        params.max_iter = argc > 10 ? atoi(argv[10]) : 20;
        params.start_res = argc > 11 ? atoi(argv[11]) : 16;
        const size_t npx = (size_t)w * h;
        std::vector<unsigned char> frames = read_all<unsigned char>(argv[4], npx * 4 * 2 * d);
        std::vector<float> flows = read_all<float>(argv[5], npx * 2 * 2 * d);
        std::vector<int> cons = read_all<int>(argv[6], (size_t)ncons * 6);
        for (int k = 0; k < ncons; ++k) {
            const int *q = &cons[6 * k];
            params.lp.push_back({vmorph::Conp{{q[0], q[1], q[2], 1}, 1.0f}});
            params.rp.push_back({vmorph::Conp{{q[3], q[4], q[5], 1}, 1.0f}});
            params.cnt.push_back({vmorph::Connect{{k, 0}, {k, 0}}});
        }
        std::vector<const unsigned char *> v0, v1;
        std::vector<const float *> f0, f1;
        for (int t = 0; t < d; ++t) {
            v0.push_back(frames.data() + npx * 4 * (2 * t));
            v1.push_back(frames.data() + npx * 4 * (2 * t + 1));
            f0.push_back(flows.data() + npx * 2 * (2 * t));
            f1.push_back(flows.data() + npx * 2 * (2 * t + 1));
        }
        vmorph::SyncPyramid pyramid(ctx);
        pyramid.build(v0, v1, f0, f1, w, h, params.start_res);
        vmorph::SyncThread thread(params, pyramid);
        thread.start();
        thread.wait();
        const vmorph::SyncLevel l1 = pyramid[1];
        const size_t n1 = (size_t)l1.width * l1.height * l1.depth;
        std::vector<float> field(3 * n1);
        vmorph::check(vm_sync_get_field(pyramid.handle(), 1, field.data(), field.data() + n1, field.data() + 2 * n1));
        FILE *f = fopen(argv[8], "wb");
        if (!f || fwrite(field.data(), sizeof(float), field.size(), f) != field.size()) { fprintf(stderr, "cannot write %s\n", argv[8]); return 2; }
        fclose(f);
        f = fopen(argv[9], "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", argv[9]); return 2; }
        for (int t = 0; t < d; ++t)
            for (int side = 0; side < 2; ++side) { // MdiEditor::NextStage: RenderStage1(.., 0, i), RenderStage1(.., 1, i)
                std::vector<unsigned char> img = pyramid.render_resample((float)side, t);
                fwrite(img.data(), 1, img.size(), f);
            }
        fclose(f);
        printf("levels %zu, %.1f %% done in %.3f s\n", pyramid.size() - 1, thread.percentage, thread.run_time);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
