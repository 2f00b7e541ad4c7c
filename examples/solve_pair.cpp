// examples/solve_pair.cpp -- the reference's driver sequence (MdiEditor::match_start
// -> CMatchingThread::run -> render) on the C++ facade: reads two raw float32 luma
// images, solves, writes the full-resolution halfway field.
//   solve_pair W H img0.f32 img1.f32 out_v.f32 [max_iter] [start_res] [exact|fast]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vmorph/morph.hpp"

static std::vector<float> read_f32(const char *path, size_t n)
{
    std::vector<float> v(n);
    FILE *f = fopen(path, "rb");
    if (!f || fread(v.data(), 4, n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s W H img0 img1 out [max_iter] [start_res] [exact|fast]\n", argv[0]); return 2; }
    int w = atoi(argv[1]), h = atoi(argv[2]);
    try {
        vmorph::Context ctx(0, argc > 8 && !strcmp(argv[8], "fast") ? VM_MATH_FAST : VM_MATH_EXACT);
        vmorph::Parameters params;
        params.max_iter = argc > 6 ? atoi(argv[6]) : 100;
        params.start_res = argc > 7 ? atoi(argv[7]) : 32;
        params.max_iter_drop_factor = 1.0f;
        std::vector<float> i0 = read_f32(argv[3], (size_t)w * h), i1 = read_f32(argv[4], (size_t)w * h);
        vmorph::Pyramid pyramid(ctx);
        pyramid.build(i0.data(), i1.data(), w, h, params.start_res);
        vmorph::MatchingThread thread(params, pyramid);
        thread.start();
        thread.wait();
        FILE *f = fopen(argv[5], "wb");
        fwrite(pyramid._vector[0].data(), 4, pyramid._vector[0].size(), f);
        fclose(f);
        printf("levels %zu  run_time %.3f s  progress %.1f %%\n", pyramid.size() - 1, thread.run_time, thread.percentage);
        for (auto &kv : thread.gpu_morph.progress)
            printf("  level %d (%dx%d): %d iterations, %.2f ms\n", kv.first, pyramid[kv.first].width,
                   pyramid[kv.first].height, kv.second.iters, kv.second.elapsed_ms);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
