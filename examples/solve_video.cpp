// examples/solve_video.cpp -- a video pair through the C++ facade: Pyramid::build for a video
// (frames + optical flows), the temporally coupled Morph, the halfway field of every frame.
//   solve_video W H D frames.u8 flows.f32 out_v.f32 [max_iter] [start_res] [exact|fast]
// frames.u8: D x 2 RGB8 frames (video 0 frame t, video 1 frame t, ...), flows.f32: D x 4 fields
// (f0, f1, b0, b1 of frame t), out: the full-resolution halfway field of every frame
// (Pyramid::_vector after CMatchingThread::update_result).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vmorph/video.hpp"

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
    if (argc < 7) { fprintf(stderr, "usage: %s W H D frames.u8 flows.f32 out [max_iter] [start_res] [exact|fast]\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), d = atoi(argv[3]);
    try {
        vmorph::Context ctx(0, argc > 9 && !strcmp(argv[9], "fast") ? VM_MATH_FAST : VM_MATH_EXACT);
        vmorph::Parameters params;
        params.max_iter = argc > 7 ? atoi(argv[7]) : 50;
        params.start_res = argc > 8 ? atoi(argv[8]) : 16;
        params.max_iter_drop_factor = 1.0f;
        const size_t npx = (size_t)w * h;
        std::vector<unsigned char> frames = read_all<unsigned char>(argv[4], npx * 3 * 2 * d);
        std::vector<float> flows = read_all<float>(argv[5], npx * 2 * 4 * d);
        std::vector<const unsigned char *> v0, v1;
        std::vector<const float *> f0, f1, b0, b1;
        for (int t = 0; t < d; ++t) {
            v0.push_back(frames.data() + npx * 3 * (2 * t));
            v1.push_back(frames.data() + npx * 3 * (2 * t + 1));
            f0.push_back(flows.data() + npx * 2 * (4 * t));
            f1.push_back(flows.data() + npx * 2 * (4 * t + 1));
            b0.push_back(flows.data() + npx * 2 * (4 * t + 2));
            b1.push_back(flows.data() + npx * 2 * (4 * t + 3));
        }
        vmorph::VideoPyramid pyramid(ctx);
        pyramid.build(v0, v1, f0, f1, b0, b1, w, h, params.start_res);
        // class CMatchingThread over the video pair: the solve on a worker thread, then update_result()
        vmorph::VideoMatchingThread thread(params, pyramid, w, h);
        thread.start();
        thread.wait();
        vmorph::VideoMorph &morph = thread.gpu_morph;
        FILE *f = fopen(argv[6], "wb");
        for (int t = 0; t < d; ++t)
            fwrite(pyramid._vector[t].data(), 4, pyramid._vector[t].size(), f);
        fclose(f);
        printf("levels %zu:", pyramid.levels.size());
        for (auto &l : pyramid.levels) printf(" %dx%dx%d", l.width, l.height, l.depth);
        printf("\n");
        for (int t = 0; t < pyramid.levels[0].depth; ++t)
            printf("  finest level page %d: %d iterations\n", t, morph.progress[t].iters);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
