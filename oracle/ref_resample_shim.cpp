// ref_resample_shim.cpp -- C-ABI around the REFERENCE's own resampling library
// (include/resample/*, Nehab-Hoppe generalized sampling), compiled by
// `make -C oracle ref` together with the reference's sources WHERE THEY LIE under
// /root/reference into oracle/_ref/libresample_ref.so.  Test infrastructure only:
// it exists to produce golden luma pyramids (tests/golden/make_pyramid_golden.py)
// and to check the restatement oracle/vm_oracle_pyramid.c.  Nothing of the
// reference is copied: this file only CALLS image::load / scale / image::store_gray
// in the order Pyramid::build does (Algorithm/pyramid.cu:203-211, 268-279, 355-364).
#include <cstdio>
#include <cstring>
#include <vector>

#include <resample/scale.h>

// image.h declares load/store_gray as templates while image.cpp defines them as plain
// functions on rgba<float> (MSVC links the two; g++ needs to see the plain overloads):
// declare what image.cpp actually defines.
namespace image {
int load(image::rgba<float> *rgba, float *data, int w, int h);
int store_gray(float *data, const image::rgba<float> &rgba);
}

extern "C" int ref_luma_pyramid(const float *rgb_0_255, int w, int h, int nlevels, float *out)
{
    // use cardinal bspline3 prefilter for downsampling (pyramid.cu:203-211)
    kernel::base *pre = new kernel::generalized(new kernel::discrete::delta,
                                                new kernel::discrete::sampled(new kernel::generating::bspline3),
                                                new kernel::generating::bspline3);
    kernel::discrete::base *delta = new kernel::discrete::delta;
    extension::base *ext = new extension::mirror;
    image::rgba<float> rgba;
    std::vector<float> data(rgb_0_255, rgb_0_255 + (size_t)w * h * 3);
    // level 1 (el == 0): load, scale to the working size (here: unchanged), store_gray
    image::load(&rgba, data.data(), w, h);
    scale(h, w, pre, delta, delta, ext, &rgba, &rgba);
    image::store_gray(out, rgba);
    out += (size_t)w * h;
    // coarser levels: scaled from the previous level's linear-light image, in place
    for (int el = 1; el < nlevels; ++el) {
        w = (w + 1) / 2; // ceil(w/2.0f), pyramid.cu:466-467
        h = (h + 1) / 2;
        scale(h, w, pre, delta, delta, ext, &rgba, &rgba);
        image::store_gray(out, rgba);
        out += (size_t)w * h;
    }
    delete pre;
    delete delta;
    delete ext;
    return 0;
}
