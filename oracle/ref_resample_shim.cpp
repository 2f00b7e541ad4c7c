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

// The flow half of Pyramid::build (Algorithm/pyramid.cu:284-321, 375-404) for ONE flow field:
// image::load(rgba, data, w, h, rowstride, -50, 50) -> scale() -> image::store(data, rgba,
// rowstride, -50, 50) of the reference's own library, then the x (wout/w, hout/h) rescale the
// caller applies there when the size shrinks.  flow: h*w*2 floats, out: hout*wout*2.
namespace image {
int load(image::rgba<float> *rgba, float *data, int w, int h, int rowstride, float min, float max);
int store(float *data, const image::rgba<float> &rgba, int rowstride, float min, float max);
}

extern "C" int ref_flow_scale(const float *flow, int w, int h, int wout, int hout, float *out)
{
    kernel::base *pre = new kernel::generalized(new kernel::discrete::delta,
                                                new kernel::discrete::sampled(new kernel::generating::bspline3),
                                                new kernel::generating::bspline3);
    kernel::discrete::base *delta = new kernel::discrete::delta;
    extension::base *ext = new extension::mirror;
    image::rgba<float> temp;
    std::vector<float> data(flow, flow + (size_t)w * h * 2);
    image::load(&temp, data.data(), w, h, w, -50, 50);
    scale(hout, wout, pre, delta, delta, ext, &temp, &temp);
    image::store(out, temp, wout, -50, 50);
    const float ratiox = (float)wout / (float)w, ratioy = (float)hout / (float)h;
    if (ratiox < 1 || ratioy < 1)
        for (size_t p = 0; p < (size_t)wout * hout; ++p) {
            out[2 * p] *= ratiox;
            out[2 * p + 1] *= ratioy;
        }
    delete pre;
    delete delta;
    delete ext;
    return 0;
}
