/*
 * vm_oracle_pyramid.c -- CPU ORACLE (test infrastructure, NOT the product path):
 * the luma pyramid of Pyramid::build (Algorithm/pyramid.cu:203-211, 268-279,
 * 355-364), i.e. the reference's Nehab-Hoppe generalized-sampling library
 * (include/resample) as it is driven there, restated in plain C.
 *
 * PINNED BY REFERENCE-RUN OUTPUTS: the reference's resample library builds here
 * from its own sources (make -C oracle ref -> oracle/_ref/libresample_ref.so);
 * tests/golden/pyramid_ref_*.npz holds luma pyramids produced by it
 * (tests/golden/make_pyramid_golden.py) and tests/test_pyramid_oracle.py checks
 * this file against them.
 *
 * Algorithm (citations relative to /root/reference/include/resample):
 *  load        image.cpp:10-31    x/255, sRGB -> linear light
 *  scale       scale.cpp:225-272  per axis: downsample (scale.cpp:125-223) when the
 *                                 output is smaller, else "upsample" (scale.cpp:9-123,
 *                                 also the same-size case); the axis with the larger
 *                                 reduction first
 *   downsample one axis: out[o] = sum_i w_i in[mirror(i)] / sum_i w_i,
 *                                 w_i = B3(0.5 + o - (i+0.5) out/in), i over the
 *                                 support, then solve [1/6 4/6 1/6] x = out along the
 *                                 axis with mirror boundary (dlti.cpp:66-129, 232-270:
 *                                 banded LU without pivoting, float)
 *   upsample one axis:   to gamma space, the same tridiagonal solve, cubic B-spline
 *                                 reconstruction at (o+.5) in/out - .5, back to linear
 *  store_gray  image.cpp:87-103   clamp, linear -> sRGB, x255, .299 R + .587 G + .114 B
 */
#include "vm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static float srgbcurve(float f) /* color.h:7-17 */
{
    const float a = 0.055f;
    if (f <= 0.0031308f) return 12.92f * f;
    return (1.f + a) * powf(f, 1.f / 2.4f) - a;
}

static float srgbuncurve(float f) /* color.h:26-35 */
{
    const float a = 0.055f;
    if (f <= 0.04045f) return f / 12.92f;
    return powf((f + a) / (1.f + a), 2.4f);
}

static float bspline3(float r) /* generating.h:220-232 */
{
    r = fabsf(r);
    if (r < 1.f) return (4.f + r * r * (-6.f + 3.f * r)) / 6.f;
    else if (r < 2.f) return (8.f + r * (-12.f + (6.f - r) * r)) / 6.f;
    return 0.f;
}

static int ext_repeat(int i, int n) { return i >= 0 ? i % n : (n - 1) - ((-i - 1) % n); }
static int ext_mirror(int i, int n) /* extension.h:54-66 */
{
    i = ext_repeat(i, 2 * n);
    return i >= n ? 2 * n - i - 1 : i;
}
static int ext_clamp(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

/* the factored tridiagonal operator [B3(1) B3(0) B3(-1)] with mirror boundary:
 * dlti.cpp:232-257 (assembly), :66-93 (factor).  band[(i-j+1)*n + j] = A(i,j) */
static float *tri_factor(int n)
{
    float *A = (float *)calloc((size_t)3 * n, sizeof(float));
    const float kern[3] = {bspline3(1.f), bspline3(0.f), bspline3(-1.f)};
#define A_(i, j) A[((i) - (j) + 1) * n + (j)]
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            A_(i, ext_mirror(i + k - 1, n)) += kern[k];
    for (int p = 0; p < n; ++p) {
        float inv_p = (A_(p, p) = 1.f / A_(p, p));
        for (int i = p + 1; i <= p + 1 && i < n; ++i) {
            float m = (A_(i, p) *= inv_p);
            for (int j = p + 1; j <= p + 1 && j < n; ++j)
                A_(i, j) -= m * A_(p, j);
        }
    }
    return A;
}

/* solve_rows / solve_columns, dlti.cpp:97-168, on one line with a stride */
static void tri_solve(const float *A, int n, float *x, int stride)
{
    for (int j = 0; j < n; ++j)
        if (j - 1 >= 0)
            x[j * stride] -= A_(j, j - 1) * x[(j - 1) * stride];
    for (int j = n - 1; j >= 0; --j) {
        if (j + 1 < n)
            x[j * stride] -= A_(j, j + 1) * x[(j + 1) * stride];
        x[j * stride] *= A_(j, j);
    }
#undef A_
}

/* one axis of one channel.  in: nin samples with stride sin, out: nout with stride sout */
static void down_axis(const float *in, int nin, int sin, float *out, int nout, int sout)
{
    const float inv_nin = 1.f / (float)nin;
    const float inv_sw = (float)nout * inv_nin;
    const float sw = 1.f / inv_sw;
    const float s = 4.f; /* support of the generalized bspline3 kernel */
    for (int o = 0; o < nout; ++o) {
        int lo = (int)ceilf(.5f * sw * (2.f * o + 1.f - s) - .5f);
        int hi = (int)floorf(.5f * sw * (2.f * o + 1.f + s) - .5f);
        if (lo > hi)
            lo = hi = (int)(.5f * sw * (2.f * o + 1.f));
        float sum = 0.f, sum_w = 0.f;
        for (int i = lo; i <= hi; ++i) {
            float kj = (float)(0.5 + o - (i + 0.5f) * inv_sw); /* `0.5+iout-...` is double there */
            float w = bspline3(kj);
            sum += in[ext_clamp(ext_mirror(i, nin), nin) * sin] * w;
            sum_w += w;
        }
        out[o * sout] = sum / sum_w;
    }
}

static void up_axis(const float *in_prefiltered, int nin, int sin, float *out, int nout, int sout)
{
    const float inv_nout = 1.f / (float)nout;
    const float inv_sw = (float)nin * inv_nout;
    for (int o = 0; o < nout; ++o) {
        float f = ((float)o + .5f) * inv_sw - .5f;
        int c = (int)floorf(f);
        float d = f - c;
        float sum = 0.f;
        for (int j = -1; j <= 2; ++j)
            sum += in_prefiltered[ext_clamp(ext_mirror(c + j, nin), nin) * sin] * bspline3(d - j);
        out[o * sout] = sum;
    }
}

/* one axis of a 3-channel planar image; axis 0 = rows (x), 1 = columns (y) */
static float *scale_axis(float *img, int *w, int *h, int nout, int axis)
{
    const int win = *w, hin = *h;
    const int nin = axis == 0 ? win : hin;
    const int wout = axis == 0 ? nout : win, hout = axis == 0 ? hin : nout;
    float *out = (float *)malloc(sizeof(float) * 3 * (size_t)wout * hout);
    const int lines = axis == 0 ? hin : win;
    for (int c = 0; c < 3; ++c) {
        float *src = img + (size_t)c * win * hin, *dst = out + (size_t)c * wout * hout;
        if (nout < nin) {
            for (int l = 0; l < lines; ++l) {
                if (axis == 0) down_axis(src + (size_t)l * win, nin, 1, dst + (size_t)l * wout, nout, 1);
                else down_axis(src + l, nin, win, dst + l, nout, wout);
            }
            float *A = tri_factor(nout);
            for (int l = 0; l < lines; ++l)
                tri_solve(A, nout, axis == 0 ? dst + (size_t)l * wout : dst + l, axis == 0 ? 1 : wout);
            free(A);
        } else {
            /* move to gamma space, digital prefilter, reconstruct, back to linear */
            for (size_t i = 0; i < (size_t)win * hin; ++i) src[i] = srgbcurve(src[i]);
            float *A = tri_factor(nin);
            for (int l = 0; l < lines; ++l)
                tri_solve(A, nin, axis == 0 ? src + (size_t)l * win : src + l, axis == 0 ? 1 : win);
            free(A);
            for (int l = 0; l < lines; ++l) {
                if (axis == 0) up_axis(src + (size_t)l * win, nin, 1, dst + (size_t)l * wout, nout, 1);
                else up_axis(src + l, nin, win, dst + l, nout, wout);
            }
            for (size_t i = 0; i < (size_t)wout * hout; ++i) dst[i] = srgbuncurve(dst[i]);
        }
    }
    free(img);
    *w = wout;
    *h = hout;
    return out;
}

/* scale(), scale.cpp:225-272 (the discrete post-filter is delta * delta^-1: identity) */
static float *scale_image(float *img, int *w, int *h, int wout, int hout)
{
    if (hout * *w < wout * *h) {
        img = scale_axis(img, w, h, hout, 1);
        img = scale_axis(img, w, h, wout, 0);
    } else {
        img = scale_axis(img, w, h, wout, 0);
        img = scale_axis(img, w, h, hout, 1);
    }
    return img;
}

static void store_gray(float *out, const float *img, int w, int h) /* image.cpp:87-103 */
{
    const size_t n = (size_t)w * h;
    for (size_t p = 0; p < n; ++p) {
        float c[3];
        for (int k = 0; k < 3; ++k) {
            float v = img[k * n + p];
            v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
            c[k] = srgbcurve(v) * 255;
        }
        out[p] = (float)(c[0] * 0.299 + c[1] * 0.587 + c[2] * 0.114);
    }
}

/* rgb: h*w*3 bytes; out: the lumas of levels 1..nlevels (finest first), concatenated */
void vmo_luma_pyramid(const uint8_t *rgb, int w, int h, int nlevels, float *out)
{
    const size_t n = (size_t)w * h;
    float *img = (float *)malloc(sizeof(float) * 3 * n);
    const float tof = 1.f / 255.f;
    for (size_t p = 0; p < n; ++p)
        for (int k = 0; k < 3; ++k)
            img[k * n + p] = srgbuncurve((float)rgb[3 * p + k] * tof); /* image.cpp:10-31 */
    img = scale_image(img, &w, &h, w, h); /* el == 0: same size (no decimation below 14 Mpx) */
    store_gray(out, img, w, h);
    out += (size_t)w * h;
    for (int el = 1; el < nlevels; ++el) {
        img = scale_image(img, &w, &h, (w + 1) / 2, (h + 1) / 2);
        store_gray(out, img, w, h);
        out += (size_t)w * h;
    }
    free(img);
}

/* ------------------------------------------------------------------------- */
/* Flow half of Pyramid::build, Algorithm/pyramid.cu:284-321 (level 1) and
 * :375-404 (coarser levels): image::load(min = -50, max = 50) (image.cpp:33-54:
 * (f - min) / (max - min) through srgbuncurve into the r and g planes, b = 1),
 * scale(), image::store(min, max) (image.cpp:72-85: clamp, srgbcurve, x (max - min)
 * + min), then x (wout/w, hout/h) when either ratio is below one (:306-321). */
void vmo_flow_scale(const float *flow, int w, int h, int wout, int hout, float *out)
{
    const int w_in = w, h_in = h;
    const size_t n = (size_t)w * h;
    float *img = (float *)malloc(sizeof(float) * 3 * n);
    const float mn = -50.f, mx = 50.f;
    const float tof = 1.f / (mx - mn);
    for (size_t p = 0; p < n; ++p) {
        img[p] = srgbuncurve((flow[2 * p] - mn) * tof);
        img[n + p] = srgbuncurve((flow[2 * p + 1] - mn) * tof);
        img[2 * n + p] = 1.0f;
    }
    img = scale_image(img, &w, &h, wout, hout);
    const size_t m = (size_t)w * h;
    const float ratiox = (float)wout / (float)w_in, ratioy = (float)hout / (float)h_in;
    for (size_t p = 0; p < m; ++p)
        for (int k = 0; k < 2; ++k) {
            float v = img[k * m + p];
            v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
            float f = srgbcurve(v) * (mx - mn) + mn;
            if (ratiox < 1 || ratioy < 1)
                f *= k == 0 ? ratiox : ratioy;
            out[2 * p + k] = f;
        }
    free(img);
}

/* Pyramid::BiLinear<cv::Vec2f>, pyramid.cu:486-522 */
static void bilinear_f2(const float *img, int cols, int rows, float px, float py, float *o)
{
    int x[2], y[2];
    x[0] = (int)floorf(px);
    y[0] = (int)floorf(py);
    x[1] = (int)ceilf(px);
    y[1] = (int)ceilf(py);
    const float u = px - x[0], v = py - y[0];
    float val[2][2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            int tx = x[i], ty = y[j];
            tx = tx < 0 ? 0 : tx;
            tx = tx > cols - 1 ? cols - 1 : tx;
            ty = ty < 0 ? 0 : ty;
            ty = ty > rows - 1 ? rows - 1 : ty;
            val[i][j][0] = img[2 * (ty * cols + tx)];
            val[i][j][1] = img[2 * (ty * cols + tx) + 1];
        }
    for (int k = 0; k < 2; ++k)
        o[k] = val[0][0][k] * (1 - u) * (1 - v) + val[0][1][k] * (1 - u) * v + val[1][0][k] * u * (1 - v) +
               val[1][1][k] * u * v;
}

/* the temporal concatenation of two consecutive flows, pyramid.cu:406-442:
 * f(p) += BiLinear(f_next, p + f(p)) */
void vmo_flow_concat(float *f, const float *f_next, int w, int h)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float *p = f + 2 * ((size_t)y * w + x);
            float o[2];
            bilinear_f2(f_next, w, h, (float)x + p[0], (float)y + p[1], o);
            p[0] += o[0];
            p[1] += o[1];
        }
}
