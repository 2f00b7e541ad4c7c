/*
 * vm_oracle_render.c -- CPU ORACLE (test infrastructure, NOT the product
 * path): compositor kernel and result delivery.  See vm_oracle.h.
 * Citations relative to /root/reference.
 */
#include "vm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* tex2D on a float4 canvas (linear, clamp, unnormalised), exact float weights */
static void tex2d_f4(const float *img, int w, int h, float x, float y, float *out4)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = xb - fi, b = yb - fj;
    fi = fminf(fmaxf(fi, -1.0f), (float)w);
    fj = fminf(fmaxf(fj, -1.0f), (float)h);
    int i0 = clampi((int)fi, 0, w - 1), i1 = clampi((int)fi + 1, 0, w - 1);
    int j0 = clampi((int)fj, 0, h - 1), j1 = clampi((int)fj + 1, 0, h - 1);
    for (int c = 0; c < 4; ++c) {
        float t00 = img[4 * ((size_t)j0 * w + i0) + c], t10 = img[4 * ((size_t)j0 * w + i1) + c];
        float t01 = img[4 * ((size_t)j1 * w + i0) + c], t11 = img[4 * ((size_t)j1 * w + i1) + c];
        out4[c] = (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
    }
}

/* kernel_render_halfway_image, render.cu:16-60 */
void vmo_render_halfway(uint8_t *out, int w, int h, int ex,
                        float color_fa, float geo_fa, int color_from,
                        const float *ext0, const float *ext1,
                        const float *vf, const float *uf)
{
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    const float alpha = 0.8f;
    const float s1 = 2 * geo_fa - 1;
    const float s2 = 4 * geo_fa - 4 * geo_fa * geo_fa;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float qx = (float)x, qy = (float)y, px = qx, py = qy;
            float v[2], u[2], t[2];
            vmo_tex2d_f2(vf, w, h, px + 0.5f, py + 0.5f, v);
            vmo_tex2d_f2(uf, w, h, px + 0.5f, py + 0.5f, u);
            for (int i = 0; i < 20; ++i) {
                px = qx - s1 * v[0] - s2 * u[0];
                py = qy - s1 * v[1] - s2 * u[1];
                vmo_tex2d_f2(vf, w, h, px + 0.5f, py + 0.5f, t);
                v[0] = alpha * t[0] + (1 - alpha) * v[0];
                v[1] = alpha * t[1] + (1 - alpha) * v[1];
                vmo_tex2d_f2(uf, w, h, px + 0.5f, py + 0.5f, t);
                u[0] = alpha * t[0] + (1 - alpha) * u[0];
                u[1] = alpha * t[1] + (1 - alpha) * u[1];
            }
            float c0[4], c1[4];
            tex2d_f4(ext0, cw, ch, px - v[0] + ex + 0.5f, py - v[1] + ex + 0.5f, c0);
            tex2d_f4(ext1, cw, ch, px + v[0] + ex + 0.5f, py + v[1] + ex + 0.5f, c1);
            uint8_t *o = out + 3 * ((size_t)y * w + x);
            for (int c = 0; c < 3; ++c) {
                double val;
                switch (color_from) {
                case 0: val = c0[c] + 0.5; break;
                case 1: val = c0[c] * (1 - color_fa) + c1[c] * color_fa + 0.5; break;
                default: val = c1[c] + 0.5; break;
                }
                o[c] = (uint8_t)val; /* make_uchar3: truncation, render.cu:49-56 */
            }
        }
}

/* CMatchingThread::BiLinear, MatchingThread.cpp:103-136 (Vec2f) */
static void bilinear_v2(const float *img, int cols, int rows, float px, float py, float *out2)
{
    int x[2], y[2];
    x[0] = (int)floorf(px);
    y[0] = (int)floorf(py);
    x[1] = (int)ceilf(px);
    y[1] = (int)ceilf(py);
    float u = px - x[0], v = py - y[0];
    for (int c = 0; c < 2; ++c) {
        float val[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                int tx = clampi(x[i], 0, cols - 1), ty = clampi(y[j], 0, rows - 1);
                val[i][j] = img[2 * ((size_t)ty * cols + tx) + c];
            }
        out2[c] = val[0][0] * (1 - u) * (1 - v) + val[0][1] * (1 - u) * v +
                  val[1][0] * u * (1 - v) + val[1][1] * u * v;
    }
}

/* CMatchingThread::update_result + Resize, MatchingThread.cpp:22-100:
 * scale by (W0/W, H0/H), then bilinear resize to w0 x h0 (a copy when the
 * sizes already agree) */
void vmo_upscale_result(float *dst, int w0, int h0, const float *v, int w, int h)
{
    float rx = (float)w0 / (float)w, ry = (float)h0 / (float)h;
    float *tmp = (float *)malloc(sizeof(float) * 2 * (size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        if (rx != 1 || ry != 1) {
            tmp[2 * i] = v[2 * i] * rx;
            tmp[2 * i + 1] = v[2 * i + 1] * ry;
        } else {
            tmp[2 * i] = v[2 * i];
            tmp[2 * i + 1] = v[2 * i + 1];
        }
    }
    if (w == w0 && h == h0) {
        memcpy(dst, tmp, sizeof(float) * 2 * (size_t)w * h);
    } else {
        for (int y = 0; y < h0; ++y)
            for (int x = 0; x < w0; ++x) {
                float fy = (float)((y + 0.5) / h0 * h - 0.5);
                float fx = (float)((x + 0.5) / w0 * w - 0.5);
                bilinear_v2(tmp, w, h, fx, fy, dst + 2 * ((size_t)y * w0 + x));
            }
    }
    free(tmp);
}

/* the temporal half of CMatchingThread::update_result (MatchingThread.cpp:62-78): the frames the
 * temporal pyramid skipped, _vector[beg] * (1 - fa) + _vector[end] * fa.  A cv::Mat expression:
 * OpenCV evaluates it as addWeighted in float -- two products and their sum per component (the
 * binary is not in the reference tree: parity of this one line is unpinned). */
void vmo_blend_v(float *dst, const float *a, const float *b, float fa, size_t n_floats)
{
    const float alpha = 1 - fa, beta = fa;
    for (size_t i = 0; i < n_floats; ++i) {
        const float t = a[i] * alpha, u = b[i] * beta;
        dst[i] = t + u;
    }
}
