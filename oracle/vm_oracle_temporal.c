/*
 * vm_oracle_temporal.c -- CPU ORACLE (test infrastructure, NOT the product path):
 * the temporal coherence path of the halfway optimizer -- temp_ref,
 * interpolate_temp_ref, smooth, fill_zeros_x/y, kernel_initialize_temp and
 * initialize_temp (Algorithm/upsample.cu:28-258) and the temporal half of
 * upsample() (upsample.cu:297-338).  The `flag == true` energy term itself
 * (morph.cu:752-759) is in vm_oracle.c:energy_change.
 *
 * PARITY STATUS: unpinned by reference-run outputs (upsample.cu is CUDA with texture
 * references; nvcc is absent).  Pinned by analytic known-answer tests
 * (tests/test_oracle_kat.py: constant flows translate the field, zero flows reproduce
 * it, hole filling by rows).
 *
 * ORDER: temp_ref accumulates with float atomicAdd in an unspecified order
 * (upsample.cu:57-58).  Each contribution is computed in the reference's own float /
 * double expressions and then accumulated in 64-bit fixed point (x 2^32, round to
 * nearest even), which is order-independent; see vm_oracle.h.
 */
#include "vm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int inside(int w, int h, int x, int y) { return x >= 0 && x < w && y >= 0 && y < h; }

static inline int64_t to_fixed(float c) { return (int64_t)llrint((double)c * 4294967296.0); }

/* temp_ref, upsample.cu:28-62.  tex_f0 / tex_f1 are float2 textures with linear
 * filtering and clamp addressing (upsample.cu:227-233): vmo_tex2d_f2. */
void vmo_temp_splat(int w, int h, const float *v_prev, const float *f0, const float *f1,
                    const float *ssim, int64_t *acc)
{
    for (int py = 0; py < h; ++py)
        for (int px = 0; px < w; ++px) {
            const int idx = py * w + px;
            const float p_x = (float)px, p_y = (float)py;
            const float vx = v_prev[2 * idx], vy = v_prev[2 * idx + 1];
            float a[2], b[2];
            vmo_tex2d_f2(f0, w, h, p_x - vx + 0.5f, p_y - vy + 0.5f, a);
            vmo_tex2d_f2(f1, w, h, p_x + vx + 0.5f, p_y + vy + 0.5f, b);
            /* float2 p_ref = p + 0.5*(f0+f1); float2 v_ref = v + 0.5*(f1-f0);  (0.5 binds to the
             * float overload of operator*, include/util/dmath.h:752) */
            const float prx = p_x + 0.5f * (a[0] + b[0]), pry = p_y + 0.5f * (a[1] + b[1]);
            const float vrx = vx + 0.5f * (b[0] - a[0]), vry = vy + 0.5f * (b[1] - a[1]);
            const int xx = (int)floorf(prx), yy = (int)floorf(pry);
            for (int y = yy; y <= yy + 1; ++y)
                for (int x = xx; x <= xx + 1; ++x) {
                    if (!inside(w, h, x, y))
                        continue;
                    float ssim_fa = 1;
                    if (ssim)
                        ssim_fa = ssim[idx];
                    /* `ssim_fa*(1.0-abs((float)x-p_ref.x))*(1.0-abs((float)y-p_ref.y))`: the
                     * literals are double, the product is formed in double and rounded once */
                    const float fa = (float)((double)ssim_fa * (1.0 - (double)fabsf((float)x - prx)) *
                                             (1.0 - (double)fabsf((float)y - pry)));
                    int64_t *d = acc + 3 * (size_t)(y * w + x);
                    d[0] += to_fixed(vrx * fa);
                    d[1] += to_fixed(vry * fa);
                    d[2] += to_fixed(fa);
                }
        }
}

/* fixed point -> float; interpolate_temp_ref, upsample.cu:64-77 */
void vmo_temp_normalise(int w, int h, const int64_t *acc, float *v_cur, float *weight)
{
    const size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; ++i) {
        float x = (float)((double)acc[3 * i] / 4294967296.0);
        float y = (float)((double)acc[3 * i + 1] / 4294967296.0);
        const float wt = (float)((double)acc[3 * i + 2] / 4294967296.0);
        if (wt > 0) {
            x /= wt;
            y /= wt;
        }
        v_cur[2 * i] = x;
        v_cur[2 * i + 1] = y;
        weight[i] = wt;
    }
}

/* initialize_temp (upsample.cu:214-258) with kernel_initialize_temp (:190-211) */
void vmo_initialize_temp(vmo_level *dst, const vmo_level *src, const float *fa, const float *fb)
{
    const int w = dst->w, h = dst->h;
    const size_t n = (size_t)w * h;
    int64_t *acc = (int64_t *)calloc(3 * n, sizeof(int64_t));
    float *ref_v = (float *)calloc(2 * n, sizeof(float));
    float *weight = (float *)calloc(n, sizeof(float));
    vmo_temp_splat(w, h, src->v, fa, fb, src->value, acc);
    vmo_temp_normalise(w, h, acc, ref_v, weight);
    for (size_t i = 0; i < n; ++i) {
        if (weight[i] > 0) {
            dst->temp_ref[2 * i] = ref_v[2 * i];
            dst->temp_ref[2 * i + 1] = ref_v[2 * i + 1];
            dst->temp_mask[i] = weight[i];
        } else {
            dst->temp_mask[i] = 0.0f;
        }
    }
    dst->flag = 1;
    free(acc);
    free(ref_v);
    free(weight);
}

/* smooth, upsample.cu:80-111: mean of the valid (weight > 0) pixels of the 3x3 window;
 * v_out keeps its zero where there is none */
static void smooth(int w, int h, float *v_out, const float *v_cur, const float *weight)
{
    for (int py = 0; py < h; ++py)
        for (int px = 0; px < w; ++px) {
            float ww = 0.0f, sx = 0, sy = 0;
            for (int y = py - 1; y <= py + 1; ++y)
                for (int x = px - 1; x <= px + 1; ++x) {
                    if (!inside(w, h, x, y))
                        continue;
                    const int idx = y * w + x;
                    if (weight[idx] > 0) {
                        ww += 1;
                        sx += v_cur[2 * idx];
                        sy += v_cur[2 * idx + 1];
                    }
                }
            if (ww > 0) {
                v_out[2 * (py * w + px)] = sx / ww;
                v_out[2 * (py * w + px) + 1] = sy / ww;
            }
        }
}

/* fill_zeros_x, upsample.cu:115-151: a pixel without weight takes (v_left + v_right) /
 * (1/d_left + 1/d_right) of the nearest weighted pixels of its row -- the sum of the two
 * values is NOT weighted by the inverse distances (as written there).  Only pixels with
 * weight <= 0 are written and only pixels with weight > 0 are read: no ordering issue. */
static void fill_zeros_x(int w, int h, float *v_out, const float *weight)
{
    for (int py = 0; py < h; ++py)
        for (int px = 0; px < w; ++px) {
            const int idx = py * w + px;
            if (weight[idx] > 0)
                continue;
            float ww = 0.0f, sx = 0, sy = 0;
            for (int x = px; x >= 0; --x)
                if (weight[py * w + x] > 0) {
                    ww = (float)((double)ww + 1.0 / (px - x)); /* `ww+=1.0/(pos.x-x)`: double */
                    sx += v_out[2 * (py * w + x)];
                    sy += v_out[2 * (py * w + x) + 1];
                    break;
                }
            for (int x = px; x < w; ++x)
                if (weight[py * w + x] > 0) {
                    ww = (float)((double)ww + 1.0 / (x - px));
                    sx += v_out[2 * (py * w + x)];
                    sy += v_out[2 * (py * w + x) + 1];
                    break;
                }
            if (ww > 0) {
                v_out[2 * idx] = sx / ww;
                v_out[2 * idx + 1] = sy / ww;
            }
        }
}

/* the in-between page of upsample(), upsample.cu:303-337.  fill_zeros_y (:153-189) computes
 * a column average it never stores: it only sets weight[idx] = 1, and the weight array is
 * freed right after (:333) -- no effect on the result, so it is not restated. */
void vmo_temporal_fill(int w, int h, const float *v_prev, const float *f0_prev, const float *f1_prev,
                       const float *v_next, const float *b0_next, const float *b1_next, float *v_out)
{
    const size_t n = (size_t)w * h;
    int64_t *acc = (int64_t *)calloc(3 * n, sizeof(int64_t));
    float *v_cur = (float *)calloc(2 * n, sizeof(float));
    float *weight = (float *)calloc(n, sizeof(float));
    vmo_temp_splat(w, h, v_prev, f0_prev, f1_prev, NULL, acc);
    vmo_temp_splat(w, h, v_next, b0_next, b1_next, NULL, acc);
    vmo_temp_normalise(w, h, acc, v_cur, weight);
    memset(v_out, 0, 2 * n * sizeof(float));
    smooth(w, h, v_out, v_cur, weight);
    fill_zeros_x(w, h, v_out, weight);
    free(acc);
    free(v_cur);
    free(weight);
}
