/*
 * vm_oracle_poisson.c -- CPU ORACLE (test infrastructure, NOT the product
 * path): Poisson boundary extension, CPoissonExt::prepare / poissonExtend,
 * Algorithm/PoissonExt.cpp:49-362 (citations relative to /root/reference).
 *
 * The reference factorises the 5-point system with Intel MKL DSS
 * (PoissonExt.cpp:321-329; MKL 2015 is an un-vendored binary dependency,
 * README.txt:14) in single precision.  MKL is absent, so the arithmetic of
 * that solve cannot be reproduced: "parity unpinned" for the solve.  The
 * matrix is a symmetric, irreducibly diagonally dominant M-matrix, so the
 * solution is unique; the oracle solves the SAME system (assembled exactly as
 * PoissonExt.cpp:214-312) by conjugate gradients in double precision to a
 * relative residual `tol`, which any correct solver must agree with to within
 * the final integer truncation (+-1 colour level).
 */
#include "vm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* BilineaGetColor_clamp<Vec2f,Vec2f>, PoissonExt.cpp:367-397 */
static void bil_v2(const float *img, int cols, int rows, float px, float py, float *out)
{
    int x[2], y[2];
    x[0] = (int)floorf(px); y[0] = (int)floorf(py);
    x[1] = (int)ceilf(px);  y[1] = (int)ceilf(py);
    float u = px - x[0], v = py - y[0];
    for (int c = 0; c < 2; ++c) {
        float val[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                val[i][j] = img[2 * ((size_t)clampi(y[j], 0, rows - 1) * cols + clampi(x[i], 0, cols - 1)) + c];
        out[c] = val[0][0] * (1 - u) * (1 - v) + val[0][1] * (1 - u) * v +
                 val[1][0] * u * (1 - v) + val[1][1] * u * v;
    }
}

/* BilineaGetColor_clamp<Vec4b,Vec4f>, then the Vec4f -> Vec4b conversion of
 * `Vec4b rgba = ...` (cv::saturate_cast<uchar>: round half to even, clamp) */
static void bil_rgba8(const uint8_t *img, int cols, int rows, float px, float py, uint8_t *out)
{
    int x[2], y[2];
    x[0] = (int)floorf(px); y[0] = (int)floorf(py);
    x[1] = (int)ceilf(px);  y[1] = (int)ceilf(py);
    float u = px - x[0], v = py - y[0];
    for (int c = 0; c < 4; ++c) {
        float val[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                val[i][j] = (float)img[4 * ((size_t)clampi(y[j], 0, rows - 1) * cols + clampi(x[i], 0, cols - 1)) + c];
        float f = val[0][0] * (1 - u) * (1 - v) + val[0][1] * (1 - u) * v +
                  val[1][0] * u * (1 - v) + val[1][1] * u * v;
        long r = lrintf(f); /* default rounding mode: nearest even */
        out[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

/* CPoissonExt::prepare, PoissonExt.cpp:49-141 */
int vmo_poisson_prepare(uint8_t *ext, int w, int h, int ex,
                        const uint8_t *other, const float *vec, int side, int *type)
{
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    const int sign = side == 1 ? 1 : -1;
    int size = 0;
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) {
            int ii = y * cw + x;
            if (ext[4 * (size_t)ii + 3] > 0) { type[ii] = 2; size++; continue; }
            if ((y > 0 && ext[4 * (size_t)(ii - cw) + 3] > 0) ||
                (y < ch - 1 && ext[4 * (size_t)(ii + cw) + 3] > 0) ||
                (x > 0 && ext[4 * (size_t)(ii - 1) + 3] > 0) ||
                (x < cw - 1 && ext[4 * (size_t)(ii + 1) + 3] > 0)) {
                type[ii] = 1; size++; continue;
            }
            type[ii] = 0;
        }
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) {
            int ii = y * cw + x;
            if (type[ii] != 2)
                continue;
            float q[2] = {(float)(x - ex), (float)(y - ex)}, p[2], v[2], t[2];
            p[0] = q[0]; p[1] = q[1];
            bil_v2(vec, w, h, p[0], p[1], v);
            const float a = 0.8f;
            for (int i = 0; i < 20; ++i) {
                p[0] = q[0] + v[0] * sign;
                p[1] = q[1] + v[1] * sign;
                bil_v2(vec, w, h, p[0], p[1], t);
                v[0] = a * t[0] + (1 - a) * v[0];
                v[1] = a * t[1] + (1 - a) * v[1];
            }
            q[0] = p[0] + v[0] * sign;
            q[1] = p[1] + v[1] * sign;
            uint8_t *d = ext + 4 * (size_t)ii;
            if (q[0] >= 0 && q[1] >= 0 && q[0] < w && q[1] < h) {
                uint8_t rgba[4];
                bil_rgba8(other, w, h, q[0], q[1], rgba);
                if (rgba[3] == 0) { d[0] = rgba[0]; d[1] = rgba[1]; d[2] = rgba[2]; d[3] = rgba[3]; }
                else { d[0] = 255; d[1] = 0; d[2] = 255; d[3] = 0; }
            } else { d[0] = 255; d[1] = 0; d[2] = 255; d[3] = 0; }
        }
    return size;
}

static int is_marker(const uint8_t *p) { return p[0] == 255 && p[1] == 0 && p[2] == 255 && p[3] == 0; }

/* y = A x for the 5-point system of PoissonExt.cpp:214-312 (matrix-free) */
static void apply_A(const int *type, const float *diag, int cw, int ch, const double *x, double *y)
{
    const int nt = vmo_get_threads(); /* never the machine's logical CPU count: containers cap it */
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int yy = 0; yy < ch; ++yy)
        for (int xx = 0; xx < cw; ++xx) {
            int ii = yy * cw + xx;
            if (type[ii] == 0) { y[ii] = 0; continue; }
            double s = (double)diag[ii] * x[ii];
            if (yy - 1 >= 0 && type[ii - cw] > 0) s -= x[ii - cw];
            if (xx - 1 >= 0 && type[ii - 1] > 0) s -= x[ii - 1];
            if (xx + 1 < cw && type[ii + 1] > 0) s -= x[ii + 1];
            if (yy + 1 < ch && type[ii + cw] > 0) s -= x[ii + cw];
            y[ii] = s;
        }
}

int vmo_poisson_extend(uint8_t *ext, int w, int h, int ex,
                       const uint8_t *other, const float *vec, int side,
                       double tol, int max_it, double *rel_res)
{
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    const size_t N = (size_t)cw * ch;
    int *type = (int *)malloc(N * sizeof(int));
    vmo_poisson_prepare(ext, w, h, ex, other, vec, side, type);

    /* gradients, PoissonExt.cpp:146-183 */
    float *gx = (float *)calloc(N * 3, sizeof(float));
    float *gy = (float *)calloc(N * 3, sizeof(float));
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) {
            int ii = y * cw + x;
            if (type[ii] <= 1) continue;
            const uint8_t *c1 = ext + 4 * (size_t)ii;
            if (x > 0 && type[ii - 1] > 1) {
                const uint8_t *c0 = ext + 4 * (size_t)(ii - 1);
                if (!is_marker(c0) && !is_marker(c1))
                    for (int c = 0; c < 3; ++c) gx[3 * (size_t)ii + c] = (float)c1[c] - (float)c0[c];
            }
            if (y > 0 && type[ii - cw] > 1) {
                const uint8_t *c0 = ext + 4 * (size_t)(ii - cw);
                if (!is_marker(c0) && !is_marker(c1))
                    for (int c = 0; c < 3; ++c) gy[3 * (size_t)ii + c] = (float)c1[c] - (float)c0[c];
            }
        }
    /* system, PoissonExt.cpp:214-312 */
    float *diag = (float *)calloc(N, sizeof(float));
    float *B = (float *)calloc(N * 3, sizeof(float));
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) {
            int ii = y * cw + x;
            if (type[ii] == 0) continue;
            float a2 = 0, b[3] = {0, 0, 0};
            if (type[ii] == 1) {
                a2 += 1.0f;
                for (int c = 0; c < 3; ++c) b[c] += (float)ext[4 * (size_t)ii + c];
            }
            if (y - 1 >= 0 && type[ii - cw] > 0) { a2 += 1.0f; for (int c = 0; c < 3; ++c) b[c] += gy[3 * (size_t)ii + c]; }
            if (x - 1 >= 0 && type[ii - 1] > 0) { a2 += 1.0f; for (int c = 0; c < 3; ++c) b[c] += gx[3 * (size_t)ii + c]; }
            if (x + 1 < cw && type[ii + 1] > 0) { a2 += 1.0f; for (int c = 0; c < 3; ++c) b[c] -= gx[3 * (size_t)(ii + 1) + c]; }
            if (y + 1 < ch && type[ii + cw] > 0) { a2 += 1.0f; for (int c = 0; c < 3; ++c) b[c] -= gy[3 * (size_t)(ii + cw) + c]; }
            diag[ii] = a2;
            for (int c = 0; c < 3; ++c) B[3 * (size_t)ii + c] = b[c];
        }
    /* conjugate gradients per channel, Jacobi-preconditioned, double */
    double *x = (double *)malloc(N * sizeof(double)), *r = (double *)malloc(N * sizeof(double));
    double *z = (double *)malloc(N * sizeof(double)), *p = (double *)malloc(N * sizeof(double));
    double *q = (double *)malloc(N * sizeof(double));
    int total_it = 0;
    double worst = 0;
    for (int c = 0; c < 3; ++c) {
        double bnorm = 0;
        for (size_t i = 0; i < N; ++i) {
            x[i] = 0;
            r[i] = type[i] ? (double)B[3 * i + c] : 0;
            bnorm += r[i] * r[i];
        }
        bnorm = sqrt(bnorm);
        double rz = 0;
        for (size_t i = 0; i < N; ++i) {
            z[i] = type[i] ? r[i] / diag[i] : 0;
            p[i] = z[i];
            rz += r[i] * z[i];
        }
        double rn = bnorm;
        int it = 0;
        while (bnorm > 0 && rn > tol * bnorm && it < max_it) {
            apply_A(type, diag, cw, ch, p, q);
            double pq = 0;
            for (size_t i = 0; i < N; ++i) pq += p[i] * q[i];
            double alpha = rz / pq;
            double rz_new = 0, rr = 0;
            for (size_t i = 0; i < N; ++i) {
                if (!type[i]) continue;
                x[i] += alpha * p[i];
                r[i] -= alpha * q[i];
                z[i] = r[i] / diag[i];
                rz_new += r[i] * z[i];
                rr += r[i] * r[i];
            }
            double beta = rz_new / rz;
            rz = rz_new;
            for (size_t i = 0; i < N; ++i) p[i] = z[i] + beta * p[i];
            rn = sqrt(rr);
            ++it;
        }
        total_it += it;
        if (bnorm > 0 && rn / bnorm > worst) worst = rn / bnorm;
        /* paste, PoissonExt.cpp:333-346 */
        for (size_t i = 0; i < N; ++i)
            if (type[i] > 0) {
                float xf = (float)x[i];
                int val = (int)fminf(fmaxf(xf, 0), 255);
                ext[4 * i + c] = (uint8_t)val;
            }
    }
    for (size_t i = 0; i < N; ++i)
        if (type[i] > 0) ext[4 * i + 3] = 0;
    if (rel_res) *rel_res = worst;
    free(type); free(gx); free(gy); free(diag); free(B);
    free(x); free(r); free(z); free(p); free(q);
    return total_it;
}
