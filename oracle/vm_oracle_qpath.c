/*
 * vm_oracle_qpath.c -- CPU ORACLE (test infrastructure, NOT the product path): the quadratic
 * motion path of one frame, CQuadraticPath::optimize, Algorithm/QuadraticPath.cpp:24-223
 * (citations relative to /root/reference).
 *
 * Per pixel the Jacobians J0 = I - grad v and J1 = I + grad v (backward differences, forward on
 * the first row/column, :37-73) are blended column-wise: directions averaged, lengths by their
 * geometric mean (:75-109); u solves the Neumann Poisson problem div(grad u) = div(J_opt - I)
 * assembled at :111-203.  The reference runs 10 001 unpreconditioned float CG iterations from
 * u = 0 through cuSPARSE/cuBLAS (:226-305, tol 1e-12 is never met): its rounding cannot be
 * reproduced ("parity unpinned" for the solve).  The matrix is the singular graph Laplacian and
 * the right-hand side sums to zero, so CG from zero converges to THE zero-mean solution; the
 * oracle computes that one in double precision to a relative residual `tol`.
 */
#include "vm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* :37-109 */
static void j_opt_at(const float *v, int cols, int rows, int x, int y, float *jo)
{
#define V(yy, xx, c) v[2 * ((size_t)(yy) * cols + (xx)) + (c)]
    float j0[4], j1[4];
    float vx_x, vy_x, vx_y, vy_y;
    if (x == 0) { vx_x = V(y, x + 1, 0) - V(y, x, 0); vy_x = V(y, x + 1, 1) - V(y, x, 1); }
    else        { vx_x = V(y, x, 0) - V(y, x - 1, 0); vy_x = V(y, x, 1) - V(y, x - 1, 1); }
    j0[0] = 1.0f - vx_x; j0[2] = -vy_x; j1[0] = 1.0f + vx_x; j1[2] = vy_x;
    if (y == 0) { vx_y = V(y + 1, x, 0) - V(y, x, 0); vy_y = V(y + 1, x, 1) - V(y, x, 1); }
    else        { vx_y = V(y, x, 0) - V(y - 1, x, 0); vy_y = V(y, x, 1) - V(y - 1, x, 1); }
    j0[1] = -vx_y; j0[3] = 1.0f - vy_y; j1[1] = vx_y; j1[3] = 1.0f + vy_y;
#undef V
    float nj0[4], nj1[4];
    float la0 = sqrtf(j0[0] * j0[0] + j0[2] * j0[2]), lb0 = sqrtf(j0[1] * j0[1] + j0[3] * j0[3]);
    nj0[0] = j0[0] / la0; nj0[2] = j0[2] / la0; nj0[1] = j0[1] / lb0; nj0[3] = j0[3] / lb0;
    float la1 = sqrtf(j1[0] * j1[0] + j1[2] * j1[2]), lb1 = sqrtf(j1[1] * j1[1] + j1[3] * j1[3]);
    nj1[0] = j1[0] / la1; nj1[2] = j1[2] / la1; nj1[1] = j1[1] / lb1; nj1[3] = j1[3] / lb1;
    float nj[4];
    for (int i = 0; i < 4; ++i) nj[i] = nj0[i] + nj1[i];
    float la = sqrtf(nj[0] * nj[0] + nj[2] * nj[2]), lb = sqrtf(nj[1] * nj[1] + nj[3] * nj[3]);
    nj[0] /= la; nj[2] /= la; nj[1] /= lb; nj[3] /= lb;
    la = sqrtf(la0 * la1);
    lb = sqrtf(lb0 * lb1);
    jo[0] = nj[0] * la; jo[2] = nj[2] * la; jo[1] = nj[1] * lb; jo[3] = nj[3] * lb;
}

static void lap(const double *x, double *y, int cols, int rows)
{
    for (int yy = 0; yy < rows; ++yy)
        for (int xx = 0; xx < cols; ++xx) {
            size_t ii = (size_t)yy * cols + xx;
            double s = 0;
            if (yy > 0) s += x[ii] - x[ii - cols];
            if (xx > 0) s += x[ii] - x[ii - 1];
            if (xx + 1 < cols) s += x[ii] - x[ii + 1];
            if (yy + 1 < rows) s += x[ii] - x[ii + cols];
            y[ii] = s;
        }
}

/* v: rows*cols*2 floats; u_out likewise.  Returns the CG iterations of both channels. */
int vmo_quadratic_path(const float *v, int cols, int rows, double tol, int max_it, float *u_out,
                       float *jopt_out, double *rel_res)
{
    const size_t N = (size_t)cols * rows;
    float *jo = (float *)malloc(N * 4 * sizeof(float));
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            j_opt_at(v, cols, rows, x, y, jo + 4 * ((size_t)y * cols + x));
    if (jopt_out) memcpy(jopt_out, jo, N * 4 * sizeof(float));
    /* right-hand sides, :137-170 (float accumulation in this order) */
    float *B = (float *)calloc(N * 2, sizeof(float));
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            size_t ii = (size_t)y * cols + x;
            float bx = 0, by = 0;
            if (y - 1 >= 0) { bx += jo[4 * ii + 1]; by += jo[4 * ii + 3] - 1.0f; }
            if (x - 1 >= 0) { bx += jo[4 * ii + 0] - 1.0f; by += jo[4 * ii + 2]; }
            if (x + 1 < cols) { bx -= jo[4 * (ii + 1) + 0] - 1.0f; by -= jo[4 * (ii + 1) + 2]; }
            if (y + 1 < rows) { bx -= jo[4 * (ii + cols) + 1]; by -= jo[4 * (ii + cols) + 3] - 1.0f; }
            B[2 * ii] = bx;
            B[2 * ii + 1] = by;
        }
    double *x = (double *)malloc(N * sizeof(double)), *r = (double *)malloc(N * sizeof(double));
    double *p = (double *)malloc(N * sizeof(double)), *q = (double *)malloc(N * sizeof(double));
    int total = 0;
    double worst = 0;
    for (int c = 0; c < 2; ++c) {
        double bn = 0, mean = 0;
        for (size_t i = 0; i < N; ++i) mean += B[2 * i + c];
        mean /= (double)N; /* rounding leaves the float sums a hair off zero: project */
        for (size_t i = 0; i < N; ++i) { x[i] = 0; r[i] = B[2 * i + c] - mean; p[i] = r[i]; bn += r[i] * r[i]; }
        double rr = bn;
        int it = 0;
        while (bn > 0 && sqrt(rr / bn) > tol && it < max_it) {
            lap(p, q, cols, rows);
            double pq = 0;
            for (size_t i = 0; i < N; ++i) pq += p[i] * q[i];
            double a = rr / pq, rr2 = 0;
            for (size_t i = 0; i < N; ++i) { x[i] += a * p[i]; r[i] -= a * q[i]; rr2 += r[i] * r[i]; }
            double be = rr2 / rr;
            rr = rr2;
            for (size_t i = 0; i < N; ++i) p[i] = r[i] + be * p[i];
            ++it;
        }
        total += it;
        if (bn > 0 && sqrt(rr / bn) > worst) worst = sqrt(rr / bn);
        double m = 0;
        for (size_t i = 0; i < N; ++i) m += x[i];
        m /= (double)N;
        for (size_t i = 0; i < N; ++i) u_out[2 * i + c] = (float)(x[i] - m);
    }
    if (rel_res) *rel_res = worst;
    free(jo); free(B); free(x); free(r); free(p); free(q);
    return total;
}
