/*
 * vm_oracle_sync.c -- CPU ORACLE (test infrastructure, NOT the product path):
 * the synchronisation stage that precedes the morph in the reference's app --
 * SURVEY section 8(f) "(later)" row.  See vm_oracle.h.  Citations relative to
 * /root/reference.
 *
 *   sync pyramid geometry      Pyramid::build(video0, video1, f0, f1, start_res), pyramid.cu:57-165
 *   system A x = b per level   CSyncThread::genMatrix, SyncThread.cpp:129-289 (matrix-free here)
 *   CG per component           CSyncThread::optimize_level, SyncThread.cpp:290-480
 *   level transfer             CSyncThread::upsample_level + Kernel_upsample, SyncThread.cpp:103-128,
 *                              upsample.cu:343-375
 *   result delivery            CSyncThread::update_result, SyncThread.cpp:482-521 (cv::resize, linear)
 *   time-warped resampling     kernel_render_resample_image0/1, render.cu:99-199
 *
 * PARITY UNPINNED: the reference holds no test or golden vector for this stage, and its
 * arithmetic lives in cuBLAS / cuSPARSE (CUDA 6.5) and cv::resize (OpenCV 3.0), none of which is
 * here.  What the libraries leave unspecified is FIXED here (and mirrored by the HIP path):
 *   - csrmv accumulates a row in CSR order (z, then y, then x ascending) by fused multiply-adds
 *     in float (a GPU library's a * x + y is one FMA);
 *   - sdot multiplies in float and accumulates in double in a blocked order: bricks of
 *     32 x 8 x 8 voxels; inside a brick each (x, y) column of 8 voxels sequentially, the 64 columns
 *     of two brick rows by a butterfly (strides 32, 16, ... 1), the four row pairs in sequence; brick
 *     partials in groups t, t + 256, ... sequentially, then the same butterfly over 256 groups.
 *     Rounded to float once at the end (the reference keeps the scalars in float);
 *   - saxpy is one fused multiply-add per element (alpha * x + y), sscal one multiply.
 * Replicated quirks: the CG starts from the UPSAMPLED solution but with r = b, not b - A x
 * (SyncThread.cpp:372-375 with :95-117), so a level adds A^-1 b to what it inherited; the
 * constraint's frame midpoint is compared with z unscaled; the loop runs floor(max_iter) + 1
 * times (`k = 0; while (k <= _max_iter) k++`).
 * Not replicated: Kernel_upsample writes without a bounds check (upsample.cu:345-353: threads past
 * the right edge race with the next row's first threads) -- out-of-range threads do nothing here.
 */
#include "vm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* geometry, pyramid.cu:143-163 (level 0 = full resolution; levels 1.. are the solved ones)    */

int vmo_sync_levels(int w, int h, int d, int start_res, int *lw, int *lh, int *ld, int cap)
{
    int n = 0;
    if (cap < 1) return 0;
    lw[n] = w; lh[n] = h; ld[n] = d; ++n;
    float fa = (float)(w * h * d) / (float)4000000;
    float s = sqrtf(fa);
    fa = s > 1 ? s : 1;
    w = (int)((float)w / fa);
    h = (int)((float)h / fa);
    int el_t = 1;
    int el_y = (int)(logf((float)h) / logf(2.0f) - logf((float)start_res) / logf(2.0f) + 1);
    int el_x = (int)(logf((float)w) / logf(2.0f) - logf((float)start_res) / logf(2.0f) + 1);
    int maxl = el_x > el_y ? el_x : el_y;
    if (el_t > maxl) maxl = el_t;
    for (int el = 0; el < maxl && n < cap; ++el) {
        lw[n] = w; lh[n] = h; ld[n] = d; ++n;
        if (maxl - el <= el_x) w = (int)ceilf(w / 2.0f);
        if (maxl - el <= el_y) h = (int)ceilf(h / 2.0f);
        if (maxl - el <= el_t) d = (int)ceilf(d / 2.0f);
    }
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* one row of A as the 5 x 5 x 5 array genMatrix fills (SyncThread.cpp:190-262), in its order   */

typedef struct { int dz, dy, dx; float c; } sync_inc;

/* the increments one operator contributes to the row of voxel (x, y, z): `m` entries, each
 * (offset, coefficient x 2 w_tps) */
static void add_incs(float data[5][5][5], const sync_inc *inc, int m, float wt)
{
    for (int i = 0; i < m; ++i)
        data[2 + inc[i].dz][2 + inc[i].dy][2 + inc[i].dx] += inc[i].c * 2.0f * wt;
}

void vmo_sync_row(int x, int y, int z, int w, int h, int d, float w_tps, float ui, float *data125)
{
    float (*data)[5][5] = (float (*)[5][5])data125;
    memset(data125, 0, 125 * sizeof(float));
    data[2][2][2] += ui; /* the UI term is accumulated first, :155-187 */
    /* second differences: the three 1-2-1 operators that contain this voxel, per axis */
    for (int axis = 0; axis < 3; ++axis) {
        int p = axis == 0 ? x : (axis == 1 ? y : z), n = axis == 0 ? w : (axis == 1 ? h : d);
        int ax = axis == 0, ay = axis == 1, az = axis == 2;
        if (p > 1) {
            sync_inc a[3] = {{-2 * az, -2 * ay, -2 * ax, 1.0f}, {-az, -ay, -ax, -2.0f}, {0, 0, 0, 1.0f}};
            add_incs(data, a, 3, w_tps);
        }
        if (p > 0 && p < n - 1) {
            sync_inc a[3] = {{-az, -ay, -ax, -2.0f}, {0, 0, 0, 4.0f}, {az, ay, ax, -2.0f}};
            add_incs(data, a, 3, w_tps);
        }
        if (p < n - 2) {
            sync_inc a[3] = {{0, 0, 0, 1.0f}, {az, ay, ax, -2.0f}, {2 * az, 2 * ay, 2 * ax, 1.0f}};
            add_incs(data, a, 3, w_tps);
        }
    }
    /* mixed differences (weight 2): the four 2 x 2 cells that contain this voxel, per plane */
    if (x > 0 && y > 0)         { sync_inc a[4] = {{0,-1,-1, 2}, {0,-1, 0,-2}, {0, 0,-1,-2}, {0, 0, 0, 2}}; add_incs(data, a, 4, w_tps); }
    if (x < w - 1 && y > 0)     { sync_inc a[4] = {{0,-1, 0,-2}, {0,-1, 1, 2}, {0, 0, 0, 2}, {0, 0, 1,-2}}; add_incs(data, a, 4, w_tps); }
    if (x > 0 && y < h - 1)     { sync_inc a[4] = {{0, 0,-1,-2}, {0, 0, 0, 2}, {0, 1,-1, 2}, {0, 1, 0,-2}}; add_incs(data, a, 4, w_tps); }
    if (x < w - 1 && y < h - 1) { sync_inc a[4] = {{0, 0, 0, 2}, {0, 0, 1,-2}, {0, 1, 0,-2}, {0, 1, 1, 2}}; add_incs(data, a, 4, w_tps); }
    if (z > 0 && y > 0)         { sync_inc a[4] = {{-1,-1, 0, 2}, {-1, 0, 0,-2}, {0,-1, 0,-2}, {0, 0, 0, 2}}; add_incs(data, a, 4, w_tps); }
    if (z > 0 && y < h - 1)     { sync_inc a[4] = {{-1, 0, 0,-2}, {-1, 1, 0, 2}, {0, 0, 0, 2}, {0, 1, 0,-2}}; add_incs(data, a, 4, w_tps); }
    if (z < d - 1 && y > 0)     { sync_inc a[4] = {{0,-1, 0,-2}, {0, 0, 0, 2}, {1,-1, 0, 2}, {1, 0, 0,-2}}; add_incs(data, a, 4, w_tps); }
    if (z < d - 1 && y < h - 1) { sync_inc a[4] = {{0, 0, 0, 2}, {0, 1, 0,-2}, {1, 0, 0,-2}, {1, 1, 0, 2}}; add_incs(data, a, 4, w_tps); }
    if (x > 0 && z > 0)         { sync_inc a[4] = {{-1, 0,-1, 2}, {0, 0,-1,-2}, {-1, 0, 0,-2}, {0, 0, 0, 2}}; add_incs(data, a, 4, w_tps); }
    if (x > 0 && z < d - 1)     { sync_inc a[4] = {{0, 0,-1,-2}, {1, 0,-1, 2}, {0, 0, 0, 2}, {1, 0, 0,-2}}; add_incs(data, a, 4, w_tps); }
    if (x < w - 1 && z > 0)     { sync_inc a[4] = {{-1, 0, 0,-2}, {0, 0, 0, 2}, {-1, 0, 1, 2}, {0, 0, 1,-2}}; add_incs(data, a, 4, w_tps); }
    if (x < w - 1 && z < d - 1) { sync_inc a[4] = {{0, 0, 0, 2}, {1, 0, 0,-2}, {0, 0, 1,-2}, {1, 0, 1, 2}}; add_incs(data, a, 4, w_tps); }
}

/* the 25 positions a row can be non-zero at, in CSR order (z, y, x ascending) */
const int vmo_sync_taps[25][3] = {
    {-2, 0, 0},
    {-1,-1, 0}, {-1, 0,-1}, {-1, 0, 0}, {-1, 0, 1}, {-1, 1, 0},
    { 0,-2, 0}, { 0,-1,-1}, { 0,-1, 0}, { 0,-1, 1}, { 0, 0,-2}, { 0, 0,-1}, { 0, 0, 0}, { 0, 0, 1}, { 0, 0, 2},
    { 0, 1,-1}, { 0, 1, 0}, { 0, 1, 1}, { 0, 2, 0},
    { 1,-1, 0}, { 1, 0,-1}, { 1, 0, 0}, { 1, 0, 1}, { 1, 1, 0},
    { 2, 0, 0}};

/* The row pattern of genMatrix depends on a coordinate only through p > 1, p > 0, p < n - 1,
 * p < n - 2.  Positions with equal answers share a STATE: for n > 5 the five border classes
 * (0, 1 = distance to the low edge, 3, 4 = to the high edge, 2 = interior), for n <= 5 every
 * position is its own state. */
int vmo_sync_state(int p, int n)
{
    if (n <= 5 || p < 2) return p;
    if (p == n - 1) return 4;
    if (p == n - 2) return 3;
    return 2;
}

/* a position that is in state s */
static int sync_rep(int s, int n)
{
    if (n <= 5 || s <= 2) return s;
    return n - 5 + s;
}

/* off-diagonal (and UI-free diagonal) entries per state triple, [sz][sy][sx][tap] */
void vmo_sync_table(int w, int h, int d, float w_tps, float *tab125x25)
{
    memset(tab125x25, 0, 125 * 25 * sizeof(float));
    for (int sz = 0; sz < 5; ++sz)
        for (int sy = 0; sy < 5; ++sy)
            for (int sx = 0; sx < 5; ++sx) {
                int px = sync_rep(sx, w), py = sync_rep(sy, h), pz = sync_rep(sz, d);
                if (px >= w || py >= h || pz >= d) continue;
                float data[125];
                vmo_sync_row(px, py, pz, w, h, d, w_tps, 0.0f, data);
                for (int t = 0; t < 25; ++t) {
                    const int *o = vmo_sync_taps[t];
                    tab125x25[((sz * 5 + sy) * 5 + sx) * 25 + t] = data[((o[0] + 2) * 5 + (o[1] + 2)) * 5 + (o[2] + 2)];
                }
            }
}

/* ------------------------------------------------------------------------------------------ */
/* UI part of genMatrix (SyncThread.cpp:155-187): diagonal and right-hand sides.  The reference
 * visits every voxel and every constraint; visiting the constraints and the <= 8 voxels each
 * touches keeps the per-voxel accumulation order (constraint order).                           */

void vmo_sync_ui(int w, int h, int d, int w0, int h0, const int *cons6, int n, float w_ui,
                 float *diag, float *bx, float *by, float *bz)
{
    const size_t N = (size_t)w * h * d;
    memset(diag, 0, N * sizeof(float));
    memset(bx, 0, N * sizeof(float));
    memset(by, 0, N * sizeof(float));
    memset(bz, 0, N * sizeof(float));
    const float ratio_x = (float)w / (float)w0, ratio_y = (float)h / (float)h0;
    for (int c = 0; c < n; ++c) {
        const int *q = cons6 + 6 * c;
        float x0 = q[0] * ratio_x, y0 = q[1] * ratio_y, z0 = (float)q[2];
        float x1 = q[3] * ratio_x, y1 = q[4] * ratio_y, z1 = (float)q[5];
        float con_x = (x0 + x1) / 2.0f, con_y = (y0 + y1) / 2.0f, con_z = (z0 + z1) / 2.0f;
        float vx = (x1 - x0) / 2.0f, vy = (y1 - y0) / 2.0f, vz = (z1 - z0) / 2.0f;
        int xa = (int)floorf(con_x), ya = (int)floorf(con_y), za = (int)floorf(con_z);
        for (int z = za - 1; z <= za + 2; ++z)
            for (int y = ya - 1; y <= ya + 2; ++y)
                for (int x = xa - 1; x <= xa + 2; ++x) {
                    if (x < 0 || y < 0 || z < 0 || x >= w || y >= h || z >= d) continue;
                    float faz = fabsf(z - con_z), fay = fabsf(y - con_y), fax = fabsf(x - con_x);
                    if (faz < 1 && fay < 1 && fax < 1) {
                        float bw = (float)((1.0 - fax) * (1.0 - fay) * (1.0 - faz));
                        size_t idx = ((size_t)z * h + y) * w + x;
                        diag[idx] += bw * w_ui;
                        bx[idx] += bw * vx * w_ui;
                        by[idx] += bw * vy * w_ui;
                        bz[idx] += bw * vz * w_ui;
                    }
                }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* blocked dot product (the fixed summation order, see the header)                              */

#define SB_X 32
#define SB_Y 8
#define SB_Z 8

static double butterfly(double *s, int n)
{
    for (int off = n / 2; off >= 1; off /= 2)
        for (int l = 0; l < off; ++l) s[l] = s[l] + s[l + off];
    return s[0];
}

float vmo_sync_dot(const float *a, const float *b, int w, int h, int d)
{
    const int nbx = (w + SB_X - 1) / SB_X, nby = (h + SB_Y - 1) / SB_Y, nbz = (d + SB_Z - 1) / SB_Z;
    const int nb = nbx * nby * nbz;
    double *part = (double *)malloc((size_t)nb * sizeof(double));
    for (int bz = 0; bz < nbz; ++bz)
        for (int by = 0; by < nby; ++by)
            for (int bx = 0; bx < nbx; ++bx) {
                double pair[4];
                for (int wv = 0; wv < 4; ++wv) {
                    double s[64];
                    for (int l = 0; l < 64; ++l) {
                        int x = bx * SB_X + (l & 31), y = by * SB_Y + 2 * wv + (l >> 5);
                        double t = 0;
                        if (x < w && y < h)
                            for (int zz = 0; zz < SB_Z; ++zz) {
                                int z = bz * SB_Z + zz;
                                if (z >= d) break;
                                size_t i = ((size_t)z * h + y) * w + x;
                                float pr = a[i] * b[i];
                                t += (double)pr;
                            }
                        s[l] = t;
                    }
                    pair[wv] = butterfly(s, 64);
                }
                part[(bz * nby + by) * nbx + bx] = ((pair[0] + pair[1]) + pair[2]) + pair[3];
            }
    double g[256];
    for (int t = 0; t < 256; ++t) {
        double s = 0;
        for (int i = t; i < nb; i += 256) s += part[i];
        g[t] = s;
    }
    double pair[4];
    for (int wv = 0; wv < 4; ++wv) pair[wv] = butterfly(g + 64 * wv, 64);
    free(part);
    return (float)(((pair[0] + pair[1]) + pair[2]) + pair[3]);
}

/* ------------------------------------------------------------------------------------------ */
/* omega = A p, row by row in CSR order                                                         */

typedef struct {
    int w, h, d;
    float off[125][25]; /* off-diagonal pattern per state triple, vmo_sync_table */
    float *diag;        /* full diagonal, UI first then the stencil increments in order */
} sync_sys;

/* the diagonal of A: the UI term, then the stencil's increments in genMatrix's order */
void vmo_sync_diag(int w, int h, int d, float w_tps, const float *ui, float *diag)
{
#pragma omp parallel for num_threads(vmo_get_threads()) schedule(static)
    for (int z = 0; z < d; ++z)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float data[125];
                vmo_sync_row(x, y, z, w, h, d, w_tps, ui[((size_t)z * h + y) * w + x], data);
                diag[((size_t)z * h + y) * w + x] = data[62];
            }
}

static void sync_sys_build(sync_sys *S, int w, int h, int d, float w_tps, const float *ui)
{
    S->w = w; S->h = h; S->d = d;
    S->diag = (float *)malloc((size_t)w * h * d * sizeof(float));
    vmo_sync_diag(w, h, d, w_tps, ui, S->diag);
}

void vmo_sync_apply(int w, int h, int d, float w_tps, const float *ui, const float *p, float *out)
{
    float data[125];
    for (int z = 0; z < d; ++z)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                vmo_sync_row(x, y, z, w, h, d, w_tps, ui[((size_t)z * h + y) * w + x], data);
                float sum = 0;
                for (int t = 0; t < 25; ++t) {
                    const int *o = vmo_sync_taps[t];
                    float c = data[((o[0] + 2) * 5 + (o[1] + 2)) * 5 + (o[2] + 2)];
                    if (c != 0.0f) /* only stored entries take part, :264-275 */
                        sum = fmaf(c, p[((size_t)(z + o[0]) * h + (y + o[1])) * w + (x + o[2])], sum);
                }
                out[((size_t)z * h + y) * w + x] = sum;
            }
}

/* fast form of the same: rows from the per-state table + the stored diagonal (identical bits:
 * the off-diagonal entries depend on the state triple only, the diagonal is the stored one).
 * fmaf is exact either way; where the CPU has FMA instructions the body is also compiled for
 * them (a libm call per tap is ~10x slower) and picked at run time. */
static inline __attribute__((always_inline)) void apply_body(const sync_sys *S, const float *p, float *out)
{
    const int w = S->w, h = S->h, d = S->d;
#pragma omp parallel for num_threads(vmo_get_threads()) schedule(static)
    for (int z = 0; z < d; ++z)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const float *row = S->off[(vmo_sync_state(z, d) * 5 + vmo_sync_state(y, h)) * 5 + vmo_sync_state(x, w)];
                const size_t i = ((size_t)z * h + y) * w + x;
                float sum = 0;
                for (int t = 0; t < 25; ++t) {
                    float c = t == 12 ? S->diag[i] : row[t];
                    if (c != 0.0f) {
                        const int *o = vmo_sync_taps[t];
                        sum = __builtin_fmaf(c, p[((size_t)(z + o[0]) * h + (y + o[1])) * w + (x + o[2])], sum);
                    }
                }
                out[i] = sum;
            }
}

static inline __attribute__((always_inline)) void update_body(size_t N, float alpha, const float *p, const float *om, float *sol, float *r)
{
    const float nalpha = -alpha;
    for (size_t i = 0; i < N; ++i) {
        sol[i] = __builtin_fmaf(alpha, p[i], sol[i]);   /* cublasSaxpy(alpha, p, x) */
        r[i] = __builtin_fmaf(nalpha, om[i], r[i]);     /* cublasSaxpy(-alpha, omega, r) */
    }
}

#if defined(__x86_64__)
__attribute__((target("fma"))) static void apply_hw(const sync_sys *S, const float *p, float *out) { apply_body(S, p, out); }
__attribute__((target("fma"))) static void update_hw(size_t N, float alpha, const float *p, const float *om, float *sol, float *r) { update_body(N, alpha, p, om, sol, r); }
static int have_fma(void) { return __builtin_cpu_supports("fma"); }
#else
#define apply_hw apply_sw
#define update_hw update_sw
static int have_fma(void) { return 0; }
#endif
static void apply_sw(const sync_sys *S, const float *p, float *out) { apply_body(S, p, out); }
static void update_sw(size_t N, float alpha, const float *p, const float *om, float *sol, float *r) { update_body(N, alpha, p, om, sol, r); }

static void sync_apply_fast(const sync_sys *S, const float *p, float *out)
{
    if (have_fma()) apply_hw(S, p, out);
    else apply_sw(S, p, out);
}

/* CSyncThread::optimize_level, SyncThread.cpp:290-480.  x, y, z: the level's solution, in/out
 * (zero at the coarsest level, upsampled otherwise).  Returns the iterations of the loop. */
int vmo_sync_solve_level(int w, int h, int d, int w0, int h0, const int *cons6, int ncons,
                         float w_ui, float w_tps, float max_iter, float *x, float *y, float *z,
                         float *resid3)
{
    const size_t N = (size_t)w * h * d;
    float *ui = (float *)malloc(N * sizeof(float));
    float *r[3], *p[3], *om = (float *)malloc(N * sizeof(float));
    float *sol[3] = {x, y, z};
    for (int c = 0; c < 3; ++c) {
        r[c] = (float *)malloc(N * sizeof(float));
        p[c] = (float *)malloc(N * sizeof(float));
    }
    vmo_sync_ui(w, h, d, w0, h0, cons6, ncons, w_ui, ui, r[0], r[1], r[2]);
    sync_sys S;
    sync_sys_build(&S, w, h, d, w_tps, ui);
    vmo_sync_table(w, h, d, w_tps, &S.off[0][0]);
    const float tol = 1e-12f;
    float r0[3] = {0, 0, 0}, r1[3];
    for (int c = 0; c < 3; ++c) r1[c] = vmo_sync_dot(r[c], r[c], w, h, d);
    int k = 0;
    while (k <= max_iter) {
        k++;
        for (int c = 0; c < 3; ++c) {
            if (!(r1[c] > tol * tol)) continue;
            if (k == 1)
                memcpy(p[c], r[c], N * sizeof(float));
            else {
                float beta = r1[c] / r0[c];
                for (size_t i = 0; i < N; ++i) {
                    float t = beta * p[c][i];           /* cublasSscal */
                    p[c][i] = fmaf(1.0f, r[c][i], t);   /* cublasSaxpy(1, r, p) */
                }
            }
            sync_apply_fast(&S, p[c], om);
            float dot = vmo_sync_dot(p[c], om, w, h, d);
            float alpha = r1[c] / dot;
            if (have_fma()) update_hw(N, alpha, p[c], om, sol[c], r[c]);
            else update_sw(N, alpha, p[c], om, sol[c], r[c]);
            r0[c] = r1[c];
            r1[c] = vmo_sync_dot(r[c], r[c], w, h, d);
        }
    }
    if (resid3)
        for (int c = 0; c < 3; ++c) resid3[c] = r1[c];
    for (int c = 0; c < 3; ++c) { free(r[c]); free(p[c]); }
    free(om); free(ui); free(S.diag);
    return k;
}

/* Kernel_upsample, upsample.cu:343-353: normalised linear sampling at the destination pixel
 * centres, times `ratio` */
void vmo_sync_upsample(float *dst, int dw, int dh, const float *src, int sw, int sh, float ratio)
{
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            float px = (float)((x + 0.5) / (float)dw), py = (float)((y + 0.5) / (float)dh);
            dst[(size_t)y * dw + x] = vmo_tex2d(src, sw, sh, px * (float)sw, py * (float)sh) * ratio;
        }
}


/* ------------------------------------------------------------------------------------------ */
/* CSyncThread::update_result, SyncThread.cpp:482-521: (X ratio_x, Y ratio_y, Z, 0) per frame,
 * cv::resize(..., INTER_LINEAR) to full resolution.  OpenCV 3.0 (README.txt) is not here; what
 * follows restates its published generic linear resize for 32F: scale = 1 / (dst / src) in
 * double; source coordinate f = (float)((dx + 0.5) * scale - 0.5), s = floor(f), f -= s; columns
 * with s < 0 or s >= w - 1 collapse onto the border sample (f = 0); rows only clip their indices
 * (the weights stay); horizontal pass `S[s] * (1 - f) + S[s + 1] * f` (`S[s] * 1` from the first
 * column whose s + 1 leaves the image), then vertical pass `R0 * b0 + R1 * b1`, all in float.   */

void vmo_sync_result(const float *X, const float *Y, const float *Z, int w, int h, int w0, int h0, float *out4)
{
    const float ratio_x = (float)w0 / (float)w, ratio_y = (float)h0 / (float)h;
    const double sx = 1.0 / ((double)w0 / w), sy = 1.0 / ((double)h0 / h);
    int *xi = (int *)malloc((size_t)w0 * sizeof(int));
    float *xa = (float *)malloc((size_t)w0 * 2 * sizeof(float));
    int xmax = w0;
    for (int dx = 0; dx < w0; ++dx) {
        float fx = (float)((dx + 0.5) * sx - 0.5);
        int s = (int)floorf(fx);
        fx -= s;
        if (s < 0) { fx = 0; s = 0; }
        if (s + 1 >= w) {
            if (dx < xmax) xmax = dx;
            if (s >= w - 1) { fx = 0; s = w - 1; }
        }
        xi[dx] = s;
        xa[2 * dx] = 1.f - fx;
        xa[2 * dx + 1] = fx;
    }
    const float *src[3] = {X, Y, Z};
    float *rows = (float *)malloc((size_t)w0 * 3 * 2 * sizeof(float));
    for (int dy = 0; dy < h0; ++dy) {
        float fy = (float)((dy + 0.5) * sy - 0.5);
        int s = (int)floorf(fy);
        fy -= s;
        const float b0 = 1.f - fy, b1 = fy;
        for (int k = 0; k < 2; ++k) {
            int yy = s + k;
            yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
            for (int dx = 0; dx < w0; ++dx)
                for (int c = 0; c < 3; ++c) {
                    const float *S = src[c] + (size_t)yy * w;
                    float v0 = S[xi[dx]], v1 = dx < xmax ? S[xi[dx] + 1] : 0.0f;
                    if (c == 0) { v0 = v0 * ratio_x; v1 = v1 * ratio_x; }
                    if (c == 1) { v0 = v0 * ratio_y; v1 = v1 * ratio_y; }
                    rows[((size_t)k * w0 + dx) * 3 + c] = dx < xmax ? v0 * xa[2 * dx] + v1 * xa[2 * dx + 1] : v0 * 1.f;
                }
        }
        for (int dx = 0; dx < w0; ++dx) {
            float *o = out4 + 4 * ((size_t)dy * w0 + dx);
            for (int c = 0; c < 3; ++c)
                o[c] = rows[(size_t)dx * 3 + c] * b0 + rows[((size_t)w0 + dx) * 3 + c] * b1;
            o[3] = 0.0f;
        }
    }
    free(rows); free(xi); free(xa);
}

/* ------------------------------------------------------------------------------------------ */
/* kernel_render_resample_image0/1, render.cu:99-199                                            */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

typedef struct { int i0, i1, j0, j1; float a, b; } tex_pos;

static tex_pos tex_locate(int w, int h, float x, float y)
{
    tex_pos t;
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    t.a = xb - fi; t.b = yb - fj;
    fi = fminf(fmaxf(fi, -1.0f), (float)w);
    fj = fminf(fmaxf(fj, -1.0f), (float)h);
    t.i0 = clampi((int)fi, 0, w - 1); t.i1 = clampi((int)fi + 1, 0, w - 1);
    t.j0 = clampi((int)fj, 0, h - 1); t.j1 = clampi((int)fj + 1, 0, h - 1);
    return t;
}

/* tex2D, linear, clamp, on nc interleaved float channels (exact float weights, as vmo_tex2d) */
static void tex2d_n(const float *img, int nc, int w, int h, float x, float y, float *out)
{
    tex_pos t = tex_locate(w, h, x, y);
    const float a = t.a, b = t.b;
    for (int c = 0; c < nc; ++c) {
        float t00 = img[nc * ((size_t)t.j0 * w + t.i0) + c], t10 = img[nc * ((size_t)t.j0 * w + t.i1) + c];
        float t01 = img[nc * ((size_t)t.j1 * w + t.i0) + c], t11 = img[nc * ((size_t)t.j1 * w + t.i1) + c];
        out[c] = (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
    }
}

/* the same on an RGBA8 frame: the reference converts the frames to float4 0..255 before the
 * upload (pyramid.cu:93-96), which is exact */
static void tex2d_rgba8(const uint8_t *img, int w, int h, float x, float y, float *out4)
{
    tex_pos t = tex_locate(w, h, x, y);
    const float a = t.a, b = t.b;
    for (int c = 0; c < 4; ++c) {
        float t00 = img[4 * ((size_t)t.j0 * w + t.i0) + c], t10 = img[4 * ((size_t)t.j0 * w + t.i1) + c];
        float t01 = img[4 * ((size_t)t.j1 * w + t.i0) + c], t11 = img[4 * ((size_t)t.j1 * w + t.i1) + c];
        out4[c] = (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
    }
}

/* one side: sign = +1 for video0 (kernel 0: p = q + v, q.z -= v.z), -1 for video1 (kernel 1).
 * vec: w*h float4 (x, y, z, 0) of this frame; video: d frames of w*h RGBA8; flow: d frames of
 * w*h float2.  Writes the sampled colour (3 floats) per pixel. */
static void resample_side(int w, int h, int d, int frame, float sign, const float *vec,
                          const uint8_t *video, const float *flow, float *c_out)
{
    const size_t page = (size_t)w * h;
    const float alpha = 0.5f;
#pragma omp parallel for num_threads(vmo_get_threads()) schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float q[3] = {(float)x, (float)y, (float)frame}, p[3], v[4], t[4];
            p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            tex2d_n(vec, 4, w, h, (float)(p[0] + 0.5), (float)(p[1] + 0.5), v);
            v[2] = 0;
            for (int i = 0; i < 50; ++i) {
                p[0] = q[0] + sign * v[0];
                p[1] = q[1] + sign * v[1];
                tex2d_n(vec, 4, w, h, (float)(p[0] + 0.5), (float)(p[1] + 0.5), t);
                v[0] = alpha * t[0] + (1 - alpha) * v[0];
                v[1] = alpha * t[1] + (1 - alpha) * v[1];
                v[2] = 0;
            }
            tex2d_n(vec, 4, w, h, (float)(p[0] + 0.5), (float)(p[1] + 0.5), v);
            q[2] = q[2] - sign * v[2];
            float c[4];
            if (q[2] <= 0)
                tex2d_rgba8(video, w, h, (float)(q[0] + 0.5), (float)(q[1] + 0.5), c);
            else if (q[2] >= d - 1)
                tex2d_rgba8(video + (size_t)(d - 1) * page * 4, w, h, (float)(q[0] + 0.5), (float)(q[1] + 0.5), c);
            else {
                p[0] = q[0]; p[1] = q[1]; p[2] = floorf(q[2]);
                const float fa_z = q[2] - p[2];
                float f[2], g[2];
                int lz = clampi((int)(p[2] + 0.5), 0, d - 1);
                tex2d_n(flow + (size_t)lz * page * 2, 2, w, h, (float)(p[0] + 0.5), (float)(p[1] + 0.5), f);
                for (int i = 0; i < 50; ++i) {
                    p[0] = q[0] - f[0] * fa_z;
                    p[1] = q[1] - f[1] * fa_z;
                    p[2] = q[2] - 1.0f * fa_z;
                    lz = clampi((int)(p[2] + 0.5), 0, d - 1);
                    tex2d_n(flow + (size_t)lz * page * 2, 2, w, h, (float)(p[0] + 0.5), (float)(p[1] + 0.5), g);
                    f[0] = alpha * g[0] + (1 - alpha) * f[0];
                    f[1] = alpha * g[1] + (1 - alpha) * f[1];
                }
                const int l0 = clampi((int)(p[2] + 0.5), 0, d - 1), l1 = clampi((int)(p[2] + 0.5 + 1), 0, d - 1);
                float c0[4], c1[4];
                tex2d_rgba8(video + (size_t)l0 * page * 4, w, h, (float)(p[0] + 0.5), (float)(p[1] + 0.5), c0);
                tex2d_rgba8(video + (size_t)l1 * page * 4, w, h, (float)(p[0] + 0.5 + f[0]), (float)(p[1] + 0.5 + f[1]), c1);
                for (int k = 0; k < 3; ++k) c[k] = c0[k] * (1 - fa_z) + c1[k] * fa_z;
            }
            for (int k = 0; k < 3; ++k) c_out[3 * ((size_t)y * w + x) + k] = c[k];
        }
}

/* render_resample_image, render.cu:203-246: out = 0; side 0 if fa < 1, side 1 if fa > 0; each adds
 * `(c + 0.5) * weight` to the byte already there and truncates (so a blend truncates twice) */
void vmo_render_resample(uint8_t *out, int w, int h, int d, float fa, int frame, const float *vec,
                         const uint8_t *video0, const uint8_t *video1, const float *forw0, const float *forw1)
{
    const size_t page = (size_t)w * h;
    float *c = (float *)malloc(page * 3 * sizeof(float));
    memset(out, 0, page * 3);
    if (fa < 1) {
        resample_side(w, h, d, frame, 1.0f, vec, video0, forw0, c);
        for (size_t i = 0; i < page * 3; ++i) out[i] = (uint8_t)(out[i] + (c[i] + 0.5) * (1 - fa));
    }
    if (fa > 0) {
        resample_side(w, h, d, frame, -1.0f, vec, video1, forw1, c);
        for (size_t i = 0; i < page * 3; ++i) out[i] = (uint8_t)(out[i] + (c[i] + 0.5) * fa);
    }
    free(c);
}
