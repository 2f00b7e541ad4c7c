/*
 * vm_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 * See vm_oracle.h for scope, parity status and numerics conventions.
 * All file:line citations are relative to /root/reference.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -fopenmp
 */
#include "vm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* tile geometry of kernel_optimize_level, morph.cu:594-598, 1291-1292 */
#define OPT_BW 32
#define OPT_BH 8
#define SPACING 5
#define TILE_W (OPT_BW * 2)              /* 64 */
#define TILE_H (OPT_BH * 2)              /* 16 */
#define PITCH_X (TILE_W + SPACING)       /* 69 */
#define PITCH_Y (TILE_H + SPACING)       /* 21 */

static int g_threads = 1;
void vmo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int  vmo_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------------- */
/* calc_border, morph.cu:39-81 (the readable #if 0 branch, same function)     */
int vmo_calc_border(int p, int dim)
{
    if (p < 2) return p;
    if (p == dim - 2) return 3;
    if (p == dim - 1) return 4;
    return 2;
}

/* ssim(), morph.cu:85-118 */
float vmo_ssim(float mx, float my, float vx, float vy, float cross,
               float counter, float clamp)
{
    if (counter <= 1)
        return 0;
    const float c2 = (float)((255 * 0.03) * (255 * 0.03)); /* pow2(255*0.03) */
    mx /= counter;
    my /= counter;
    vx = (vx - counter * mx * mx) / counter;
    vy = (vy - counter * my * my) / counter;
    vx = fmaxf(0.0f, vx);
    vy = fmaxf(0.0f, vy);
    cross = (cross - counter * mx * my) / counter;
    const float c3 = 29.26125f;
    float sx = sqrtf(vx), sy = sqrtf(vy);
    float c = (2 * sx * sy + c2) / (vx + vy + c2);
    float s = (fabsf(cross) + c3) / (sx * sy + c3);
    float value = c * s;
    return fmaxf(fminf(1.0f, value), clamp);
}

/* CUDA tex2D, linear filter, clamp addressing, unnormalised coordinates
 * (set up at morph.cu:316-322): texel centres sit at i+0.5.  Exact float
 * weights by default (the 1.8 fixed-point weights of the
 * hardware are not emulated).
 *
 * vmo_set_tex_filter(1 | 2): DIAGNOSTIC -- the weights quantised as CUDA's
 * linear filter does, "9-bit fixed point format with 8 bits of fractional
 * value (so 1.0 is exactly represented)" (CUDA C Programming Guide, Texture
 * Fetching / Linear Filtering).  The guide does not state the rounding rule:
 * 1 = round to nearest, floor(256 a + 0.5) / 256 (the only reading under which
 * the fraction can reach the 1.0 the format represents), 2 = truncation.
 * Mirrors VM_MATH_REF_TEX8 / _TRUNC of the HIP path (vm_morph_common.h:
 * tex8_weight) bit for bit; used to measure how far the reference BINARY's
 * arithmetic sits from the exact-weight runs. */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int g_tex_filter = 0;
void vmo_set_tex_filter(int mode) { g_tex_filter = (mode == 1 || mode == 2) ? mode : 0; }
int  vmo_get_tex_filter(void) { return g_tex_filter; }
static inline float texw(float a)
{
    if (g_tex_filter == 1) return floorf(a * 256.0f + 0.5f) * 0.00390625f;
    if (g_tex_filter == 2) return floorf(a * 256.0f) * 0.00390625f;
    return a;
}

float vmo_tex2d(const float *img, int w, int h, float x, float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = texw(xb - fi), b = texw(yb - fj);
    fi = fminf(fmaxf(fi, -1.0f), (float)w);
    fj = fminf(fmaxf(fj, -1.0f), (float)h);
    int i0 = clampi((int)fi, 0, w - 1), i1 = clampi((int)fi + 1, 0, w - 1);
    int j0 = clampi((int)fj, 0, h - 1), j1 = clampi((int)fj + 1, 0, h - 1);
    float t00 = img[j0 * w + i0], t10 = img[j0 * w + i1];
    float t01 = img[j1 * w + i0], t11 = img[j1 * w + i1];
    return (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
}

/* same for interleaved 2-channel data */
void vmo_tex2d_f2(const float *img, int w, int h, float x, float y, float *out2)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = texw(xb - fi), b = texw(yb - fj);
    fi = fminf(fmaxf(fi, -1.0f), (float)w);
    fj = fminf(fmaxf(fj, -1.0f), (float)h);
    int i0 = clampi((int)fi, 0, w - 1), i1 = clampi((int)fi + 1, 0, w - 1);
    int j0 = clampi((int)fj, 0, h - 1), j1 = clampi((int)fj + 1, 0, h - 1);
    for (int c = 0; c < 2; ++c) {
        float t00 = img[2 * (j0 * w + i0) + c], t10 = img[2 * (j0 * w + i1) + c];
        float t01 = img[2 * (j1 * w + i0) + c], t11 = img[2 * (j1 * w + i1) + c];
        out2[c] = (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
    }
}

/* ------------------------------------------------------------------------- */
/* Stencil tables                                                             */

/* is the neighbour at signed offset `off` inside the image for border class B
 * (classes of calc_border: 0,1 = distance to the low edge, 3,4 = to the high
 * edge, 2 = at least two pixels from both) */
static int class_inside(int B, int off)
{
    switch (B) {
    case 0: return off >= 0;
    case 1: return off >= -1;
    case 3: return off <= 1;
    case 4: return off <= 0;
    default: return 1;
    }
}

/* calc_nb_io_stencil, stencils.cpp:10-71: mask[By][Bx][y][x] = 1 iff the
 * neighbour (x-2, y-2) exists */
void vmo_io_stencil(int *out)
{
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j)
            for (int y = 0; y < 5; ++y)
                for (int x = 0; x < 5; ++x)
                    out[((i * 5 + j) * 5 + y) * 5 + x] =
                        class_inside(i, y - 2) && class_inside(j, x - 2);
}

/* calc_nb_improvmask_check_stencil, stencils.cpp:90-126:
 * mask[oy][ox][by][bx] = bits of the block at (bx-1,by-1) relative to the
 * pixel's own 5x5 block that fall inside the pixel's 5x5 window, the pixel
 * sitting at (ox,oy) inside its block */
void vmo_improvmask_stencil(uint32_t *out)
{
    memset(out, 0, 225 * sizeof(uint32_t));
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j)
            for (int y = 0; y < 5; ++y)
                for (int x = 0; x < 5; ++x) {
                    int ax = j + 5 + (x - 2), ay = i + 5 + (y - 2);
                    int bx = ax / 5, by = ay / 5;
                    int rx = ax - bx * 5, ry = ay - by * 5;
                    out[((i * 5 + j) * 3 + by) * 3 + bx] |=
                        (1u << (rx + ry * 5)) & ((1u << 25) - 1);
                }
}

/* calc_tps_stencil, stencils.cpp:156-261.  For every border class (m,n) and
 * every placement of the discrete operators dxx, dyy (weight 2) and dxy
 * (weight 4) that contains the centre pixel and whose non-zero rows/columns
 * all lie inside the image, accumulate K * K[centre] * weight. */
void vmo_tps_stencil(float *out)
{
    static const float dxx[3][3] = {{0, 0, 0}, {1, -2, 1}, {0, 0, 0}};
    static const float dxy[3][3] = {{0, -1, 1}, {0, 1, -1}, {0, 0, 0}};
    static const float dyy[3][3] = {{0, 1, 0}, {0, -2, 0}, {0, 1, 0}};
    const float (*K[3])[3] = {dxx, dxy, dyy};
    const float wgt[3] = {2, 4, 2};
    memset(out, 0, 625 * sizeof(float));
    for (int m = 0; m < 5; ++m)
        for (int n = 0; n < 5; ++n) {
            float *t = out + (m * 5 + n) * 25;
            for (int k = 0; k < 3; ++k)
                for (int i = -1; i <= 1; ++i)
                    for (int j = -1; j <= 1; ++j) {
                        /* operator centre at p + (j,i); p is its element [1-i][1-j] */
                        int ok = 1;
                        for (int r = 0; r < 3 && ok; ++r)
                            for (int c = 0; c < 3 && ok; ++c)
                                if (K[k][r][c] != 0 &&
                                    !(class_inside(m, i + r - 1) && class_inside(n, j + c - 1)))
                                    ok = 0;
                        if (!ok)
                            continue;
                        float kc = K[k][1 - i][1 - j] * wgt[k];
                        for (int u = 0; u < 3; ++u)
                            for (int v = 0; v < 3; ++v)
                                t[(2 + i - 1 + u) * 5 + (2 + j - 1 + v)] += K[k][u][v] * kc;
                    }
        }
}

/* The same operator as assembled row by row into the dense matrix of
 * Morph::cpu_optimize_level, morph.cu:439-469, for a 5x5 image (whose pixel
 * (n,m) has border class (m,n)), divided by w_tps.  Second, independent
 * statement of the table above; tests assert the two agree. */
void vmo_tps_rows_from_dense(float *out)
{
    const int w = 5, h = 5;
    memset(out, 0, 625 * sizeof(float));
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float row[25];
            memset(row, 0, sizeof(row));
            int i = y * w + x;
#define A_(col) row[(col)]
            /* dxx */
            if (x > 1) { A_(i - 2) += 1.0f * 2.0f; A_(i - 1) += -2.0f * 2.0f; A_(i) += 1.0f * 2.0f; }
            if (x > 0 && x < w - 1) { A_(i - 1) += -2.0f * 2.0f; A_(i) += 4.0f * 2.0f; A_(i + 1) += -2.0f * 2.0f; }
            if (x < w - 2) { A_(i) += 1.0f * 2.0f; A_(i + 1) += -2.0f * 2.0f; A_(i + 2) += 1.0f * 2.0f; }
            /* dyy */
            if (y > 1) { A_(i - 2 * w) += 1.0f * 2.0f; A_(i - w) += -2.0f * 2.0f; A_(i) += 1.0f * 2.0f; }
            if (y > 0 && y < h - 1) { A_(i - w) += -2.0f * 2.0f; A_(i) += 4.0f * 2.0f; A_(i + w) += -2.0f * 2.0f; }
            if (y < h - 2) { A_(i) += 1.0f * 2.0f; A_(i + w) += -2.0f * 2.0f; A_(i + 2 * w) += 1.0f * 2.0f; }
            /* dxy */
            if (x > 0 && y > 0) { A_(i - w - 1) += 2.0f * 2.0f; A_(i - w) += -2.0f * 2.0f; A_(i - 1) += -2.0f * 2.0f; A_(i) += 2.0f * 2.0f; }
            if (x < w - 1 && y > 0) { A_(i - w) += -2.0f * 2.0f; A_(i - w + 1) += 2.0f * 2.0f; A_(i) += 2.0f * 2.0f; A_(i + 1) += -2.0f * 2.0f; }
            if (x > 0 && y < h - 1) { A_(i - 1) += -2.0f * 2.0f; A_(i) += 2.0f * 2.0f; A_(i + w - 1) += 2.0f * 2.0f; A_(i + w) += -2.0f * 2.0f; }
            if (x < w - 1 && y < h - 1) { A_(i) += 2.0f * 2.0f; A_(i + 1) += -2.0f * 2.0f; A_(i + w) += -2.0f * 2.0f; A_(i + w + 1) += 2.0f * 2.0f; }
#undef A_
            float *t = out + (y * 5 + x) * 25;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    int qx = x + dx, qy = y + dy;
                    if (qx >= 0 && qx < w && qy >= 0 && qy < h)
                        t[(dy + 2) * 5 + (dx + 2)] = row[qy * w + qx];
                }
        }
}

/* tables shared by the hot-path functions below */
static float    T_tps[5][5][5][5];
static int      T_io[5][5][5][5];
static uint32_t T_imp[5][5][3][3];
static int      T_ready = 0;

static void tables_init(void)
{
    if (T_ready)
        return;
#pragma omp critical(vmo_tables)
    {
        if (!T_ready) {
            vmo_tps_stencil(&T_tps[0][0][0][0]);
            vmo_io_stencil(&T_io[0][0][0][0]);
            vmo_improvmask_stencil(&T_imp[0][0][0][0]);
            T_ready = 1;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Level objects: PyramidLevel ctor, pyramid.cu:531-543 (rows kept tight)     */

vmo_level *vmo_level_create(int w, int h)
{
    vmo_level *l = (vmo_level *)calloc(1, sizeof(vmo_level));
    size_t n = (size_t)w * h;
    l->w = w;
    l->h = h;
    l->inv_wh = 1.0f / (w * h);
    l->imp_rs = (w + 4) / 5 + 2;
    l->imp_rows = (h + 4) / 5 + 2;
    l->img0 = (float *)calloc(n, sizeof(float));
    l->img1 = (float *)calloc(n, sizeof(float));
    l->v = (float *)calloc(2 * n, sizeof(float));
    l->luma = (float *)calloc(2 * n, sizeof(float));
    l->mean = (float *)calloc(2 * n, sizeof(float));
    l->var = (float *)calloc(2 * n, sizeof(float));
    l->cross = (float *)calloc(n, sizeof(float));
    l->value = (float *)calloc(n, sizeof(float));
    l->counter = (float *)calloc(n, sizeof(float));
    l->tps_axy = (float *)calloc(n, sizeof(float));
    l->tps_b = (float *)calloc(2 * n, sizeof(float));
    l->ui_axy = (float *)calloc(n, sizeof(float));
    l->ui_b = (float *)calloc(2 * n, sizeof(float));
    l->impmask = (uint32_t *)calloc((size_t)l->imp_rs * l->imp_rows, sizeof(uint32_t));
    l->temp_ref = (float *)calloc(2 * n, sizeof(float));
    l->temp_mask = (float *)calloc(n, sizeof(float));
    l->factor_d = 1.0f; /* pyramid.cu:541 */
    l->flag = 0;
    return l;
}

void vmo_level_destroy(vmo_level *l)
{
    if (!l)
        return;
    free(l->img0); free(l->img1); free(l->v); free(l->luma); free(l->mean);
    free(l->var); free(l->cross); free(l->value); free(l->counter);
    free(l->tps_axy); free(l->tps_b); free(l->ui_axy); free(l->ui_b);
    free(l->impmask);
    free(l->temp_ref); free(l->temp_mask);
    free(l);
}

/* the page's temporal state: `flag` of kernel_optimize_level, factor_d of its level */
void vmo_level_set_temporal(vmo_level *l, int flag, float factor_d)
{
    l->flag = flag;
    l->factor_d = factor_d;
}

void *vmo_level_field(vmo_level *l, int f)
{
    switch (f) {
    case VMO_F_IMG0: return l->img0;
    case VMO_F_IMG1: return l->img1;
    case VMO_F_V: return l->v;
    case VMO_F_LUMA: return l->luma;
    case VMO_F_MEAN: return l->mean;
    case VMO_F_VAR: return l->var;
    case VMO_F_CROSS: return l->cross;
    case VMO_F_VALUE: return l->value;
    case VMO_F_COUNTER: return l->counter;
    case VMO_F_TPS_AXY: return l->tps_axy;
    case VMO_F_TPS_B: return l->tps_b;
    case VMO_F_UI_AXY: return l->ui_axy;
    case VMO_F_UI_B: return l->ui_b;
    case VMO_F_IMPMASK: return l->impmask;
    case VMO_F_TEMP_REF: return l->temp_ref;
    case VMO_F_TEMP_MASK: return l->temp_mask;
    }
    return 0;
}

static inline int contains(const vmo_level *l, int x, int y)
{
    return x >= 0 && x < l->w && y >= 0 && y < l->h;
}

/* ------------------------------------------------------------------------- */
/* kernel_initialize_level, morph.cu:173-244; init_improving_mask :246-260;   */
/* zero-fill of the per-level state :300-314                                  */
void vmo_init_level(vmo_level *l, float ssim_clamp)
{
    tables_init();
    const int w = l->w, h = l->h;
    size_t n = (size_t)w * h;
    memset(l->ui_axy, 0, n * sizeof(float));
    memset(l->ui_b, 0, 2 * n * sizeof(float));

#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int By = vmo_calc_border(y, h), Bx = vmo_calc_border(x, w);
            int counter = 0;
            float mx = 0, my = 0, vx = 0, vy = 0, cross = 0, bx = 0, by = 0;
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j < 5; ++j) {
                    if (T_io[By][Bx][i][j] == 0)
                        continue;
                    int qx = x + j - 2, qy = y + i - 2;
                    int nb = qy * w + qx;
                    float nvx = l->v[2 * nb], nvy = l->v[2 * nb + 1];
                    float tx = (float)qx + 0.5f, ty = (float)qy + 0.5f;
                    float lx = vmo_tex2d(l->img0, w, h, tx - nvx, ty - nvy);
                    float ly = vmo_tex2d(l->img1, w, h, tx + nvx, ty + nvy);
                    float c = T_tps[By][Bx][i][j];
                    bx += nvx * c;
                    by += nvy * c;
                    counter += 1;
                    mx += lx;
                    my += ly;
                    vx += lx * lx;
                    vy += ly * ly;
                    cross += lx * ly;
                    if (i == 2 && j == 2) {
                        l->luma[2 * nb] = lx;
                        l->luma[2 * nb + 1] = ly;
                    }
                }
            int idx = y * w + x;
            l->counter[idx] = (float)counter;
            l->mean[2 * idx] = mx;
            l->mean[2 * idx + 1] = my;
            l->var[2 * idx] = vx;
            l->var[2 * idx + 1] = vy;
            l->cross[idx] = cross;
            l->value[idx] = vmo_ssim(mx, my, vx, vy, cross, (float)counter, ssim_clamp);
            l->tps_axy[idx] = T_tps[By][Bx][2][2] / 2;
            l->tps_b[2 * idx] = bx;
            l->tps_b[2 * idx + 1] = by;
        }

    for (int by = 0; by < l->imp_rows; ++by)
        for (int bx = 0; bx < l->imp_rs; ++bx)
            l->impmask[by * l->imp_rs + bx] =
                (bx == 0 || by == 0 || bx == l->imp_rs - 1 || by == l->imp_rows - 1)
                    ? 0u : ((1u << 25) - 1);
}

/* UI constraint linearisation, morph.cu:345-388 (one page, depth 1) */
void vmo_splat_constraints(vmo_level *l, int w0, int h0,
                           const vmo_constraint *c, int n)
{
    for (int k = 0; k < n; ++k) {
        float x0 = (float)((c[k].lx + 0.5) / w0 * l->w - 0.5f);
        float y0 = (float)((c[k].ly + 0.5) / h0 * l->h - 0.5f);
        float x1 = (float)((c[k].rx + 0.5) / w0 * l->w - 0.5f);
        float y1 = (float)((c[k].ry + 0.5) / h0 * l->h - 0.5f);
        float weight = c[k].weight;
        float con_x = (x0 + x1) / 2.0f, con_y = (y0 + y1) / 2.0f;
        float vx = (x1 - x0) / 2.0f, vy = (y1 - y0) / 2.0f;
        for (int y = (int)floor(con_y); y <= (int)ceil(con_y); ++y)
            for (int x = (int)floor(con_x); x <= (int)ceil(con_x); ++x)
                if (contains(l, x, y)) {
                    int idx = y * l->w + x;
                    float bw = (1 - fabsf(y - con_y)) * (1 - fabsf(x - con_x)) * weight;
                    l->ui_axy[idx] += bw;
                    l->ui_b[2 * idx] += 2 * bw * (l->v[2 * idx] - vx);
                    l->ui_b[2 * idx + 1] += 2 * bw * (l->v[2 * idx + 1] - vy);
                }
    }
}

/* ------------------------------------------------------------------------- */
/* Device helpers of kernel_optimize_level                                    */

/* get_improve_mask_idx, morph.cu:621-646 */
static int improve_mask_idx(const vmo_level *l, int px, int py)
{
    int bx = px / 5, by = py / 5, ox = px % 5, oy = py % 5;
    int begi = oy >= 2 ? 1 : 0, begj = ox >= 2 ? 1 : 0;
    int idx = (by + 1) * l->imp_rs + (bx + 1);
    for (int i = begi; i < begi + 2; ++i)
        for (int j = begj; j < begj + 2; ++j) {
            int d = idx + (i - 1) * l->imp_rs + (j - 1); /* stencils.cpp:121-126 */
            if (l->impmask[d] & T_imp[oy][ox][i][j])
                return idx;
        }
    return -1;
}

/* pixel_on_border, morph.cu:648-667 -- including the BCOND_CORNER expression
 * exactly as written (&& binds tighter than ||, so the right-hand corners
 * never match) */
static int pixel_on_border(const vmo_level *l, const vmo_params *P, int px, int py)
{
    int W = l->w, H = l->h;
    switch (P->bcond) {
    case VMO_BCOND_NONE:
        break;
    case VMO_BCOND_CORNER:
        if ((px == 0 && py == 0) || (px == 0 && py == H - 1) ||
            (px == W - 1 && py == 0 && px == W - 1 && py == H - 1))
            return 1;
        break;
    case VMO_BCOND_BORDER:
        if (px == 0 || py == 0 || px == W - 1 || py == H - 1)
            return 1;
        break;
    }
    return 0;
}

typedef struct { double visits, cand, commits, evals; } sweep_stats;

/* ssim_change, morph.cu:671-728 */
static float ssim_change(const vmo_level *l, const vmo_params *P, int px, int py,
                         float vx, float vy, float olx, float oly)
{
    const int w = l->w, h = l->h;
    float lx = vmo_tex2d(l->img0, w, h, px - vx + 0.5f, py - vy + 0.5f);
    float ly = vmo_tex2d(l->img1, w, h, px + vx + 0.5f, py + vy + 0.5f);
    float change = 0;
    float dmx = lx - olx, dmy = ly - oly;
    float dvx = lx * lx - olx * olx, dvy = ly * ly - oly * oly;
    float dcross = lx * ly - olx * oly;
    int need_counter = px < 4 || px >= w - 4 || py < 4 || py >= h - 4;
    int By = vmo_calc_border(py, h), Bx = vmo_calc_border(px, w);
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) {
            if (T_io[By][Bx][i][j] == 0)
                continue;
            int nb = (py + i - 2) * w + (px + j - 2);
            float counter = need_counter ? l->counter[nb] : 25;
            float mx = l->mean[2 * nb] + dmx, my = l->mean[2 * nb + 1] + dmy;
            float qx = l->var[2 * nb] + dvx, qy = l->var[2 * nb + 1] + dvy;
            float cr = l->cross[nb] + dcross;
            float ns = vmo_ssim(mx, my, qx, qy, cr, counter, P->ssim_clamp);
            change += l->value[nb] - ns;
        }
    return change;
}

/* energy_change, morph.cu:730-761.  flag == false (a single frame pair, the middle page of a
 * video level): v_temp = 0 and lvl.temp.mask = 0, the temporal term multiplies out to +0 */
static float energy_change(const vmo_level *l, const vmo_params *P, int px, int py,
                           float vx, float vy, float olx, float oly,
                           float dx, float dy, sweep_stats *st)
{
    st->evals += 1;
    float v_ssim = ssim_change(l, P, px, py, vx + dx, vy + dy, olx, oly);
    int idx = py * l->w + px;
    float v_tps = l->tps_axy[idx] * (dx * dx + dy * dy);
    v_tps += l->tps_b[2 * idx] * dx;
    v_tps += l->tps_b[2 * idx + 1] * dy;
    float v_ui = l->ui_axy[idx] * (dx * dx + dy * dy);
    v_ui += l->ui_b[2 * idx] * dx;
    v_ui += l->ui_b[2 * idx + 1] * dy;
    float v_temp = 0.0f;
    if (l->flag) {
        v_temp += fabsf(vx + dx - l->temp_ref[2 * idx]) - fabsf(vx - l->temp_ref[2 * idx]);
        v_temp += fabsf(vy + dy - l->temp_ref[2 * idx + 1]) - fabsf(vy - l->temp_ref[2 * idx + 1]);
    }
    const float mask = l->flag ? l->temp_mask[idx] : 0.0f;
    return (P->w_ui * v_ui + P->w_ssim * v_ssim + P->w_temp * v_temp * mask * l->factor_d) * l->inv_wh
           + P->w_tps * v_tps;
}

/* fover_calc_vtx, morph.cu:782-792 -- note `p - off` as written */
static void fover_vtx(const vmo_level *l, int px, int py, int X, int Y, int SIGN,
                      float vx, float vy, float *ox, float *oy)
{
    if (contains(l, px + X, py + Y)) {
        int nb = (py + Y) * l->w + (px + X);
        vx = SIGN * l->v[2 * nb];
        vy = SIGN * l->v[2 * nb + 1];
    }
    *ox = vx + (float)(px - X);
    *oy = vy + (float)(py - Y);
}

/* fover_update_isec_min, morph.cu:794-831 */
static void fover_isec(float cx, float cy, float gx, float gy,
                       float e0x, float e0y, float e1x, float e1y, float *t_min)
{
    float dex = e1x - e0x, dey = e1y - e0y;
    float dcx = cx - e0x, dcy = cy - e0y;
    float d = dey * gx - dex * gy;
    float td = -1;
    float ud = gx * dcy - gy * dcx;
    int sign = signbit(d) ? 1 : 0;
    if (sign) {
        ud = -ud;
        d = -d;
    }
    if (ud >= 0 && ud <= d) {
        td = dex * dcy - dey * dcx;
        td *= (float)(-sign * 2 + 1);
        if (td >= 0 && td < *t_min * d)
            *t_min = td / d;
    }
}

/* fover_calc_isec_min, morph.cu:833-870 */
static void fover_ring(const vmo_level *l, int px, int py, int SIGN,
                       float vx, float vy, float gx, float gy, float *t_min)
{
    static const int ring[8][2] = {{-1, -1}, {0, -1}, {1, -1}, {1, 0},
                                   {1, 1}, {0, 1}, {-1, 1}, {-1, 0}};
    float ex[9], ey[9];
    for (int k = 0; k < 8; ++k)
        fover_vtx(l, px, py, ring[k][0], ring[k][1], SIGN, vx, vy, &ex[k], &ey[k]);
    ex[8] = ex[0];
    ey[8] = ey[0];
    float cx = (float)px + vx, cy = (float)py + vy;
    for (int k = 0; k < 8; ++k)
        fover_isec(cx, cy, gx, gy, ex[k], ey[k], ex[k + 1], ey[k + 1], t_min);
}

/* prevent_foldover, morph.cu:872-883 */
static float prevent_foldover(const vmo_level *l, const vmo_params *P, int px, int py,
                              float vx, float vy, float gx, float gy)
{
    float t_min = 10;
    fover_ring(l, px, py, -1, -vx, -vy, -gx, -gy, &t_min);
    fover_ring(l, px, py, 1, vx, vy, gx, gy, &t_min);
    return fmaxf(t_min - P->eps, 0.0f);
}

/* golden_section_search, morph.cu:885-947 */
static void golden_section(const vmo_level *l, const vmo_params *P, int px, int py,
                           float a, float c, float vx, float vy, float gx, float gy,
                           float olx, float oly, float *fmin, float *tmin, sweep_stats *st)
{
    const float R = 0.618033989f, C = 1.0f - R;
    float b = a * R + c * C, x = b * R + c * C;
    float fb = energy_change(l, P, px, py, vx, vy, olx, oly, gx * b, gy * b, st);
    float fx = energy_change(l, P, px, py, vx, vy, olx, oly, gx * x, gy * x, st);
    while (c - a > P->eps) {
        if (fx < fb) {
            a = b;
            b = x;
            x = b * R + c * C;
        } else {
            c = x;
            x = b * R + a * C;
        }
        float f = energy_change(l, P, px, py, vx, vy, olx, oly, gx * x, gy * x, st);
        if (fx < fb) {
            fb = fx;
            fx = f;
        } else {
            float t = b; b = x; x = t;
            fx = fb;
            fb = f;
        }
    }
    if (fx < fb) {
        *tmin = x;
        *fmin = fx;
    } else {
        *tmin = b;
        *fmin = fb;
    }
}

typedef struct {
    int px, py;
    int ok;           /* commit */
    int imp_idx;      /* -1: not a candidate */
    float nvx, nvy;   /* new v */
    float sx, sy;     /* accepted step (grad * tmin) */
    float olx, oly;
} decision;

/* optimize_pixel, morph.cu:1030-1083 */
static void optimize_pixel(const vmo_level *l, const vmo_params *P, int px, int py,
                           decision *d, sweep_stats *st)
{
    d->px = px;
    d->py = py;
    d->ok = 0;
    d->imp_idx = -1;
    if (!contains(l, px, py))
        return;
    st->visits += 1;
    int idx = py * l->w + px;
    float vx = l->v[2 * idx], vy = l->v[2 * idx + 1];
    float olx = l->luma[2 * idx], oly = l->luma[2 * idx + 1];
    d->olx = olx;
    d->oly = oly;
    d->imp_idx = improve_mask_idx(l, px, py);
    if (d->imp_idx < 0)
        return;
    if (pixel_on_border(l, P, px, py))
        return;
    st->cand += 1;
    /* compute_gradient, morph.cu:763-778 */
    float gx = energy_change(l, P, px, py, vx, vy, olx, oly, P->eps, 0, st) -
               energy_change(l, P, px, py, vx, vy, olx, oly, -P->eps, 0, st);
    float gy = energy_change(l, P, px, py, vx, vy, olx, oly, 0, P->eps, st) -
               energy_change(l, P, px, py, vx, vy, olx, oly, 0, -P->eps, st);
    gx = -gx;
    gy = -gy;
    float ng = sqrtf(gx * gx + gy * gy);
    if (ng != 0) {
        gx /= ng;
        gy /= ng;
        float t = prevent_foldover(l, P, px, py, vx, vy, gx, gy);
        float tmin, fmin;
        golden_section(l, P, px, py, 0, t, vx, vy, gx, gy, olx, oly, &fmin, &tmin, st);
        if (fmin < 0) {
            gx *= tmin;
            gy *= tmin;
            d->sx = gx;
            d->sy = gy;
            d->nvx = vx + gx;
            d->nvy = vy + gy;
            d->ok = 1;
        }
    }
}

float vmo_dbg_foldover(const vmo_level *l, const vmo_params *P, int px, int py, float gx, float gy)
{
    int idx = py * l->w + px;
    return prevent_foldover(l, P, px, py, l->v[2 * idx], l->v[2 * idx + 1], gx, gy);
}

float vmo_dbg_energy_change(const vmo_level *l, const vmo_params *P, int px, int py, float dx, float dy)
{
    tables_init();
    sweep_stats st = {0, 0, 0, 0};
    int idx = py * l->w + px;
    return energy_change(l, P, px, py, l->v[2 * idx], l->v[2 * idx + 1],
                         l->luma[2 * idx], l->luma[2 * idx + 1], dx, dy, &st);
}

/* One thread block of kernel_optimize_level, morph.cu:1281-1345: the 64x16
 * tile whose first pixel is (ox,oy); 4 phases, each Jacobi on the pre-phase
 * state, commits applied in row-major order of the committing pixels
 * (commit_pixel_motion :990-1026, ssim_update :951-988), then the SSIM value
 * of every tile+halo cell recomputed (UpdateSSIM :1258-1279). */
/* Diagnostic: the order in which the commits of a phase are applied.  The reference leaves it to
 * float atomics (morph.cu:951-1015).  0 = row-major over the committing pixels (the order this
 * oracle fixes), bit 0 = that sequence reversed, bit 1 = column-major (tx outer, ty inner). */
static int g_commit_order = 0;
void vmo_set_commit_order(int order) { g_commit_order = order & 3; }

static int optimize_tile(vmo_level *l, const vmo_params *P, int ox, int oy, sweep_stats *st)
{
    const int w = l->w, h = l->h;
    int improving = 0;
    decision dec[OPT_BW * OPT_BH];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            int nd = 0;
            for (int ty = 0; ty < OPT_BH; ++ty)
                for (int tx = 0; tx < OPT_BW; ++tx)
                    optimize_pixel(l, P, ox + tx * 2 + j, oy + ty * 2 + i, &dec[nd++], st);
            /* __syncthreads(); commits (row-major: ty outer, tx inner) */
            for (int kk = 0; kk < nd; ++kk) {
                int ks = (g_commit_order & 1) ? nd - 1 - kk : kk;
                if (g_commit_order & 2) /* column-major: position ks of the sequence tx outer, ty inner */
                    ks = (ks % OPT_BH) * OPT_BW + ks / OPT_BH;
                decision *d = &dec[ks];
                if (d->ok) {
                    int px = d->px, py = d->py, idx = py * w + px;
                    float lx = vmo_tex2d(l->img0, w, h, px - d->nvx + 0.5f, py - d->nvy + 0.5f);
                    float ly = vmo_tex2d(l->img1, w, h, px + d->nvx + 0.5f, py + d->nvy + 0.5f);
                    l->luma[2 * idx] = lx;
                    l->luma[2 * idx + 1] = ly;
                    float dmx = lx - d->olx, dmy = ly - d->oly;
                    float dvx = lx * lx - d->olx * d->olx, dvy = ly * ly - d->oly * d->oly;
                    float dcross = lx * ly - d->olx * d->oly;
                    int By = vmo_calc_border(py, h), Bx = vmo_calc_border(px, w);
                    for (int a = 0; a < 5; ++a)
                        for (int b = 0; b < 5; ++b) {
                            if (!T_io[By][Bx][a][b])
                                continue;
                            int nb = (py + a - 2) * w + (px + b - 2);
                            l->mean[2 * nb] += dmx;
                            l->mean[2 * nb + 1] += dmy;
                            l->var[2 * nb] += dvx;
                            l->var[2 * nb + 1] += dvy;
                            l->cross[nb] += dcross;
                            float c = T_tps[By][Bx][a][b];
                            l->tps_b[2 * nb] += d->sx * c;
                            l->tps_b[2 * nb + 1] += d->sy * c;
                        }
                    l->ui_b[2 * idx] += 2 * d->sx * l->ui_axy[idx];
                    l->ui_b[2 * idx + 1] += 2 * d->sy * l->ui_axy[idx];
                    l->v[2 * idx] = d->nvx;
                    l->v[2 * idx + 1] = d->nvy;
                    improving = 1;
                    st->commits += 1;
                    l->impmask[d->imp_idx] |= 1u << ((px % 5) + (py % 5) * 5);
                } else if (d->imp_idx >= 0) {
                    l->impmask[d->imp_idx] &= ~(1u << ((d->px % 5) + (d->py % 5) * 5));
                }
            }
            /* UpdateSSIM over tile + 2-pixel halo */
            for (int y = oy - 2; y < oy + TILE_H + 2; ++y)
                for (int x = ox - 2; x < ox + TILE_W + 2; ++x)
                    if (contains(l, x, y)) {
                        int idx = y * w + x;
                        l->value[idx] = vmo_ssim(l->mean[2 * idx], l->mean[2 * idx + 1],
                                                 l->var[2 * idx], l->var[2 * idx + 1],
                                                 l->cross[idx], l->counter[idx], P->ssim_clamp);
                    }
        }
    return improving;
}

/* one iteration = the 4 launches of morph.cu:1382-1385 */
int vmo_optimize_iter(vmo_level *l, const vmo_params *P, double *stats)
{
    tables_init();
    const int gx = (l->w + PITCH_X - 1) / PITCH_X, gy = (l->h + PITCH_Y - 1) / PITCH_Y;
    static const int offs[4][2] = {{0, 0}, {TILE_W, 0}, {0, TILE_H}, {TILE_W, TILE_H}};
    int improving = 0;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int launch = 0; launch < 4; ++launch) {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1) \
    reduction(| : improving) reduction(+ : s0, s1, s2, s3)
        for (int t = 0; t < gx * gy; ++t) {
            int bx = t % gx, by = t / gx;
            sweep_stats st = {0, 0, 0, 0};
            improving |= optimize_tile(l, P, bx * PITCH_X + offs[launch][0],
                                       by * PITCH_Y + offs[launch][1], &st);
            s0 += st.visits; s1 += st.cand; s2 += st.commits; s3 += st.evals;
        }
    }
    if (stats) {
        stats[0] += s0; stats[1] += s1; stats[2] += s2; stats[3] += s3;
    }
    return improving;
}

/* Morph::optimize_level, morph.cu:1378-1390 (middle page only: depth 1) */
int vmo_optimize_level(vmo_level *l, const vmo_params *P, float max_iter, double *stats)
{
    int iter = 0, improving;
    do {
        improving = vmo_optimize_iter(l, P, stats);
        iter++;
    } while (iter < max_iter && improving);
    return iter;
}

/* ------------------------------------------------------------------------- */
/* upsample(), spatial half: upsample.cu:260-286 with                         */
/* rod::kernel_upsample<box_sampler> include/util/imgop_upsample.cu:17-31,88  */
/* and conv_to_block_of_arrays upsample.cu:9-26                               */
void vmo_upsample_v(vmo_level *dst, const vmo_level *src)
{
    const float tw = (float)src->w / dst->w, th = (float)src->h / dst->h;
    const float mx = (float)dst->w / src->w, my = (float)dst->h / src->h;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < dst->h; ++y)
        for (int x = 0; x < dst->w; ++x) {
            float s[2];
            vmo_tex2d_f2(src->v, src->w, src->h, (x + 0.5f) * tw, (y + 0.5f) * th, s);
            dst->v[2 * (y * dst->w + x)] = s[0] * mx;
            dst->v[2 * (y * dst->w + x) + 1] = s[1] * my;
        }
}

/* ------------------------------------------------------------------------- */
/* Morph::cpu_optimize_level, morph.cu:419-590.                               */
/* The reference forms the dense wh x wh matrix in float and multiplies by    */
/* cv::Mat::inv() (LU; SVD pseudo-inverse if LU reports singular).  OpenCV is */
/* absent here, so the arithmetic of that inverse cannot be reproduced; the   */
/* oracle assembles the same matrix in band storage (half bandwidth 2w) in    */
/* double and solves by banded Cholesky.  A zero right-hand side gives v = 0  */
/* exactly, as the reference does.  A singular system with a non-zero         */
/* right-hand side (fewer than three non-collinear constraints, no boundary   */
/* condition) gets a relative ridge of 1e-9, the limit the pseudo-inverse     */
/* takes.  Returns 0 on success.                                              */
static void band_add(double *ab, int kd, int n, int i, int j, double val)
{
    /* symmetric: store lower band, ab[(i-j) + j*(kd+1)] for i>=j */
    (void)n;
    if (j > i)
        return; /* upper entries are the mirror image; assembled rows are symmetric */
    ab[(size_t)(i - j) + (size_t)j * (kd + 1)] += val;
}

int vmo_coarse_solve(vmo_level *l, int w0, int h0, const vmo_params *P,
                     const vmo_constraint *c, int ncon)
{
    return vmo_coarse_solve_page(l, w0, h0, P, c, ncon, 1);
}

/* one page of a level of `depth` pages: BCOND_BORDER adds the border diagonal `depth` times
 * (the `for(int t=0;t<d;t++)` of morph.cu:536-560 sits inside the per-page loop) */
int vmo_coarse_solve_page(vmo_level *l, int w0, int h0, const vmo_params *P,
                          const vmo_constraint *c, int ncon, int depth)
{
    const int w = l->w, h = l->h, n = w * h, kd = 2 * w;
    const double wt = (double)P->w_tps * 2.0;
    double *ab = (double *)calloc((size_t)(kd + 1) * n, sizeof(double));
    double *bx = (double *)calloc(n, sizeof(double));
    double *by = (double *)calloc(n, sizeof(double));
    int any_rhs = 0;
#define A_(r, cc, val) band_add(ab, kd, n, (r), (cc), (val))
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int i = y * w + x;
            if (x > 1) { A_(i, i - 2, 1.0 * wt); A_(i, i - 1, -2.0 * wt); A_(i, i, 1.0 * wt); }
            if (x > 0 && x < w - 1) { A_(i, i - 1, -2.0 * wt); A_(i, i, 4.0 * wt); A_(i, i + 1, -2.0 * wt); }
            if (x < w - 2) { A_(i, i, 1.0 * wt); A_(i, i + 1, -2.0 * wt); A_(i, i + 2, 1.0 * wt); }
            if (y > 1) { A_(i, i - 2 * w, 1.0 * wt); A_(i, i - w, -2.0 * wt); A_(i, i, 1.0 * wt); }
            if (y > 0 && y < h - 1) { A_(i, i - w, -2.0 * wt); A_(i, i, 4.0 * wt); A_(i, i + w, -2.0 * wt); }
            if (y < h - 2) { A_(i, i, 1.0 * wt); A_(i, i + w, -2.0 * wt); A_(i, i + 2 * w, 1.0 * wt); }
            if (x > 0 && y > 0) { A_(i, i - w - 1, 2.0 * wt); A_(i, i - w, -2.0 * wt); A_(i, i - 1, -2.0 * wt); A_(i, i, 2.0 * wt); }
            if (x < w - 1 && y > 0) { A_(i, i - w, -2.0 * wt); A_(i, i - w + 1, 2.0 * wt); A_(i, i, 2.0 * wt); A_(i, i + 1, -2.0 * wt); }
            if (x > 0 && y < h - 1) { A_(i, i - 1, -2.0 * wt); A_(i, i, 2.0 * wt); A_(i, i + w - 1, 2.0 * wt); A_(i, i + w, -2.0 * wt); }
            if (x < w - 1 && y < h - 1) { A_(i, i, 2.0 * wt); A_(i, i + 1, -2.0 * wt); A_(i, i + w, -2.0 * wt); A_(i, i + w + 1, 2.0 * wt); }
        }
    /* UI terms, morph.cu:471-505 */
    for (int k = 0; k < ncon; ++k) {
        float x0 = (float)((c[k].lx + 0.5) / w0 * w - 0.5f);
        float y0 = (float)((c[k].ly + 0.5) / h0 * h - 0.5f);
        float x1 = (float)((c[k].rx + 0.5) / w0 * w - 0.5f);
        float y1 = (float)((c[k].ry + 0.5) / h0 * h - 0.5f);
        float weight = c[k].weight;
        float con_x = (x0 + x1) / 2.0f, con_y = (y0 + y1) / 2.0f;
        float vx = (x1 - x0) / 2.0f, vy = (y1 - y0) / 2.0f;
        for (int y = (int)floor(con_y); y <= (int)ceil(con_y); ++y)
            for (int x = (int)floor(con_x); x <= (int)ceil(con_x); ++x)
                if (x >= 0 && x < w && y >= 0 && y < h) {
                    float bw = (float)((1.0 - fabs(y - con_y)) * (1.0 - fabs(x - con_x)) * weight);
                    int i = y * w + x;
                    A_(i, i, (double)(bw * P->w_ui * l->inv_wh * 2.0f));
                    bx[i] += (double)(bw * vx * P->w_ui * l->inv_wh * 2.0f);
                    by[i] += (double)(bw * vy * P->w_ui * l->inv_wh * 2.0f);
                    any_rhs = 1;
                }
    }
    /* boundary condition, morph.cu:507-562 */
    double bd = (double)(P->w_ui * l->inv_wh);
    if (P->bcond == VMO_BCOND_CORNER) {
        int idx[4] = {0, (h - 1) * w, (h - 1) * w + (w - 1), w - 1};
        for (int k = 0; k < 4; ++k) A_(idx[k], idx[k], bd);
    } else if (P->bcond == VMO_BCOND_BORDER) {
        bd *= depth;
        for (int x = 0; x < w; ++x) { A_(x, x, bd); A_((h - 1) * w + x, (h - 1) * w + x, bd); }
        for (int y = 1; y < h - 1; ++y) { A_(y * w, y * w, bd); A_(y * w + w - 1, y * w + w - 1, bd); }
    }
#undef A_
    size_t nn = (size_t)n;
    if (!any_rhs) {
        memset(l->v, 0, 2 * nn * sizeof(float));
        free(ab); free(bx); free(by);
        return 0;
    }
    /* banded Cholesky, retrying with a ridge if a pivot is not positive */
    double *L = (double *)malloc((size_t)(kd + 1) * n * sizeof(double));
    double ridge = 0, tr = 0;
    for (int j = 0; j < n; ++j) tr += ab[(size_t)j * (kd + 1)];
    int ok = 0;
    for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
        memcpy(L, ab, (size_t)(kd + 1) * n * sizeof(double));
        if (ridge > 0)
            for (int j = 0; j < n; ++j) L[(size_t)j * (kd + 1)] += ridge;
        ok = 1;
        for (int j = 0; j < n && ok; ++j) {
            double *cj = L + (size_t)j * (kd + 1);
            double piv = cj[0];
            double scale = tr / n;
            if (!(piv > 1e-12 * scale)) { ok = 0; break; }
            double dj = sqrt(piv);
            cj[0] = dj;
            int m = (n - 1 - j) < kd ? (n - 1 - j) : kd;
            for (int r = 1; r <= m; ++r) cj[r] /= dj;
            for (int cidx = 1; cidx <= m; ++cidx) {
                double lc = cj[cidx];
                if (lc == 0) continue;
                double *ck = L + (size_t)(j + cidx) * (kd + 1);
                for (int r = cidx; r <= m; ++r) ck[r - cidx] -= cj[r] * lc;
            }
        }
        if (!ok) ridge = (ridge == 0) ? 1e-9 * tr / n : ridge * 100;
    }
    int rc = ok ? 0 : -1;
    if (ok) {
        double *rhs[2] = {bx, by};
        for (int q = 0; q < 2; ++q) {
            double *b = rhs[q];
            for (int j = 0; j < n; ++j) { /* forward */
                double *cj = L + (size_t)j * (kd + 1);
                b[j] /= cj[0];
                int m = (n - 1 - j) < kd ? (n - 1 - j) : kd;
                for (int r = 1; r <= m; ++r) b[j + r] -= cj[r] * b[j];
            }
            for (int j = n - 1; j >= 0; --j) { /* backward */
                double *cj = L + (size_t)j * (kd + 1);
                int m = (n - 1 - j) < kd ? (n - 1 - j) : kd;
                double s = b[j];
                for (int r = 1; r <= m; ++r) s -= cj[r] * b[j + r];
                b[j] = s / cj[0];
            }
        }
        for (int i = 0; i < n; ++i) {
            l->v[2 * i] = (float)bx[i];
            l->v[2 * i + 1] = (float)by[i];
        }
    }
    free(L); free(ab); free(bx); free(by);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Diagnostic: total energy implied by the change terms of morph.cu:730-761   */
void vmo_energy(const vmo_level *l, const vmo_params *P, double *out3)
{
    double es = 0, et = 0, eu = 0;
    size_t n = (size_t)l->w * l->h;
    for (size_t i = 0; i < n; ++i) {
        es += 1.0 - (double)l->value[i];
        et += 0.5 * ((double)l->v[2 * i] * l->tps_b[2 * i] + (double)l->v[2 * i + 1] * l->tps_b[2 * i + 1]);
        if (l->ui_axy[i] > 0)
            eu += ((double)l->ui_b[2 * i] * l->ui_b[2 * i] + (double)l->ui_b[2 * i + 1] * l->ui_b[2 * i + 1]) /
                  (4.0 * l->ui_axy[i]);
    }
    out3[0] = (double)P->w_ssim * es * l->inv_wh;
    out3[1] = (double)P->w_tps * et;
    out3[2] = (double)P->w_ui * eu * l->inv_wh;
}
