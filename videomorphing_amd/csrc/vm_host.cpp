// vm_host.cpp -- host-side numerical pieces of the product path.
//
// Coarsest-level solve, the role of Morph::cpu_optimize_level
// (Algorithm/morph.cu:419-590; Algorithm/cpuoptim.cpp shows the same code).
// The reference fills a dense wh x wh float matrix and multiplies the right
// hand sides by cv::Mat::inv().  The matrix is the thin-plate Hessian (a
// 13-point stencil, bandwidth 2w) plus a diagonal, so this implementation
// keeps only the band, in double, and factorises it as U^T U.  Same linear
// system, O(n w^2) instead of O(n^3); a zero right-hand side returns v = 0 as
// the reference does.
#include "vm_host.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

// Hessian row of the bending energy for the pixel with border classes (By,Bx):
// identical to the device table (vm_api.cpp build_tables); recomputed here
// from the one-dimensional pieces to keep this file self-contained.
struct Band {
    int n, kd;
    std::vector<double> a; // row i holds A(i, i..i+kd)
    Band(int n_, int kd_) : n(n_), kd(kd_), a((size_t)n_ * (kd_ + 1), 0.0) {}
    double &at(int i, int j) { return a[(size_t)i * (kd + 1) + (j - i)]; }
    void add(int i, int j, double v)
    {
        if (j >= i && j < n) at(i, j) += v; // upper triangle only; the matrix is symmetric
    }
};

} // namespace

int vm_host_coarse_solve(int w, int h, int w0, int h0, const vm_kern_params &kp,
                         const vm_constraint *cons, int ncon, float *v_out, int depth)
{
    const int n = w * h, kd = 2 * w;
    memset(v_out, 0, sizeof(float) * 2 * (size_t)n);
    const float inv_wh = 1.0f / (w * h);

    std::vector<double> bx(n, 0.0), by(n, 0.0);
    bool any = false;
    Band A(n, kd);

    // user constraints (morph.cu:471-505)
    for (int k = 0; k < ncon; ++k) {
        const vm_constraint &c = cons[k];
        float x0 = (float)(((double)c.lx + 0.5) / w0 * w - 0.5f);
        float y0 = (float)(((double)c.ly + 0.5) / h0 * h - 0.5f);
        float x1 = (float)(((double)c.rx + 0.5) / w0 * w - 0.5f);
        float y1 = (float)(((double)c.ry + 0.5) / h0 * h - 0.5f);
        float con_x = (x0 + x1) / 2.0f, con_y = (y0 + y1) / 2.0f;
        float vx = (x1 - x0) / 2.0f, vy = (y1 - y0) / 2.0f;
        for (int y = (int)std::floor(con_y); y <= (int)std::ceil(con_y); ++y)
            for (int x = (int)std::floor(con_x); x <= (int)std::ceil(con_x); ++x) {
                if (x < 0 || x >= w || y < 0 || y >= h) continue;
                float bw = (float)((1.0 - std::fabs(y - con_y)) * (1.0 - std::fabs(x - con_x)) * c.weight);
                int i = y * w + x;
                A.add(i, i, (double)(bw * kp.w_ui * inv_wh * 2.0f));
                bx[i] += (double)(bw * vx * kp.w_ui * inv_wh * 2.0f);
                by[i] += (double)(bw * vy * kp.w_ui * inv_wh * 2.0f);
                any = true;
            }
    }
    if (!any)
        return VM_OK; // B = 0  =>  v = 0 (morph.cu:565-570 with a zero right-hand side)

    // thin-plate part (morph.cu:439-469): every placement of dxx, dyy (weight 1)
    // and of the 2x2 mixed difference (weight 2) that fits, Hessian = 2 w K^T K
    const double wt = 2.0 * (double)kp.w_tps;
    auto add_op = [&](const int *idx, const double *c, int m, double weight) {
        for (int p = 0; p < m; ++p)
            for (int q = 0; q < m; ++q)
                A.add(idx[p], idx[q], wt * weight * c[p] * c[q]);
    };
    const double c3[3] = {1, -2, 1}, c4[4] = {1, -1, -1, 1};
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int i = y * w + x;
            if (x >= 1 && x <= w - 2) { int id[3] = {i - 1, i, i + 1}; add_op(id, c3, 3, 1.0); }
            if (y >= 1 && y <= h - 2) { int id[3] = {i - w, i, i + w}; add_op(id, c3, 3, 1.0); }
            if (x <= w - 2 && y <= h - 2) { int id[4] = {i, i + 1, i + w, i + w + 1}; add_op(id, c4, 4, 2.0); }
        }
    // boundary condition (morph.cu:507-562); BCOND_BORDER sits in a `for (t < depth)` loop
    // inside the per-page loop there (:536-560): the border diagonal is added depth times
    double bd = (double)(kp.w_ui * inv_wh);
    if (kp.bcond == VM_BCOND_CORNER) {
        const int id[4] = {0, (h - 1) * w, (h - 1) * w + w - 1, w - 1};
        for (int k = 0; k < 4; ++k) A.add(id[k], id[k], bd);
    } else if (kp.bcond == VM_BCOND_BORDER) {
        bd *= depth;
        for (int x = 0; x < w; ++x) { A.add(x, x, bd); A.add((h - 1) * w + x, (h - 1) * w + x, bd); }
        for (int y = 1; y < h - 1; ++y) { A.add(y * w, y * w, bd); A.add(y * w + w - 1, y * w + w - 1, bd); }
    }

    // U^T U factorisation of the band; a non-positive pivot (singular system:
    // fewer than three non-collinear constraints and no boundary condition)
    // retries with a small ridge, the limit cv::DECOMP_SVD's pseudo-inverse takes
    double tr = 0;
    for (int i = 0; i < n; ++i) tr += A.at(i, i);
    const double scale = tr / n;
    double ridge = 0;
    std::vector<double> U;
    bool ok = false;
    for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
        U = A.a;
        auto u = [&](int i, int j) -> double & { return U[(size_t)i * (kd + 1) + (j - i)]; };
        if (ridge > 0)
            for (int i = 0; i < n; ++i) u(i, i) += ridge;
        ok = true;
        for (int i = 0; i < n; ++i) {
            double piv = u(i, i);
            if (!(piv > 1e-12 * scale)) { ok = false; break; }
            double d = std::sqrt(piv);
            int m = std::min(kd, n - 1 - i);
            u(i, i) = d;
            for (int r = 1; r <= m; ++r) u(i, i + r) /= d;
            for (int r = 1; r <= m; ++r) {
                double f = u(i, i + r);
                if (f == 0) continue;
                double *row = &u(i + r, i + r);
                const double *src = &u(i, i + r);
                for (int s = 0; s <= m - r; ++s) row[s] -= f * src[s];
            }
        }
        if (!ok) ridge = ridge == 0 ? 1e-9 * scale : ridge * 100;
    }
    if (!ok)
        return vm_fail(VM_E_NUMERIC, "vm_coarse_solve: factorisation failed");
    auto u = [&](int i, int j) -> double { return U[(size_t)i * (kd + 1) + (j - i)]; };
    for (std::vector<double> *b : {&bx, &by}) {
        std::vector<double> &z = *b;
        for (int i = 0; i < n; ++i) { // U^T y = b
            z[i] /= u(i, i);
            int m = std::min(kd, n - 1 - i);
            for (int r = 1; r <= m; ++r) z[i + r] -= u(i, i + r) * z[i];
        }
        for (int i = n - 1; i >= 0; --i) { // U x = y
            int m = std::min(kd, n - 1 - i);
            double s = z[i];
            for (int r = 1; r <= m; ++r) s -= u(i, i + r) * z[i + r];
            z[i] = s / u(i, i);
        }
    }
    for (int i = 0; i < n; ++i) {
        v_out[2 * i] = (float)bx[i];
        v_out[2 * i + 1] = (float)by[i];
    }
    return VM_OK;
}
