/*
 * vm_oracle.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * A plain-C restatement of the halfway-domain morph solver of
 * liaojing/videomorphing, written from the reference's CUDA/C++ sources as a
 * specification.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 *
 * Who may use this: tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg -- as the checker / the timed CPU baseline only.  The
 * product (videomorphing_amd/, include/vmorph.h) never links or calls it.
 *
 * PARITY STATUS: the reference holds no tests, golden vectors or fixtures for
 * this path, none of its CUDA sources build here (nvcc,
 * OpenCV, Qt, MKL absent) and Algorithm/stencils.cpp only builds with
 * stand-in headers, which is not allowed.  The oracle is therefore pinned by
 *   (1) cross-consistency of two independent statements of the thin-plate
 *       operator inside the reference (stencils.cpp:156-261 vs the dense
 *       matrix rows of morph.cu:439-469), checked for all 25 border classes,
 *   (2) the table rows the survey recorded from a run of stencils.cpp
 *       (tests/golden/stencil_rows.json), and
 *   (3) analytic known-answer tests (tests/test_oracle_kat.py).
 * With no reference-run outputs for the iterative optimizer itself, this is
 * "parity unpinned" in the sense of the task statement for the optimizer
 * trajectory; see DESIGN.md section (c).
 *
 * Numerics: float32 throughout, evaluated in the reference's expression
 * order, compiled with -ffp-contract=off so that every operation is an IEEE
 * basic operation (+,-,*,/,sqrt).  CUDA's 1.8 fixed-point texture filtering
 * weights are NOT emulated: bilinear taps use exact float weights.
 * Orders the reference leaves to atomics are fixed as row-major over the
 * committing pixels of a phase.
 */
#ifndef VM_ORACLE_H
#define VM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VMO_BCOND_NONE = 0, VMO_BCOND_CORNER = 1, VMO_BCOND_BORDER = 2 };

/* KernParameters, Algorithm/parameters.h:54-72 */
typedef struct {
    float w_temp, w_ui, w_tps, w_ssim;
    float ssim_clamp;
    float eps;
    int   bcond;
} vmo_params;

/* One connected point pair of Parameters::lp/rp/cnt already resolved to
 * full-resolution pixel coordinates (Algorithm/morph.cu:357-366). */
typedef struct {
    float lx, ly;      /* left  point, full-res pixel index */
    float rx, ry;      /* right point, full-res pixel index */
    float weight;      /* MIN(weight_l, weight_r)            */
} vmo_constraint;

/* One page of one PyramidLevel (Algorithm/Pyramid.h:52-98), tight rows. */
typedef struct {
    int w, h;
    float inv_wh;                 /* pyramid.cu:537 */
    int imp_rs, imp_rows;         /* pyramid.cu:538-539 */
    float *img0, *img1;           /* w*h luma, [0,255] */
    float *v;                     /* 2*w*h, interleaved (x,y) */
    float *luma, *mean, *var;     /* 2*w*h each, .x = img0, .y = img1 */
    float *cross, *value, *counter;
    float *tps_axy, *tps_b;       /* w*h, 2*w*h */
    float *ui_axy, *ui_b;         /* w*h, 2*w*h */
    uint32_t *impmask;            /* imp_rs*imp_rows */
    /* temporal coherence (a page of a video level; upsample.cu:190-258, morph.cu:752-759) */
    float *temp_ref, *temp_mask;  /* 2*w*h, w*h: lvl.temp.ref / lvl.temp.mask of this page */
    float factor_d;               /* Pyramid.h:65, pyramid.cu:470-477 */
    int   flag;                   /* the `flag` argument of kernel_optimize_level for this page */
} vmo_level;

/* field ids for vmo_level_field() */
enum {
    VMO_F_IMG0 = 0, VMO_F_IMG1, VMO_F_V, VMO_F_LUMA, VMO_F_MEAN, VMO_F_VAR,
    VMO_F_CROSS, VMO_F_VALUE, VMO_F_COUNTER, VMO_F_TPS_AXY, VMO_F_TPS_B,
    VMO_F_UI_AXY, VMO_F_UI_B, VMO_F_IMPMASK, VMO_F_TEMP_REF, VMO_F_TEMP_MASK
};

/* --- small pieces --------------------------------------------------------- */
int   vmo_calc_border(int p, int dim);                       /* morph.cu:39-81 */
float vmo_ssim(float mx, float my, float vx, float vy, float cross,
               float counter, float clamp);                  /* morph.cu:85-118 */
float vmo_tex2d(const float *img, int w, int h, float x, float y);
void  vmo_tex2d_f2(const float *img, int w, int h, float x, float y, float *out2);
/* diagnostic: 0 = exact float weights (default), 1 / 2 = CUDA's 8-bit filter weights, rounded / truncated */
void  vmo_set_tex_filter(int mode);
int   vmo_get_tex_filter(void);
void  vmo_tps_stencil(float *out625);                        /* stencils.cpp:156-261 */
void  vmo_tps_rows_from_dense(float *out625);                /* morph.cu:439-469 */
void  vmo_io_stencil(int *out625);                           /* stencils.cpp:10-71 */
void  vmo_improvmask_stencil(uint32_t *out225);              /* stencils.cpp:90-126 */

/* --- level objects -------------------------------------------------------- */
vmo_level *vmo_level_create(int w, int h);
void       vmo_level_destroy(vmo_level *l);
void      *vmo_level_field(vmo_level *l, int field);
void       vmo_level_set_temporal(vmo_level *l, int flag, float factor_d);

/* --- hot path ------------------------------------------------------------- */
/* kernel_initialize_level + init_improving_mask, morph.cu:173-260 (+ the
 * zero-fill of morph.cu:300-314) */
void vmo_init_level(vmo_level *l, float ssim_clamp);
/* UI constraint linearisation, morph.cu:345-388 */
void vmo_splat_constraints(vmo_level *l, int w0, int h0,
                           const vmo_constraint *c, int n);
/* one "iteration" = 4 tile-offset launches of kernel_optimize_level,
 * morph.cu:1281-1345 + 1382-1385.  Returns 1 if any pixel improved.
 * stats (may be NULL): [0] += pixel visits, [1] += candidate (active) visits,
 * [2] += commits, [3] += energy evaluations. */
int  vmo_optimize_iter(vmo_level *l, const vmo_params *p, double *stats);
/* the do/while of Morph::optimize_level, morph.cu:1378-1390; returns the
 * number of iterations executed */
int  vmo_optimize_level(vmo_level *l, const vmo_params *p, float max_iter,
                        double *stats);
/* spatial half of upsample(), upsample.cu:260-286 */
void vmo_upsample_v(vmo_level *dst, const vmo_level *src);
/* Morph::cpu_optimize_level, morph.cu:419-590 (banded solve, see .c) */
int  vmo_coarse_solve(vmo_level *l, int w0, int h0, const vmo_params *p,
                      const vmo_constraint *c, int n);
int  vmo_coarse_solve_page(vmo_level *l, int w0, int h0, const vmo_params *p,
                           const vmo_constraint *c, int n, int depth);
/* total energy of the current field (diagnostic, derived from the terms of
 * morph.cu:730-761): returns E_ssim, E_tps, E_ui in out3 */
void vmo_energy(const vmo_level *l, const vmo_params *p, double *out3);

/* --- compositor ----------------------------------------------------------- */
/* kernel_render_halfway_image, render.cu:16-60.  ext0/ext1: float4 RGBA
 * canvases (w+2ex)x(h+2ex); v,u: w*h float2; out: w*h*3 bytes */
void vmo_render_halfway(uint8_t *out, int w, int h, int ex,
                        float color_fa, float geo_fa, int color_from,
                        const float *ext0, const float *ext1,
                        const float *v, const float *u);
/* CMatchingThread::update_result/Resize, MatchingThread.cpp:22-136: scale v
 * by (W0/W,H0/H) and bilinearly resize to w0 x h0 */
void vmo_blend_v(float *dst, const float *a, const float *b, float fa, size_t n_floats);
void vmo_upscale_result(float *dst, int w0, int h0,
                        const float *v, int w, int h);
/* CPoissonExt::prepare + poissonExtend, PoissonExt.cpp:49-362.  rgba_ext:
 * (w+2ex)x(h+2ex) RGBA8 canvas, in/out.  other: w*h RGBA8 crop of the other
 * image; v: w*h float2; side 1 or 2.  Returns CG iterations used (the
 * reference uses MKL DSS; see .c). */
int  vmo_poisson_extend(uint8_t *rgba_ext, int w, int h, int ex,
                        const uint8_t *other, const float *v, int side,
                        double tol, int max_it, double *rel_res);
/* classification + fill only (PoissonExt.cpp:49-141): type[] (0/1/2) out */
int  vmo_poisson_prepare(uint8_t *rgba_ext, int w, int h, int ex,
                         const uint8_t *other, const float *v, int side,
                         int *type_out);

/* luma pyramid of Pyramid::build (pyramid.cu:203-211, 268-279, 355-364 driving
 * include/resample): rgb h*w*3 bytes -> lumas of levels 1..nlevels concatenated.
 * Pinned by tests/golden/pyramid_ref.npz (outputs of the reference's own library). */
void vmo_luma_pyramid(const uint8_t *rgb, int w, int h, int nlevels, float *out);

/* test hooks: prevent_foldover (morph.cu:872-883) and energy_change (:730-761)
 * evaluated at one pixel of an initialised level */
float vmo_dbg_foldover(const vmo_level *l, const vmo_params *p, int px, int py, float gx, float gy);
float vmo_dbg_energy_change(const vmo_level *l, const vmo_params *p, int px, int py, float dx, float dy);

/* --- temporal coherence path (vm_oracle_temporal.c) ------------------------------------- */
/* A video level is `depth` pages (vmo_level objects of one size) plus four flow fields per
 * page (f0/f1: forward flow of video 0/1 from frame t to t+1, b0/b1: backward), tight
 * h*w*2 floats.  The host control flow (Morph::optimize_level's page chains, upsample()'s
 * page loop) lives in tests/oracle.py; these are the kernels.
 *
 * ORDER NOTE: temp_ref scatters with float atomicAdd in an unspecified order
 * (upsample.cu:57-58), so the reference's own result is not reproducible bit for bit.
 * Here -- and in the HIP path -- each contribution (computed in float exactly as the
 * reference writes it) is accumulated in 64-bit fixed point (value * 2^32, round to nearest
 * even): order-independent, hence comparable bit for bit.  The deviation from any float
 * summation order is below one float ulp of the sum. */
/* temp_ref, upsample.cu:28-62: splat of v_prev advected by the mean of the two flows into
 * acc (3 int64 per pixel: v.x, v.y, weight; caller zero-fills).  ssim may be NULL. */
void vmo_temp_splat(int w, int h, const float *v_prev, const float *f0, const float *f1,
                    const float *ssim, int64_t *acc);
/* fixed point -> float, then interpolate_temp_ref (upsample.cu:64-77) */
void vmo_temp_normalise(int w, int h, const int64_t *acc, float *v_cur, float *weight);
/* initialize_temp, upsample.cu:214-258: dst's temp.ref / temp.mask from the neighbouring
 * page `src` (its v and ssim.value) advected by src's flows (fa, fb = f0, f1 of src for
 * dir < 0; b0, b1 of src for dir > 0); sets dst->flag = 1 */
void vmo_initialize_temp(vmo_level *dst, const vmo_level *src, const float *fa, const float *fb);
/* the temporal half of upsample(), upsample.cu:297-338, for one in-between page: splat of the
 * previous page (its f0, f1) and of the next page (its b0, b1), normalise, smooth (:80-111),
 * fill_zeros_x (:115-151); fill_zeros_y (:153-189) only touches the discarded weight array */
void vmo_temporal_fill(int w, int h, const float *v_prev, const float *f0_prev, const float *f1_prev,
                       const float *v_next, const float *b0_next, const float *b1_next, float *v_out);
/* flow half of Pyramid::build (pyramid.cu:284-321, 375-404): one flow field through
 * load(-50, 50) -> scale() -> store(-50, 50), then x (wout/w, hout/h) when the size shrinks */
void vmo_flow_scale(const float *flow, int w, int h, int wout, int hout, float *out);
/* temporal concatenation (pyramid.cu:406-442): f(p) += BiLinear(f_next, p + f(p)) */
void vmo_flow_concat(float *f, const float *f_next, int w, int h);
/* order in which the commits of a phase are applied (the reference leaves it to atomics,
 * morph.cu:951-1015): 0 = row-major over the committing pixels (the oracle's definition), 1 = that
 * sequence reversed, 2 = column-major, 3 = column-major reversed -- equally legal orders, used to
 * measure how far legal trajectories drift apart */
void vmo_set_commit_order(int order);

void vmo_set_threads(int n);   /* OpenMP threads for the timed CPU baseline */
int  vmo_get_threads(void);

#ifdef __cplusplus
}
#endif
/* quadratic motion path of one frame (vm_oracle_qpath.c), QuadraticPath.cpp:24-223:
 * v, u_out: rows*cols*2 floats; jopt_out (optional): rows*cols*4 */
int vmo_quadratic_path(const float *v, int cols, int rows, double tol, int max_it, float *u_out,
                       float *jopt_out, double *rel_res);


/* ---- synchronisation stage (vm_oracle_sync.c; SURVEY 8(f) "(later)" row) ------------------- */
int   vmo_sync_levels(int w, int h, int d, int start_res, int *lw, int *lh, int *ld, int cap); /* pyramid.cu:143-163 */
void  vmo_sync_row(int x, int y, int z, int w, int h, int d, float w_tps, float ui, float *data125); /* SyncThread.cpp:190-262 */
extern const int vmo_sync_taps[25][3];
int   vmo_sync_state(int p, int n);
void  vmo_sync_table(int w, int h, int d, float w_tps, float *tab125x25);
void  vmo_sync_ui(int w, int h, int d, int w0, int h0, const int *cons6, int n, float w_ui,
                  float *diag, float *bx, float *by, float *bz);                              /* SyncThread.cpp:155-187 */
void  vmo_sync_diag(int w, int h, int d, float w_tps, const float *ui, float *diag);
float vmo_sync_dot(const float *a, const float *b, int w, int h, int d);
void  vmo_sync_apply(int w, int h, int d, float w_tps, const float *ui, const float *p, float *out);
int   vmo_sync_solve_level(int w, int h, int d, int w0, int h0, const int *cons6, int ncons,
                           float w_ui, float w_tps, float max_iter, float *x, float *y, float *z,
                           float *resid3);                                                     /* SyncThread.cpp:290-480 */
void  vmo_sync_upsample(float *dst, int dw, int dh, const float *src, int sw, int sh, float ratio); /* upsample.cu:343-375 */
void  vmo_sync_result(const float *X, const float *Y, const float *Z, int w, int h, int w0, int h0, float *out4); /* SyncThread.cpp:482-521 */
void  vmo_render_resample(uint8_t *out, int w, int h, int d, float fa, int frame, const float *vec,
                          const uint8_t *video0, const uint8_t *video1, const float *forw0, const float *forw1); /* render.cu:99-246 */

#endif
