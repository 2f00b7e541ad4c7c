/*
 * vmorph.h -- C-ABI of the MI355X-native halfway-domain morph solver.
 *
 * This is the drop-in boundary for the hot path of liaojing/videomorphing:
 * the entry points below are what a binding of the reference's L3 solver API
 * (Algorithm/morph.h, Pyramid.h, parameters.h; UI/RenderWidget.h:52-57;
 * Algorithm/PoissonExt.h) would call.  Each declaration cites the reference
 * interface it replaces (paths relative to the reference repository).
 * INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *  - extern "C", opaque handles, plain pointers and sizes; no exceptions
 *    cross the ABI.  Every function returns VM_OK (0) or a negative VM_E_*
 *    code; vm_last_error() returns a thread-local message for the last error.
 *    (The reference throws std::runtime_error from rod::check_cuda_error,
 *    include/util/error.cpp:9-23; the C++ facade re-raises from these codes.)
 *  - One vm_ctx per device and host thread; all work of a context is
 *    stream-ordered on the context's HIP stream.  The caller owns host memory,
 *    the context owns device memory.
 *  - One vm_pyr holds ONE frame pair (a depth-1 pyramid: the independent-pair
 *    formulation of BASELINE.json; the reference's temporal coupling,
 *    morph.cu:1394-1439, is out of scope).  Level 0 is the finest level, level
 *    nlevels-1 the coarsest one, which only ever holds `v` (it is solved
 *    densely on the host, morph.cu:152).
 *  - Images and fields cross the boundary as tight or pitched row-major host
 *    arrays; `pitch` counts ELEMENTS of the array's scalar type per row
 *    (floats), 0 meaning tight.  float2 fields are interleaved (x,y).
 *  - There is no CPU fallback: if the HIP device or the code object is
 *    missing, vm_ctx_create fails with VM_E_DEVICE.
 */
#ifndef VMORPH_H
#define VMORPH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VM_OK            0
#define VM_E_INVALID    -1   /* bad argument */
#define VM_E_DEVICE     -2   /* HIP error (message in vm_last_error) */
#define VM_E_STATE      -3   /* call out of order (e.g. optimize before init) */
#define VM_E_NUMERIC    -4   /* coarse solve / Poisson solve failed */
#define VM_E_CANCELLED  -5   /* run_flag went to 0 */

typedef struct vm_ctx vm_ctx;
typedef struct vm_pyr vm_pyr;

/* enum BoundaryCondition, Algorithm/parameters.h:9-14 */
enum { VM_BCOND_NONE = 0, VM_BCOND_CORNER = 1, VM_BCOND_BORDER = 2 };

/* struct KernParameters, Algorithm/parameters.h:54-72 (same fields, same order) */
typedef struct {
    float w_temp, w_ui, w_tps, w_ssim;
    float ssim_clamp;
    float eps;
    int   bcond;
} vm_kern_params;

/* One connected point pair (Parameters::lp/rp/cnt, parameters.h:16-27, 45-47)
 * resolved by the caller to full-resolution pixel indices, as consumed at
 * Algorithm/morph.cu:357-366: weight = MIN(weight_l, weight_r). */
typedef struct {
    float lx, ly, rx, ry;
    float weight;
} vm_constraint;

/* arithmetic mode of the optimizer kernels */
enum {
    VM_MATH_EXACT = 0,  /* IEEE +,-,*,/,sqrt, no contraction: bit-identical to the CPU oracle */
    VM_MATH_FAST  = 1,  /* fused multiply-add, v_rcp/v_sqrt approximations: the
                           analogue of the reference's --use_fast_math build
                           (MdiEditor.vcxproj:208-213) */
    VM_MATH_EXACT_FMA = 2, /* diagnostic: the EXACT source compiled with -ffp-contract=fast --
                           fused multiply-adds wherever the compiler contracts, IEEE division
                           and square root -- i.e. what nvcc's default --fmad=true makes of the
                           reference source WITHOUT --use_fast_math: one more legal rounding of
                           the same algorithm (sweep kernels only; not bit-comparable to the
                           oracle), used to measure the reproducibility floor FAST is judged
                           against (tests/test_gpu_fullsize.py) */
    VM_MATH_REF_FASTMATH = 3, /* diagnostic: the EXACT source as the reference's project file compiles
                           it -- --use_fast_math (MdiEditor.vcxproj:208-213): contraction plus
                           approximate division (x * rcp(y), CUDA's __fdividef) and square root.  The
                           reference's own expressions in the reference's own arithmetic: the third
                           member of the family of legal builds */
    VM_MATH_REF_TEX8 = 4,  /* diagnostic: the EXACT source (IEEE, no contraction) with every texture
                           fetch -- both images, Algorithm/morph.cu:29-30, 316-322 (taps at :212-213,
                           680-681, 960-961), and the inter-level upsample, include/util/
                           imgop_upsample.cu:17-31 -- filtered with CUDA's 9-bit fixed-point bilinear
                           weights (8 fractional bits, rounded to nearest): the one piece of the
                           reference BINARY's arithmetic no other mode has; the finite-difference
                           step eps = 0.01 px is 2.56 weight quanta.  The CPU checker under tests/ has
                           the same switch; the two agree bit for bit */
    VM_MATH_REF_TEX8_TRUNC = 5 /* the same with truncated weights (the CUDA guide does not state the
                           rounding rule): sensitivity check */
};

/* progress of one optimize_level call; mirrors the public progress members of
 * class Morph (Algorithm/morph.h:19-20) */
typedef struct {
    int    iters;          /* iterations executed (4 tile-offset sweeps each)  */
    int    improving;      /* 1 if the last executed iteration still improved  */
    double pixel_iters;    /* iters * W * H  (morph.cu:1389)                    */
    float  elapsed_ms;     /* HIP-event time of the sweep kernels on the stream */
    int    launches;       /* sweep kernel launches enqueued                    */
    double active_tiles;   /* tile visits that were not skipped by the mask     */
    double candidates;     /* pixel visits that ran the line search             */
    double commits;        /* accepted moves                                    */
    double evaluations;    /* energy evaluations executed (each: 2 bilinear taps
                              + 25 SSIM terms, morph.cu:671-761)                 */
    /* HIP-event time and launch count of the sweep kernels per schedule:
     * [0] TILE, dense kernel; [1] TILE, lean kernel (pruned sweeps); [2] STEP / SPLIT;
     * [3] SPARSE; [4] PASS */
    float  sched_ms[5];
    int    sched_launches[5];
    /* iterations up to and including the first that accepted no move: what the reference's
     * stopping rule (morph.cu:1390) executes.  A fixed-work run reports iters = max_iter; the
     * iters - iters_live sweeps past convergence are provable no-ops (every mask bit is clear)
     * and are skipped on the device. */
    int    iters_live;
    /* in-kernel clock probe: the first workgroup of every launch of the dense TILE kernel ([0]) and of the PASS
     * kernel ([1]) brackets its sweep with s_memtime (shader cycles) and s_memrealtime (constant 100 MHz); the
     * sums over the call's launches.  clk_shader_ticks / clk_wall_ticks x 100 MHz = the shader clock the chip
     * actually held inside that kernel (0 / 0: the schedule did not run). */
    double clk_shader_ticks[2];
    double clk_wall_ticks[2];
} vm_progress;

/* device-state arrays a test or a UI may read back (vm_level_get_field) */
enum {
    VM_F_IMG0 = 0, VM_F_IMG1, VM_F_V, VM_F_LUMA, VM_F_MEAN, VM_F_VAR, VM_F_CROSS,
    VM_F_VALUE, VM_F_COUNTER, VM_F_TPS_AXY, VM_F_TPS_B, VM_F_UI_AXY, VM_F_UI_B,
    VM_F_IMPMASK,
    /* pages of a video (vm_video_get_field): lvl.temp.ref (float2), lvl.temp.mask, the four
     * flow fields of the page (float2) */
    VM_F_TEMP_REF, VM_F_TEMP_MASK, VM_F_FLOW_F0, VM_F_FLOW_F1, VM_F_FLOW_B0, VM_F_FLOW_B1
};

const char *vm_last_error(void);
const char *vm_version(void);

/* ---- context ------------------------------------------------------------- */
/* replaces MdiEditor::CudaInit device selection, UI/MdiEditor.cpp:54-75 */
int  vm_ctx_create(int device, vm_ctx **out);
void vm_ctx_destroy(vm_ctx *ctx);
int  vm_ctx_sync(vm_ctx *ctx);
/* copy_to_symbol(c_params, KernParameters), Algorithm/morph.cu:1355-1358 */
int  vm_set_params(vm_ctx *ctx, const vm_kern_params *p);
int  vm_get_params(vm_ctx *ctx, vm_kern_params *p);
int  vm_set_math_mode(vm_ctx *ctx, int mode);
/* scheduling of the sweep (results do not depend on it): VM_SWEEP_TILE = one launch
 * per tile-offset pass, one workgroup per tile; VM_SWEEP_SPLIT = two launches per
 * phase (line searches, then the commit), a tile's candidates spread over `parts`
 * workgroups (small levels); VM_SWEEP_STEP = one launch per phase, the commit of a
 * phase folded into the next phase's launch; VM_SWEEP_SPARSE = TILE, but every batch of
 * iterations of a pruned level (fewer than a tenth of the pixels searched) runs as ONE launch
 * in which one workgroup per pair walks the active tiles; VM_SWEEP_PASS = one launch per
 * pass for small levels: a tile's four phases stay inside the launch, its 32 workgroups (one
 * wave per phase pixel) meet at a tile-local barrier between phases; VM_SWEEP_AUTO picks per
 * batch of iterations.
 * threads/parts: 0 = automatic.  `parts` of a FORCED schedule (tests): SPLIT / STEP = workgroups per tile; SPARSE = the
 * LDS capacity of the kernel's word list; PASS = 1: write-through stores from the start, 2: tile groups spread over
 * the XCDs; TILE (FAST) = from how many workgroups per pass (tiles x pairs) a pruned pass takes its listed form -- a
 * scan of the mask lists the tiles a set bit reaches, a fixed grid of workgroups walks the list; 0 = 4096. */
enum { VM_SWEEP_AUTO = 0, VM_SWEEP_TILE = 1, VM_SWEEP_SPLIT = 2, VM_SWEEP_STEP = 3, VM_SWEEP_SPARSE = 4, VM_SWEEP_PASS = 5 };
int  vm_set_tuning(vm_ctx *ctx, int sweep_mode, int threads, int parts);
/* Diagnostic, EXACT arithmetic only: the order in which the commits of one Jacobi phase are
 * folded into the shared window sums.  The reference leaves it to float atomics
 * (morph.cu:951-1015); the oracle and this library fix it as row-major over the committing
 * pixels (order 0).  order 1 applies that sequence reversed, 2 column-major, 3 column-major
 * reversed -- equally legal trajectories, used to measure how far legal runs drift apart
 * (the per-frame chaos floor FAST is judged against, tests/test_gpu_fullsize.py).  Other
 * values: VM_E_INVALID. */
int  vm_set_commit_order(vm_ctx *ctx, int order);
/* Test hooks of the PASS schedule's safety net.  k_pass spins at tile-local barriers with a bounded
 * wait; under VM_SWEEP_AUTO a timeout (compute units masked away or held by someone else) restores the
 * level to where the batch of iterations started, reruns the batch with the STEP schedule and keeps the
 * context off PASS from then on; a forced VM_SWEEP_PASS reports VM_E_DEVICE.  force_timeout(on != 0)
 * makes one workgroup of every following PASS launch walk away from its tile group, so that the rest
 * times out for real (2 s); on == 0 ends that and re-admits PASS.  fallbacks = how many batches this
 * context has rerun with STEP. */
int  vm_dbg_pass_force_timeout(vm_ctx *ctx, int on);
int  vm_dbg_pass_fallbacks(vm_ctx *ctx);
/* Test hook of the SPARSE schedule's resident visits (FAST arithmetic): while the set bits
 * of a pruned level's improving mask fit one tile, k_sparse keeps the window sums around them in LDS across passes
 * and iterations instead of staging and writing back a tile per pass (kernel_optimize_level's LoadSSIM / SaveSSIM,
 * morph.cu:1214-1256, once per residency instead of once per visit) -- the same bits as list-driven visits.
 * mode 0 = automatic (default), 1 = never, 2 = the LDS copy is re-centred after every commit, 3 = residency is given
 * up at the first commit (the two exits a growing active region takes, forced).  Other values: VM_E_INVALID. */
int  vm_dbg_sparse_resident(vm_ctx *ctx, int mode);
/* ... and how many tile visits of this context's solves were served from that LDS copy so far (saturating). */
int  vm_dbg_sparse_resident_visits(vm_ctx *ctx);
/* Diagnostic of the PASS schedule: on which XCD (0..7) each of the first `n` (<= 2048)
 * workgroups of the most recent PASS launch ran (workgroup b belongs to tile group
 * (b / 256) * 8 + b % 8; a group whose 32 workgroups report one XCD keeps its tile in one
 * L2).  The first call only arms the recording (xcc_of_block is filled with 0xFF); placement
 * is a matter of speed, never of correctness. */
int  vm_dbg_pass_placement(vm_ctx *ctx, uint8_t *xcc_of_block, int n);
/* Do the streams of two contexts of one device run side by side?  (The reference has one device context and one stream,
 * UI/MdiEditor.cpp:54-75; a host of this library keeps several: solver streams, compositor lanes.)  The HIP runtime
 * multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, and two streams on one queue run their kernels
 * one after the other; which queue a new stream gets is not the caller's to choose.  *overlap = 1 if a 100 us do-nothing
 * kernel on each stream overlaps the other's, 0 if they serialise.  A host that needs the overlap creates another context
 * when the answer is 0 (keeping the rejected one alive meanwhile, so that the next stream gets another queue). */
int  vm_dbg_streams_overlap(vm_ctx *a, vm_ctx *b, int *overlap);
/* device facts for reports: name (<=255 chars), CU count, HBM bytes */
int  vm_device_info(vm_ctx *ctx, char *name256, int *cus, uint64_t *hbm_bytes);

/* ---- pyramid / levels ---------------------------------------------------- */
/* Pyramid::append_new + PyramidLevel ctor, Algorithm/pyramid.cu:525-543 */
int  vm_pyramid_create(vm_ctx *ctx, int nlevels, const int *w, const int *h, vm_pyr **out);
/* Pyramid::clear, Algorithm/pyramid.cu:19-49 */
void vm_pyramid_destroy(vm_pyr *pyr);
int  vm_pyramid_levels(vm_pyr *pyr);
int  vm_level_dims(vm_pyr *pyr, int lvl, int *w, int *h, int *rowstride);
/* the cudaMemcpy2DToArray uploads of luma in Pyramid::build, pyramid.cu:275-280 */
int  vm_level_upload_luma(vm_pyr *pyr, int lvl, const float *img0, const float *img1, int pitch);
/* the image half of Pyramid::build(video0, video1, ..., start_res),
 * Algorithm/pyramid.cu:166-485, for one pair of RGB8 frames of the level-0 size:
 * load -> per level scale() (include/resample: cubic B-spline generalized sampling)
 * -> store_gray, all on the device; fills img0/img1 of every level that holds images */
int  vm_pyramid_build_rgb(vm_pyr *pyr, const uint8_t *rgb0, const uint8_t *rgb1, int pitch_bytes);
/* lvl.v.copy_from_host, Algorithm/morph.cu:588 */
int  vm_level_set_v(vm_pyr *pyr, int lvl, const float *v_xy, int pitch);
/* the cudaMemcpy2D of CMatchingThread::update_result, MatchingThread.cpp:38-40 */
int  vm_level_get_v(vm_pyr *pyr, int lvl, float *v_xy, int pitch);
/* read back any state array, tight rows: float (h*w), float2 (h*w*2) or, for
 * VM_F_IMPMASK, uint32 ((h+4)/5+2) x ((w+4)/5+2) */
int  vm_level_get_field(vm_pyr *pyr, int lvl, int field, void *host);
/* Morph::clear_level, Algorithm/morph.cu:392-414 */
int  vm_level_clear(vm_pyr *pyr, int lvl);
/* Test hook: overwrite the improving mask of an initialised level (init_improving_mask's array, morph.cu:203-260; the
 * layout vm_level_get_field(VM_F_IMPMASK) returns: ((h + 4) / 5 + 2) rows of (w + 4) / 5 + 2 words, bit x % 5 + 5 (y % 5) of
 * word (y / 5 + 1, x / 5 + 1)).  Any mask is a legal state -- it only says which pixels the next sweep searches -- so tests
 * plant small clusters of set bits to drive the schedules through their pruned regimes. */
int  vm_dbg_level_set_mask(vm_pyr *pyr, int lvl, const uint32_t *words);

/* ---- solver -------------------------------------------------------------- */
/* Morph::cpu_optimize_level, Algorithm/morph.cu:419-590 (host banded solve of
 * the coarsest level, result uploaded to the level's v) */
int  vm_coarse_solve(vm_pyr *pyr, int lvl, int w0, int h0, const vm_constraint *c, int n);
/* upsample(PyramidLevel&dest, PyramidLevel&orig), Algorithm/upsample.cu:260-286 */
int  vm_upsample_v(vm_pyr *pyr, int dst_lvl, int src_lvl);
/* Morph::initialize_level, Algorithm/morph.cu:264-390: window sums, SSIM value,
 * TPS linearisation, improving mask, and the UI-constraint splat (done on the
 * device; w0,h0 = full-resolution size the constraint coordinates refer to) */
int  vm_init_level(vm_pyr *pyr, int lvl, int w0, int h0, const vm_constraint *c, int n);
/* Morph::optimize_level, Algorithm/morph.cu:1353-1391.  Runs until no pixel
 * improves, iter >= max_iter, or *run_flag == 0 (may be NULL).  fixed_work != 0
 * ignores the convergence exit (always max_iter iterations). */
int  vm_optimize_level(vm_pyr *pyr, int lvl, float max_iter, volatile const int *run_flag,
                       int fixed_work, vm_progress *out);
/* Morph::calculate_halfway_parametrization, Algorithm/morph.cu:150-168: the
 * whole coarse-to-fine solve.  per_level (may be NULL) receives nlevels-1
 * entries, index = level. */
int  vm_solve(vm_pyr *pyr, float max_iter, float max_iter_drop_factor,
              const vm_constraint *c, int n, volatile const int *run_flag,
              int fixed_work, vm_progress *per_level);
/* The same two calls for a BATCH of frame pairs that share one context and one geometry
 * (no constraints): every sweep launch covers all pairs, which is how levels too small to
 * occupy the GPU are filled -- the natural parallelism of the path is across pairs.
 * out / per_level: n entries resp. n*(nlevels-1) entries, pair-major; elapsed_ms and
 * launches are those of the batch.  Each pair converges on its own flags. */
int  vm_optimize_level_batch(vm_pyr **pyrs, int n, int lvl, float max_iter,
                             volatile const int *run_flag, int fixed_work, vm_progress *out);
int  vm_solve_batch(vm_pyr **pyrs, int n, float max_iter, float max_iter_drop_factor,
                    volatile const int *run_flag, int fixed_work, vm_progress *per_level);
/* the same with every pair's own user constraints (Parameters::lp/rp/cnt of its frame, resolved as for
 * vm_solve: Algorithm/morph.cu:345-388, 471-505): cons[i] / ncons[i], cons == NULL = none anywhere */
int  vm_solve_batch_cons(vm_pyr **pyrs, int n, float max_iter, float max_iter_drop_factor,
                         const vm_constraint *const *cons, const int *ncons,
                         volatile const int *run_flag, int fixed_work, vm_progress *per_level);
/* CMatchingThread::update_result + Resize, Algorithm/MatchingThread.cpp:22-100:
 * v of level `lvl` scaled by (W0/W, H0/H) and bilinearly resized to w0 x h0 */
int  vm_upscale_result(vm_pyr *pyr, int lvl, int w0, int h0, float *v_xy_out, int pitch);

/* ---- video pairs: the temporal coherence path ------------------------------ */
/* The reference couples the frames of a video: class Pyramid with
 * depth > 1 -- per level `depth` pages (Algorithm/Pyramid.h:52-98) and four optical-flow
 * fields per page (f0/f1: forward flow of video 0/1 from frame t to t+1, b0/b1: backward,
 * Pyramid.h:85-90) -- and Morph::optimize_level solves the middle page first, then two
 * chains outward, each page tied to its solved neighbour by w_temp |v - ref|_1
 * (Algorithm/morph.cu:752-759, 1394-1439), ref being the neighbour's field advected along
 * the flows (initialize_temp, Algorithm/upsample.cu:214-258).  vm_video is that object; a
 * page is the same device state as a frame pair and is swept by the same kernels. */
typedef struct vm_video vm_video;
/* a connected point pair with the frame it belongs to (Conp::p.z, parameters.h:22-26) */
typedef struct {
    float lx, ly, rx, ry;
    float weight;
    int   frame;
} vm_video_constraint;
/* Pyramid::append_new per level with its depth (pyramid.cu:220-240, 462-477).  Level 0 is
 * the finest; the last level holds only v.  d[l] = pages of level l (not growing with l);
 * factor_t[l] (may be NULL: 2 where the depth shrank) = temporal stride level l was built
 * with (pyramid.cu:468); depth0 = frames of the video (the placeholder level's depth). */
int  vm_video_create(vm_ctx *ctx, int nlevels, const int *w, const int *h, const int *d,
                     const int *factor_t, int depth0, vm_video **out);
void vm_video_destroy(vm_video *v);
int  vm_video_levels(vm_video *v);
int  vm_video_level_dims(vm_video *v, int lvl, int *w, int *h, int *depth, float *factor_d);
/* the uploads of Pyramid::build for one page: lumas (pyramid.cu:275-280) and the four flow
 * fields, tight or pitched (h, w, 2) floats (pyramid.cu:323-326, 452-455); NULL = keep */
int  vm_video_upload_luma(vm_video *v, int lvl, int page, const float *img0, const float *img1, int pitch);
int  vm_video_upload_flows(vm_video *v, int lvl, int page, const float *f0, const float *f1,
                           const float *b0, const float *b1, int pitch);
/* Pyramid::build(video0, video1, f0, f1, b0, b1, start_res), pyramid.cu:166-485, on the
 * device: the image half for one frame (its luma pyramid goes to every page that shows it),
 * and the flow half for the whole video (scale, x size ratio, temporal concatenation where
 * the depth halves): f0[t].. = tight (h0, w0, 2) float arrays of the depth0 frames */
int  vm_video_build_rgb(vm_video *v, int frame, const uint8_t *rgb0, const uint8_t *rgb1, int pitch_bytes);
int  vm_video_build_flows(vm_video *v, const float *const *f0, const float *const *f1,
                          const float *const *b0, const float *const *b1);
int  vm_video_set_v(vm_video *v, int lvl, int page, const float *v_xy, int pitch);
int  vm_video_get_v(vm_video *v, int lvl, int page, float *v_xy, int pitch);
int  vm_video_get_field(vm_video *v, int lvl, int page, int field, void *host);
/* CMatchingThread::update_result for depth > 1, MatchingThread.cpp:22-84: every page of level
 * `lvl` scaled by (w0 / w, h0 / h) and resized to w0 x h0 into frame min(page * factor,
 * depth0 - 1), factor = factor_d[placeholder] / factor_d[lvl]; the frames the temporal
 * pyramid skipped are blended linearly from the two frames around them.  vm_video_result
 * delivers all depth0 frames (tight (depth0, h0, w0, 2) floats).  A level whose last page
 * stops short of the last frame ((depth - 1) * factor < depth0 - 1: never with the reference's
 * own depth tables) is refused with VM_E_STATE -- the reference would leave those frames as an
 * earlier delivery wrote them; vm_frame_set_v_from_video leaves ONE of them in a compositor frame's v, on the
 * device (w0 x h0 = the frame's size, depth0 = the depth the video was created with). */
int  vm_video_result(vm_video *v, int lvl, int w0, int h0, float *v_xy_frames);
/* Morph::cpu_optimize_level for every page of the coarsest level, morph.cu:419-590 */
int  vm_video_coarse_solve(vm_video *v, const vm_video_constraint *c, int n);
/* upsample(pyr[dst], pyr[dst+1]), upsample.cu:260-340: the spatial upsample of every coarse
 * page and, where the depth doubles, the in-between pages splatted from their two
 * neighbours along the flows, smoothed and hole-filled (temp_ref, interpolate_temp_ref,
 * smooth, fill_zeros_x; fill_zeros_y has no effect on the result) */
int  vm_video_upsample(vm_video *v, int dst_lvl);
/* Morph::initialize_level for every page, morph.cu:264-390 */
int  vm_video_init_level(vm_video *v, int lvl, const vm_video_constraint *c, int n);
/* initialize_temp(lvl, page, dir), upsample.cu:214-258: temp.ref / temp.mask of `page` from
 * page + dir (dir = -1: along its forward flows, +1: backward); the page is then swept
 * with flag == true */
int  vm_video_initialize_temp(vm_video *v, int lvl, int page, int dir);
/* Morph::optimize_level, morph.cu:1353-1441: middle page, then both chains (step k of the two
 * chains shares its launches).  out (may be NULL): depth entries, index = page. */
int  vm_video_optimize_level(vm_video *v, int lvl, float max_iter, volatile const int *run_flag,
                             int fixed_work, vm_progress *out);
/* Morph::calculate_halfway_parametrization, morph.cu:150-168.  per_page (may be NULL): for
 * every level with images (finest first) its depth entries. */
int  vm_video_solve(vm_video *v, float max_iter, float max_iter_drop_factor,
                    const vm_video_constraint *c, int n, volatile const int *run_flag,
                    int fixed_work, vm_progress *per_page);

/* ---- compositor ---------------------------------------------------------- */
typedef struct vm_frame vm_frame;
/* device-resident inputs of one output frame: the two Poisson-extended RGBA8
 * canvases (w+2ex)x(h+2ex) (Pyramid::_extends1/_extends2, Pyramid.h:44-45),
 * the full-resolution halfway field and quadratic path (Pyramid::_vector,
 * _qpath).  Replaces the per-frame cudaMallocArray/H2D uploads of
 * RenderWidget::RenderStage2, UI/RenderWidget.cpp:229-266.  qpath may be NULL
 * (zero path: CQuadraticPath is disabled in the reference app). */
int  vm_frame_create(vm_ctx *ctx, int w, int h, int ex, vm_frame **out);
void vm_frame_destroy(vm_frame *f);
int  vm_frame_upload(vm_frame *f, const uint8_t *ext0_rgba, const uint8_t *ext1_rgba,
                     const float *v_xy, const float *qpath_xy);
/* the same from the two RGB8 frames themselves (h rows of pitch_bytes, 0 = tight): the canvases are built on the device as
 * Pyramid::build builds them, Algorithm/pyramid.cu:186-200 (frame + zero alpha plane pasted at (ex, ex) into a canvas of
 * (255, 255, 255, 255)); v and the quadratic path stay what they are.  12 MB over the link per 1080p pair instead of 27. */
int  vm_frame_upload_rgb(vm_frame *f, const uint8_t *rgb0, const uint8_t *rgb1, int pitch_bytes);
int  vm_frame_download_ext(vm_frame *f, int side, uint8_t *ext_rgba);
/* take v straight from a solved pyramid level (device to device, with the
 * update_result upscale) */
int  vm_frame_set_v_from_level(vm_frame *f, vm_pyr *pyr, int lvl);
/* Page-lock / release a host buffer the caller uploads frames from (the reference re-uploads 108 MB of float4
 * canvases per rendered frame from pageable memory, UI/RenderWidget.cpp:229-266; here 27 MB of RGBA8 per frame
 * pair, once): vm_frame_upload from a registered buffer runs at the link's rate.  Optional. */
int  vm_host_register(void *ptr, uint64_t bytes);
int  vm_host_unregister(void *ptr);
/* ... and frame `frame` of vm_video_result (above), computed on the device */
int  vm_frame_set_v_from_video(vm_frame *f, vm_video *v, int lvl, int frame);
/* render_halfway_image, Algorithm/render.cu:62-96 (UI/RenderWidget.h:52-57);
 * rgb_out: h rows of w RGB8 pixels, pitch in bytes (0 = tight) */
int  vm_render_halfway(vm_frame *f, float color_fa, float geo_fa, int color_from,
                       uint8_t *rgb_out, int pitch_bytes);
/* same, output left on the device (for timing / chaining) */
int  vm_render_halfway_dev(vm_frame *f, float color_fa, float geo_fa, int color_from,
                           float *elapsed_ms);
/* CPoissonExt::prepare + poissonExtend for one side (1 or 2) of the frame,
 * Algorithm/PoissonExt.cpp:49-362, on the device-resident canvases: matrix-free
 * multigrid-preconditioned CG instead of MKL DSS.  iters/rel_res may be NULL. */
int  vm_poisson_extend(vm_frame *f, int side, float tol, int max_it,
                       int *iters, float *rel_res, float *elapsed_ms);
/* The body of CPoissonExt::run's frame loop (Algorithm/PoissonExt.cpp:24-36: prepare + poissonExtend of
 * side 1, then of side 2) for n frames of one context and one size AT ONCE: side 1 samples the original
 * image 2 and side 2 the original image 1 (the copies taken at :26-27, here at vm_frame_upload), so the
 * 2 n systems are independent and share every kernel launch (blockIdx.z = system).  Same results per system
 * as vm_poisson_extend.  iters / rel_res (may be NULL): 2 n entries, [2 i] = side 1 of frames[i], [2 i + 1] =
 * side 2.  n <= 32. */
int  vm_poisson_extend_frames(vm_frame *const *frames, int n, float tol, int max_it,
                              int *iters, float *rel_res, float *elapsed_ms);
/* Diagnostic of the solver behind vm_poisson_extend* / vm_frame_quadratic_path (the reference: MKL DSS, Algorithm/PoissonExt.cpp:321-329,
 * timed by clock() around CPoissonExt::run, :21, 38-39): on != 0 arms a probe that brackets every launch of the solver's
 * dominant kernel -- the one that carries the PCG update x += alpha p, r -= alpha q, r.r: the level-0 restriction of the multigrid
 * cycle with the update fused in (76 bytes per unknown), or k_mgb_update by itself (a pure stream of 73 bytes per unknown) where a
 * system is too small to have a level-0 restriction -- with HIP events on the context's stream; on == 0 disarms it and returns the summed
 * event time (microseconds), the number of launches, the number of active systems summed over them and how many of the launches
 * were of the fused form.  Any of the four may be NULL.  bench.py's `poisson_extend_1080p_ex192.roofline.dominant_kernel`. */
int  vm_dbg_poisson_profile(vm_ctx *ctx, int on, double *update_us, int *update_launches, double *active_systems, int *fused_launches);
/* CQuadraticPath::optimize for one frame, Algorithm/QuadraticPath.cpp:24-223
 * (QuadraticPath.h:14-24): from the frame's halfway field v the per-pixel optimal
 * Jacobian blend and the Neumann Poisson solve for the quadratic motion path u,
 * which stays in the frame where vm_render_halfway reads it (the reference's
 * `_qpath`).  Multigrid-preconditioned CG to the relative residual `tol` instead
 * of 10 001 plain CG iterations; the zero-mean solution.  float32 attains about
 * 1e-4 .. 1e-5 here (|u| >> |right-hand side|; a solved, rounding-rough field: 1e-4): the best iterate
 * is delivered, VM_E_NUMERIC if its residual exceeds `tol` -- or if v folds over so that
 * the two Jacobians' columns are anti-parallel somewhere (0/0 in QuadraticPath.cpp:96-101:
 * the reference propagates the NaN).  iters/rel_res may be NULL. */
int  vm_frame_quadratic_path(vm_frame *f, float tol, int max_it,
                             int *iters, float *rel_res, float *elapsed_ms);
/* the frame's quadratic path (Pyramid::_qpath, Pyramid.h:42), tight (h, w, 2) floats */
int  vm_frame_download_qpath(vm_frame *f, float *u_xy);
/* the frame's v (set by vm_frame_upload / _set_v_from_level / _set_v_from_video), tight (h, w, 2) floats */
int  vm_frame_download_v(vm_frame *f, float *v_xy);

/* ---- synchronisation stage ("(later)" row) ------------------ */
/* Before the morph the reference's app aligns the two videos in time: the user
 * ties points of video 0 to points of video 1 -- possibly on DIFFERENT frames --
 * and CSyncThread (Algorithm/SyncThread.h:7-39, SyncThread.cpp) solves for a smooth
 * field (x, y, frame displacement) per voxel of a decimated (x, y, t) pyramid:
 * per level three conjugate-gradient solves of one thin-plate + constraint system
 * (optimize_level, SyncThread.cpp:290-480), level to level by Kernel_upsample
 * (upsample.cu:343-375).  render_resample_image (render.cu:99-246) then re-times
 * both videos with the field and the forward optical flows.  vm_sync is
 * CSyncThread's state plus the four layered arrays Pyramid::build(video0, video1,
 * f0, f1, start_res) keeps for that renderer (pyramid.cu:57-141). */
typedef struct vm_sync vm_sync;
/* a Connect between Conp lp[li] and rp[ri] (parameters.h:16-26): full-resolution
 * pixel and frame on either side, integers as in the reference */
typedef struct {
    int lx, ly, lz, rx, ry, rz;
} vm_sync_constraint;
typedef struct {
    int    iters;          /* passes of the CG loop: floor(max_iter) + 1 (`k <= _max_iter`)  */
    int    launches;
    double voxel_iters;    /* iters * W * H * D  (_current_iter += N, SyncThread.cpp:463)   */
    float  elapsed_ms;     /* HIP-event time of the level                                   */
    float  resid[3];       /* r . r of the x, y, z systems when the loop ended              */
} vm_sync_progress;
/* level geometry of Pyramid::build(video0, video1, f0, f1, start_res), pyramid.cu:143-163:
 * entry 0 is the full-resolution placeholder, entries 1.. are solved (finest first);
 * writes at most `cap` entries, *n_out = the number of levels */
int  vm_sync_level_table(int w, int h, int d, int start_res, int *lw, int *lh, int *ld, int cap, int *n_out);
/* nlevels entries (placeholder included), uses the context's w_ui / w_tps (vm_set_params) */
int  vm_sync_create(vm_ctx *ctx, int nlevels, const int *w, const int *h, const int *d, vm_sync **out);
void vm_sync_destroy(vm_sync *s);
int  vm_sync_set_constraints(vm_sync *s, const vm_sync_constraint *c, int n);
/* CSyncThread::load_identity / upsample_level (SyncThread.cpp:86-128): allocate level lvl's
 * field, zero or upsampled from lvl + 1 (which is released, as there) */
int  vm_sync_load_identity(vm_sync *s, int lvl);
int  vm_sync_upsample_level(vm_sync *s, int lvl);
/* CSyncThread::optimize_level(el): the loop runs while k <= max_iter and *run_flag (polled every
 * 64 iterations; NULL = never cancelled) */
int  vm_sync_optimize_level(vm_sync *s, int lvl, float max_iter, volatile const int *run_flag,
                            vm_sync_progress *out);
/* CSyncThread::run (SyncThread.cpp:58-84): all levels, coarsest first, max_iter * 10 halved per
 * level; out (optional) has nlevels - 1 entries, [lvl - 1] for level lvl */
int  vm_sync_solve(vm_sync *s, float max_iter, volatile const int *run_flag, vm_sync_progress *out);
/* d_x[lvl], d_y[lvl], d_z[lvl]: tight (d, h, w) float arrays (any may be NULL) */
int  vm_sync_get_field(vm_sync *s, int lvl, float *x, float *y, float *z);
int  vm_sync_set_field(vm_sync *s, int lvl, const float *x, const float *y, const float *z);
/* CSyncThread::update_result (SyncThread.cpp:482-521) for one frame: Pyramid::_vector[frame],
 * (h0, w0) float4 = (X ratio_x, Y ratio_y, Z, 0) resized to full resolution; vec4 may be NULL
 * (the frame's field is then only refreshed on the device, where vm_sync_render reads it) */
int  vm_sync_result(vm_sync *s, int lvl, int frame, float *vec4);
/* the layered arrays of Pyramid::build (pyramid.cu:93-141): frame `frame` of video `side`
 * (RGBA8, alpha ignored) and of its forward flow (float2) */
int  vm_sync_upload_frame(vm_sync *s, int side, int frame, const uint8_t *rgba, int pitch_bytes);
int  vm_sync_upload_flow(vm_sync *s, int side, int frame, const float *flow_xy, int pitch_floats);
/* render_resample_image(out, ..., fa, frame, ...) (render.cu:203-246; caller RenderWidget::
 * RenderStage1, UI/RenderWidget.cpp:205-227) with the field vm_sync_result last produced for
 * this frame (produced from level 1 if none was); rgb_out: h0 rows of pitch_bytes, RGB8 */
int  vm_sync_render(vm_sync *s, float fa, int frame, uint8_t *rgb_out, int pitch_bytes);
/* the same, result left on the device (timing without the download) */
int  vm_sync_render_dev(vm_sync *s, float fa, int frame, float *elapsed_ms);

/* ---- multi-GPU ----------------------------------------------------------- */
/* The shared parameter block every rank needs (KernParameters + iteration
 * control + constraints), flattened so that any transport -- the RCCL
 * broadcast of bench.py / the C++ driver -- can ship it. */
typedef struct {
    vm_kern_params kp;
    float max_iter, max_iter_drop_factor;
    int   start_res, math_mode, n_constraints;
} vm_param_block;
/* in-place RCCL broadcast of `bytes` at device pointer `dev_buf` from rank
 * `root` on communicator `nccl_comm` (an ncclComm_t), on the context's stream */
int  vm_rccl_bcast(vm_ctx *ctx, void *nccl_comm, void *dev_buf, uint64_t bytes, int root);
/* One PROCESS driving n devices (SURVEY 7 step 9, 8(b) `vm_bcast_params`; the reference is single-GPU,
 * UI/MdiEditor.cpp:54-75 picks one device): ncclCommInitAll over `devices` (n distinct ordinals) -> comms_out[n];
 * vm_rccl_comm_destroy frees one. */
int  vm_rccl_comm_init_all(int n, const int *devices, void **comms_out);
void vm_rccl_comm_destroy(void *nccl_comm);
/* The shared parameter block from ctxs[root] to all n contexts over RCCL (one ncclBroadcast per communicator
 * inside one ncclGroup, each on its context's stream); every context then sets its kernel parameters and
 * arithmetic mode from the copy IT received.  blocks_out: n entries (may be NULL), the block as received.
 * comms == NULL is a TEST MODE for contexts sharing one device (RCCL wants one rank per device): the copies go
 * device-to-device on that device instead. */
int  vm_bcast_params(vm_ctx *const *ctxs, void *const *nccl_comms, int n, int root,
                     const vm_param_block *blk, vm_param_block *blocks_out);
/* The same one broadcast for a payload of the caller's length -- BASELINE config[4]: the parameter block followed by the
 * frames' point constraints (Parameters::lp / rp / cnt resolved to vm_constraint rows, parameters.h:16-26, consumed at
 * Algorithm/morph.cu:345-388, 471-505), the layout videomorphing_amd/dist.py:pack_block ships between processes.  `src`
 * (host, `bytes` long) goes from ctxs[root] to all n contexts over RCCL (nccl_comms == NULL: the one-device test mode
 * of vm_bcast_params); dst_host[i] (host, `bytes` each) receives what context i got.  Nothing is interpreted. */
int  vm_bcast_bytes(vm_ctx *const *ctxs, void *const *nccl_comms, int n, int root, const void *src, uint64_t bytes,
                    void *const *dst_host);

#ifdef __cplusplus
}
#endif
#endif
