// vm_internal.h -- shared between the C-ABI implementation and the HIP kernels.
#ifndef VM_INTERNAL_H
#define VM_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vmorph.h"

// Tile geometry of the sweep schedule.  It is part of the algorithm's
// definition (which pixels are relaxed together decides the result), so it is
// kept equal to the reference's kernel_optimize_level (morph.cu:594-598,
// 1291-1292): 64x16 pixel tiles on a 69x21 pitch, four offset passes.
#define VM_TILE_W 64
#define VM_TILE_H 16
#define VM_PITCH_X 69
#define VM_PITCH_Y 21
#define VM_HALO_W (VM_TILE_W + 4)
#define VM_HALO_H (VM_TILE_H + 4)
#define VM_NCELL (VM_HALO_W * VM_HALO_H)

// Device view of one level (the role of KernPyramidLevel, Pyramid.h:100-158).
// All per-pixel arrays share the row stride `rs` (elements), a multiple of 32
// so that every row of every array starts on a 128-byte line.
struct VmLevelView {
    int w, h, rs;
    float inv_wh;
    int imp_rs, imp_rows;
    const float *img0, *img1;
    float2 *v, *luma, *mean, *var, *tps_b, *ui_b;
    float *cross, *value, *ui_axy;
    uint32_t *impmask;
    // decision records of the SPLIT / STEP sweep schedules (vm_sweep_kernels.hip)
    uint32_t *rec_tag;
    float4 *rec_a, *rec_b;
    // STEP writes the records of odd epochs here: a pixel can be decided in two consecutive
    // phases (last phase of a pass, first of the next), and a step reads the previous
    // phase's records while it writes its own
    uint32_t *rec_tag2;
    float4 *rec_a2, *rec_b2;
    // second copy of the window sums and the mask for the STEP schedule's ping-pong
    float2 *mean2, *var2, *tps_b2;
    float *cross2, *value2;
    uint32_t *impmask2;
    // temporal coherence term of a video page (energy_change with flag == true,
    // morph.cu:752-759): the halfway field advected from the neighbouring page
    // (lvl.temp.ref), its splat weight (lvl.temp.mask) and the level's factor_d.
    // temp_mask == nullptr <=> flag == false (the middle page, and every frame pair
    // solved on its own).
    const float2 *temp_ref;
    const float *temp_mask;
    float factor_d;
    // SPARSE schedule workspace: two lists of non-zero mask word indices (ping-pong, nwords
    // entries each), their lengths, and an epoch stamp per word
    uint32_t *sp_wl, *sp_cnt, *sp_stamp;
};

struct VmKParams {
    float w_ui, w_tps, w_ssim, ssim_clamp, eps;
    int bcond;
    float w_temp;
    int commit_order; // EXACT only, diagnostic (vm_set_commit_order): bit 0 = a phase's commits in reversed order, bit 1 = column-major
};

// per-iteration activity counters of the sweep kernels (uint32 words per iteration):
// [0] tile visits that were not skipped (TILE), [1] line searches, [2] commits,
// [3] tile-phases with records (SPLIT / STEP), [4] energy evaluations
#define VM_STAT_WORDS 8
// the listed form of a pruned TILE pass (vm_sweep_kernels.hip, k_tile_scan): workgroups of the sweep
#define VM_TILE_LIST_GRID 1024 // (what the chip holds of the lean kernel at once: a longer list is walked in turns)

// constant tables living in one device buffer: 625 floats (thin-plate stencil
// per border class) followed by 225 uint32 (improving-mask window bits)
#define VM_TAB_TPS 0
#define VM_TAB_IMP 625
#define VM_TAB_WORDS (625 + 225)

// PASS schedule: barrier words per tile group and launch (one 64-bit arrival counter with the
// XCD census + 32 per-workgroup flag words, on lines of their own)
#define VM_PASS_SYNC_WORDS 64

// launchers implemented once per arithmetic mode (vm_morph_kernels.hip is
// compiled twice: -DVM_EXACT=1 -ffp-contract=off and -DVM_EXACT=0)
#define VM_DECL_LAUNCHERS(SUFFIX)                                                             \
    void vm_launch_init_level_##SUFFIX(const VmLevelView &L, float ssim_clamp,                \
                                       const uint32_t *tables, hipStream_t s);                \
    void vm_launch_optimize_##SUFFIX(const VmLevelView *views, int nbatch, int cap, int w, int h, \
                                     const VmKParams &P, const uint32_t *tables, int offx,    \
                                     int offy, uint32_t *flags, uint32_t *stats, int iter_idx,\
                                     int fixed_work, int threads, const int *iter_dev,        \
                                     int dense, uint32_t *tile_list, hipStream_t s);          \
    void vm_launch_next_iter_##SUFFIX(int *iter_dev, int set, int value, hipStream_t s);      \
    void vm_launch_optimize_sparse_##SUFFIX(const VmLevelView *views, int nbatch, int cap, int w, \
                                            int h, const VmKParams &P, const uint32_t *tables, \
                                            uint32_t *flags, uint32_t *stats, int it0, int nit, \
                                            int fixed_work, int threads, int dense, int lds_cap,   \
                                            int res_mode, hipStream_t s);                      \
    void vm_launch_optimize_split_##SUFFIX(const VmLevelView *views, int nbatch, int cap, int w, \
                                           int h, const VmKParams &P, const uint32_t *tables, \
                                           int offx, int offy, int pass, uint32_t *flags,     \
                                           uint32_t *stats, int iter_idx, int fixed_work,     \
                                           int threads, int parts, hipStream_t s);            \
    void vm_launch_optimize_step_##SUFFIX(const VmLevelView *views, int nbatch, int cap, int w,  \
                                          int h, const VmKParams &P, const uint32_t *tables,  \
                                          int offx, int offy, int pi, int pj, uint32_t epoch, \
                                          uint32_t prev_epoch, int src, int decide,           \
                                          uint32_t *flags, uint32_t *stats, int iter_idx,     \
                                          int fixed_work, int threads, int parts,             \
                                          uint32_t *slots_cur, const uint32_t *slots_prev,    \
                                          int prev_iter_idx, hipStream_t s);                  \
    void vm_launch_optimize_pass_##SUFFIX(const VmLevelView *views, int nbatch, int cap, int w,  \
                                          int h, const VmKParams &P, const uint32_t *tables,  \
                                          int offx, int offy, uint32_t epoch0, uint32_t *bar, \
                                          uint32_t *flags, uint32_t *stats, int iter_idx,     \
                                          int fixed_work, uint32_t *slots_cur,                \
                                          const uint32_t *slots_prev, int prev_iter_idx,      \
                                          uint32_t *err, uint32_t *dbg, int decide,           \
                                          int force_wt, hipStream_t s);                       \
    int vm_pass_resident_blocks_##SUFFIX(int device);                                         \
    void vm_launch_upsample_##SUFFIX(float2 *dst, int dw, int dh, int drs, const float2 *src, \
                                     int sw, int sh, int srs, hipStream_t s);                 \
    void vm_launch_splat_##SUFFIX(const VmLevelView &L, int w0, int h0,                       \
                                  const vm_constraint *dev_c, int n, hipStream_t s);

VM_DECL_LAUNCHERS(exact)
VM_DECL_LAUNCHERS(fast)
VM_DECL_LAUNCHERS(exactf) // sweeps only (vm_sweep_kernels.hip with -DVM_EXACT=2 -ffp-contract=fast): VM_MATH_EXACT_FMA
VM_DECL_LAUNCHERS(reffm)  // sweeps only (-DVM_EXACT=3 -ffp-contract=fast): VM_MATH_REF_FASTMATH
VM_DECL_LAUNCHERS(tex8)   // sweeps, init_level, upsample (-DVM_EXACT=4 -ffp-contract=off): VM_MATH_REF_TEX8
VM_DECL_LAUNCHERS(tex8t)  // the same with truncated weights (-DVM_EXACT=5): VM_MATH_REF_TEX8_TRUNC

// compositor / result delivery (single arithmetic mode)
void vm_launch_upscale(float2 *dst, int w0, int h0, int dpitch, const float2 *v, int w, int h,
                       int rs, hipStream_t s);
void vm_launch_blend_v(float2 *dst, int dpitch, const float2 *a, const float2 *b, int spitch, int w0, int h0,
                       float alpha, float beta, hipStream_t s);
void vm_launch_render(uint8_t *out, int out_pitch, int w, int h, int rs, int ex, float color_fa,
                      float geo_fa, int color_from, const uchar4 *ext0, const uchar4 *ext1,
                      const float2 *v, const float2 *u, hipStream_t s);

#endif
