// vm_sweep_kernels.hip -- the sweep of the halfway optimizer for gfx950:
// kernel_optimize_level and its device helpers, Algorithm/morph.cu:592-1345.
//
// Compiled twice (see vm_morph_common.h): EXACT is bit-identical to the CPU
// oracle, FAST is the production arithmetic.
//
// What is computed is fixed by the reference (it defines the result): 64x16 tiles
// on a 69x21 pitch, four offset passes per iteration, four Jacobi phases per tile;
// per candidate pixel an improving-mask test, a finite-difference gradient, the
// fold-over bound, a golden-section line search and an accept-if-lower commit.
//
// How it is computed is CDNA4-first.  Five schedules share the device functions and
// the state in HBM (the host may switch between them from batch to batch):
//
//  TILE schedule (k_optimize, one launch per pass): one workgroup of T threads
//  owns one tile; window sums, SSIM values, tps.b and the tile's improving-mask
//  words live in LDS for the four phases.  Per phase the candidates are compacted;
//  a DENSE phase gives each a fixed, compile-time number of consecutive lanes (2 in the
//  256-VGPR kernel, 4 in the 128-VGPR one) that split the 25 window neighbours, keep
//  their share of the sums in registers for the whole line search, share the pixel's two
//  bilinear taps and combine SSIM terms with DPP permutes; a SPARSE phase (<= T/32
//  candidates) runs the LEAN line search: 32 lanes per pixel, one neighbour per lane, the bilinear taps spread over
//  quads, a branch-free golden-section loop (~105 instructions per evaluation --
//  in this regime the instruction count of one evaluation IS the time).  Tiles
//  without a set mask bit return after one 96-word load, so a pruned level costs
//  almost nothing -- this is the schedule of large levels.
//
//  SPLIT schedule (k_decide + k_commit, two launches per phase): for levels with
//  too few tiles to fill 256 CUs, a tile's candidates are divided over `parts`
//  workgroups (every pixel gets 32 lanes), decisions go to per-pixel records in
//  HBM, and a second kernel folds them into the window sums.  The reference the
//  STEP schedule is tested against.
//
//  STEP schedule (k_step, one launch per phase): the commit of phase s-1 is
//  folded into the launch of phase s from a second copy of the sums (ping-pong),
//  see the comment at k_step.  Bit-identical to SPLIT.
//
//  SPARSE schedule (k_sparse, one launch per batch of iterations of a pruned level): one
//  workgroup per pair walks the iterations, sweeping only the tiles whose mask words are set;
//  while the set bits fit one tile (FAST) the window sums around them stay in LDS across passes
//  and iterations (resident visits, see the comment at SvTile).
//
//  PASS schedule (k_pass, one launch per pass): a tile = a group of 32 workgroups on one XCD,
//  the four phases behind a tile-local barrier; see the comment at PassLds.
//
// Commits are applied by a per-cell gather of the committed pixels' records in a
// fixed order: deterministic, one owner per cell, no float atomics (the reference
// uses 75 shared + 25 global float atomics per accepted pixel, morph.cu:951-1015).
#include <algorithm>
#include <type_traits>
#include "vm_morph_common.h"

// Pointers in the level views come from memory, so the compiler knows them as generic and emits
// flat_ loads; global-address-space pointers give global_ instructions (vmcnt only, and the
// SGPR-base + 32-bit-offset addressing form).
// Loads and stores of data that workgroups of ONE launch hand to each other (PASS schedule):
// relaxed agent-scope atomics on GLOBAL-address-space pointers = global_load / global_store
// ... sc1 (L1-bypassing loads, write-through stores).  Through generic pointers the compiler
// emits flat_ instructions, which the hand-off rules exclude (MI355X_MICROARCH.md, Valid forms).
typedef __attribute__((address_space(1))) const uint32_t vm_g_cu32;
typedef __attribute__((address_space(1))) uint32_t vm_g_u32;
typedef __attribute__((address_space(1))) const unsigned long long vm_g_cu64;
typedef __attribute__((address_space(1))) unsigned long long vm_g_u64;
typedef __attribute__((address_space(1))) const float vm_g_cf32;

// a * b + c for operands below 2^23 (pixel coordinates, row strides): v_mad_i32_i24, full rate and
// 32 bits wide (the compiler's choice for a 32-bit a * b + c is v_mad_u64_u32 with an undefined high
// half of c -- a register that may be the destination of a load in flight)
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

#ifdef VM_PROF
// dev-only stage stamps of k_decide (10 ns ticks), wave 0 lane 0 of each workgroup
__device__ unsigned long long vm_prof_buf[512 * 16 * 2];
#define VM_PTSF(k)                                             \
    if (tid == 0 && b < 256)                                   \
    vm_prof_buf[8192 + 512 + b * 8 + (k)] = wall_clock64()
#define VM_PTS(ph, k)                                          \
    if (tid == 0 && b < 256 && (ph) < 4)                       \
    vm_prof_buf[b * 32 + (ph) * 8 + (k)] = wall_clock64()
// ... and of tile_sweep (TILE schedule): thread 0 of tiles 0..255 of the first pair
#define VM_TTS(ph, k)                                                              \
    if (threadIdx.x == 0 && blockIdx.x < 256 && blockIdx.z == 0)                   \
    vm_prof_buf[blockIdx.x * 32 + (ph) * 8 + (k)] = wall_clock64()
#define VM_TTSF(k)                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < 256 && blockIdx.z == 0)                   \
    vm_prof_buf[8192 + 512 + blockIdx.x * 8 + (k)] = wall_clock64()
#define VM_TS(i) ts[i] = wall_clock64()
#define VM_TS_ARG , unsigned long long *ts
#define VM_TS_PASS , ts
#else
#define VM_TS(i)
#define VM_TS_ARG
#define VM_TS_PASS
#define VM_PTS(ph, k)
#define VM_PTSF(k)
#define VM_TTS(ph, k)
#define VM_TTSF(k)
#endif

namespace {

struct TileLds {
    float2 mean[VM_NCELL], var[VM_NCELL], tpsb[VM_NCELL];
    float cross[VM_NCELL], value[VM_NCELL];
    // per phase pixel (slot = (y>>1)*32 + (x>>1) inside the tile)
    float2 d_mean[256], d_var[256], d_step[256];
    float d_cross[256];
    int d_ok[256];           // 0: untouched, 1: commit, 2: mask hit that did not move
    int list[256];           // compacted slots
    int wave_cnt[4];
    uint32_t cbits[8];       // commits of the phase: bit tx of word ty (slot = ty*32 + tx)
    float tps[625];
    uint32_t imp[225];
    uint32_t mask[6][16];    // improving-mask words covering the tile +-1 block
    uint32_t n_eval;         // energy evaluations of the workgroup (one count per candidate)
    uint32_t dirty[(VM_NCELL + 31) / 32]; // lean kernels: cells a commit of this visit reached (only those are written back)
};

// the LDS of the SPLIT kernels: no window sums
struct SplitLds {
    float2 d_mean[256], d_var[256], d_step[256];
    float d_cross[256];
    int d_ok[256];
    int list[256];
    int wave_cnt[4];
    float tps[625];
    uint32_t imp[225];
    uint32_t mask[6][16];
    uint32_t n_eval;
};

struct PixelCtx {
    int px, py;
    int idx;         // global element index
    float2 v, old_luma;
    float tps_axy, ui_axy;
    float2 tps_b, ui_b;
    // temporal term (flag == true only): reference vector and splat weight of the pixel
    float2 tref;
    float tmask;
};

// v_temp of energy_change (morph.cu:752-756): the change of |v - ref|_1 under the move d
__device__ __forceinline__ float temp_change(const PixelCtx &c, float dx, float dy)
{
    float v_temp = 0.0f;
    v_temp += fabsf(c.v.x + dx - c.tref.x) - fabsf(c.v.x - c.tref.x);
    v_temp += fabsf(c.v.y + dy - c.tref.y) - fabsf(c.v.y - c.tref.y);
    return v_temp;
}

// where the 5x5 window sums of a pixel come from
struct LdsSrc {
    const TileLds *S;
    int hc; // LDS cell of (px-2, py-2)
    __device__ __forceinline__ void load(int i, int j, float2 &m, float2 &q, float &cr, float &val) const
    {
        const int c = hc + i * VM_HALO_W + j;
        m = S->mean[c]; q = S->var[c]; cr = S->cross[c]; val = S->value[c];
    }
};
struct GlbSrc {
    const VmLevelView *L;
    int g0; // global index of (px-2, py-2)
    __device__ __forceinline__ void load(int i, int j, float2 &m, float2 &q, float &cr, float &val) const
    {
        const int g = g0 + i * L->rs + j;
        m = L->mean[g]; q = L->var[g]; cr = L->cross[g]; val = L->value[g];
    }
};

#if VM_EXACT
// ---- EXACT: literal ssim_change (morph.cu:671-728) + energy_change (:730-761),
// flag == false; one lane per pixel, neighbours visited in row-major order (dense
// phases; sparse phases and the SPLIT schedule use energy_x32 below: same bits)
#define VM_SWEEP_T 1024
#define VM_SMAX 1
#define VM_MIN_FANOUT 1
template <int SMAX>
struct NbCacheT {};
template <bool INTERIOR, int SMAX, int LF = 0, class Src>
__device__ __forceinline__ void nb_load(NbCacheT<SMAX> &, const VmLevelView &, const Src &, const PixelCtx &, int, int) {}

template <bool INTERIOR, int SMAX, int LF = 0, class Src>
__device__ __forceinline__ float energy_change(const VmLevelView &L, const VmKParams &P, const Src &src,
                                               const NbCacheT<SMAX> &, const PixelCtx &c, float dx, float dy, int)
{
    const float vx = c.v.x + dx, vy = c.v.y + dy;
    const float lx = tap(L.img0, L.w, L.h, L.rs, c.px - vx + 0.5f, c.py - vy + 0.5f);
    const float ly = tap(L.img1, L.w, L.h, L.rs, c.px + vx + 0.5f, c.py + vy + 0.5f);
    const float dmx = lx - c.old_luma.x, dmy = ly - c.old_luma.y;
    const float dvx = lx * lx - c.old_luma.x * c.old_luma.x;
    const float dvy = ly * ly - c.old_luma.y * c.old_luma.y;
    const float dcross = lx * ly - c.old_luma.x * c.old_luma.y;
    float change = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int qy = c.py + i - 2;
        const int ny = window_count(qy, L.h);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int qx = c.px + j - 2;
            if (qx < 0 || qx >= L.w || qy < 0 || qy >= L.h)
                continue;
            const float counter = (float)(ny * window_count(qx, L.w));
            float2 m, q;
            float cr, val;
            src.load(i, j, m, q, cr, val);
            const float ns = ssim_value(m.x + dmx, m.y + dmy, q.x + dvx, q.y + dvy, cr + dcross, counter,
                                        P.ssim_clamp);
            change += val - ns;
        }
    }
    float v_tps = c.tps_axy * (dx * dx + dy * dy);
    v_tps += c.tps_b.x * dx;
    v_tps += c.tps_b.y * dy;
    float v_ui = c.ui_axy * (dx * dx + dy * dy);
    v_ui += c.ui_b.x * dx;
    v_ui += c.ui_b.y * dy;
    // flag == false: v_temp = 0 and the mask is 0, the term is +0 (morph.cu:752-759)
    const float t = L.temp_mask ? P.w_temp * temp_change(c, dx, dy) * c.tmask * L.factor_d : 0.0f;
    return (P.w_ui * v_ui + P.w_ssim * change + t) * L.inv_wh + P.w_tps * v_tps;
}
#else
// ---- FAST: the same energy, evaluated by L lanes per pixel.  Lane `sub` owns the
// window neighbours sub, sub+L, sub+2L, ... (at most VM_SMAX of them) and keeps
// their sums in registers.  INTERIOR pixels (>= 4 from every border) have 25
// in-image neighbours with a full window each: count, 1/count and validity
// become compile-time constants.
#ifndef VM_SWEEP_T
#define VM_SWEEP_T 512
#endif
#if VM_SWEEP_T >= 1024
#define VM_SMAX 7
#define VM_MIN_FANOUT 4
#else
#define VM_SMAX 13
#define VM_MIN_FANOUT 2
#endif
// The mean of a window whose first-moment sum is `sum`, with the pixel's luma changed by d: (sum + d) / n, the SUM
// rounded first -- exactly the sum a commit of that move leaves in memory (m += d), so that the energy a line search
// predicts is the energy the level then has.  Rounds 1-3 kept pre-divided means and evaluated fma(d, 1/n, mean): a
// finer-grained landscape than the stored one.  A move accepted for a predicted gain below the rounding of the sums
// could then LOSE energy once committed, and the descent lost its monotonicity at the noise floor: on the 120x68
// level of config[1] a patch of border pixels (x = 0..1, y ~ 50-56, window counts 15 and 20) crept 0.2-0.37 px away
// over ~150 iterations, uphill in the oracle's energy (+2-3 % of the level's total) -- 5 px at full resolution, and
// the whole of FAST's excess distance from the family of legal builds (r04: tools/dev_level5_drift.py,
// profiles/r04_notes.md).  Costs one instruction per mean (add + mul instead of one fma).
__device__ __forceinline__ float window_mean(float sum, float d, float inv_n) { return (sum + d) * inv_n; }

// SMAX = neighbours a lane may own: 13 with a fan-out of >= 2 lanes per pixel (the 256-VGPR dense
// kernel), 7 with >= 4 (the 128-VGPR one: two workgroups per CU)
template <int SMAX>
struct NbCacheT {
    float A[SMAX], B[SMAX];                   // raw first-moment sums (window_mean() forms the means)
    float VX[SMAX], VY[SMAX], X[SMAX];        // raw second-moment sums
    float VAL[SMAX];                          // current SSIM value (value - new is summed,
                                              // as the reference does: 1e-3..1e-6 of the values)
    float N[SMAX];                            // border waves only: window count, 0 = no such neighbour
};

__device__ __forceinline__ float dpp_xor1(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_half_mirror(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_mirror(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));
}
__device__ __forceinline__ float swz_xor16(float x)
{
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F));
}
// tap() (vm_morph_common.h) on the image `off` elements behind img0 (both images sit in the level's
// slab): the same arithmetic, the texels as global loads off a uniform base + 32-bit offsets
__device__ __forceinline__ float tap_g(const float *img0, int off, int w, int h, int rs, float x, float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = xb - fi, b = yb - fj;
    fi = fminf(fmaxf(fi, -1.0f), (float)w);
    fj = fminf(fmaxf(fj, -1.0f), (float)h);
    int i0 = (int)fi, j0 = (int)fj;
    int i1 = min(max(i0 + 1, 0), w - 1), j1 = min(max(j0 + 1, 0), h - 1);
    i0 = min(max(i0, 0), w - 1);
    j0 = min(max(j0, 0), h - 1);
    const char *base = (const char *)img0;
    const int r0 = off + j0 * rs, r1 = off + j1 * rs;
    const float t00 = *(vm_g_cf32 *)(base + ((uint32_t)(r0 + i0) << 2)), t10 = *(vm_g_cf32 *)(base + ((uint32_t)(r0 + i1) << 2));
    const float t01 = *(vm_g_cf32 *)(base + ((uint32_t)(r1 + i0) << 2)), t11 = *(vm_g_cf32 *)(base + ((uint32_t)(r1 + i1) << 2));
    return (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
}

// sum over the aligned group of Lf lanes (Lf = 2, 4, 8, 16 or 32, uniform in the
// workgroup); every lane of the group ends with the same bits
__device__ __forceinline__ float group_sum(float x, int Lf)
{
    x += dpp_xor1(x);
    if (Lf >= 4) x += dpp_xor2(x);
    if (Lf >= 8) x += dpp_half_mirror(x);
    if (Lf >= 16) x += dpp_mirror(x);
    if (Lf >= 32) x += swz_xor16(x);
    return x;
}

// LF: the fan-out as a compile-time constant (0: the run-time value Lf_rt).  The generic form carries a
// uniform branch per window slot and per reduction stage; with the fan-out known the loops unroll flat.
template <bool INTERIOR, int SMAX, int LF = 0, class Src>
__device__ __forceinline__ void nb_load(NbCacheT<SMAX> &nb, const VmLevelView &L, const Src &src, const PixelCtx &c,
                                        int sub, int Lf_rt)
{
    const int Lf = LF ? LF : Lf_rt;
#pragma unroll
    for (int j = 0; j < SMAX; ++j) {
        if (j * Lf >= 25) // uniform in the workgroup: no lane has such a neighbour
            break;
        const int k = sub + j * Lf;
        const int i = k / 5, jj = k - i * 5;
        const int qx = c.px + jj - 2, qy = c.py + i - 2;
        const bool ok = k < 25 && (INTERIOR || (qx >= 0 && qx < L.w && qy >= 0 && qy < L.h));
        float2 m, q;
        float cr, val;
        src.load(ok ? i : 2, ok ? jj : 2, m, q, cr, val);
        float n = 25.0f, in = 0.04f;
        if (!INTERIOR) {
            n = ok ? (float)(window_count(qy, L.h) * window_count(qx, L.w)) : 25.0f;
            in = n == 25.0f ? 0.04f : __builtin_amdgcn_rcpf(n);
            nb.N[j] = ok ? n : 0.0f; // 1/count is recomputed per evaluation: registers are scarcer than v_rcp
        }
        (void)in;
        nb.A[j] = m.x;
        nb.B[j] = m.y;
        nb.VX[j] = q.x;
        nb.VY[j] = q.y;
        nb.X[j] = cr;
        nb.VAL[j] = val;
    }
}

template <bool INTERIOR, int SMAX, int LF = 0, class Src>
__device__ __forceinline__ float energy_change(const VmLevelView &L, const VmKParams &P, const Src &,
                                               const NbCacheT<SMAX> &nb, const PixelCtx &c, float dx, float dy, int Lf_rt)
{
    const int Lf = LF ? LF : Lf_rt;
    const float vx = c.v.x + dx, vy = c.v.y + dy;
    // The two bilinear taps of the pixel are shared by its lanes: a fan-out is >= 2 and groups are
    // aligned, so lanes 2k and 2k + 1 belong to one pixel -- the even lane samples image 0 at p - v,
    // the odd one image 1 at p + v (tap()'s arithmetic, one tap per lane instead of two), and a
    // quad-permute hands each the other's.  Same values in every lane as before, bit for bit.
    const bool odd = (threadIdx.x & 1) != 0;
    const float sx = odd ? vx : -vx, sy = odd ? vy : -vy;
    const float mine = tap_g(L.img0, odd ? (int)(L.img1 - L.img0) : 0, L.w, L.h, L.rs, (float)c.px + sx + 0.5f,
                             (float)c.py + sy + 0.5f);
    const float other = dpp_xor1(mine);
    const float lx = odd ? other : mine, ly = odd ? mine : other;
    const float dmx = lx - c.old_luma.x, dmy = ly - c.old_luma.y;
    const float dvx = lx * lx - c.old_luma.x * c.old_luma.x;
    const float dvy = ly * ly - c.old_luma.y * c.old_luma.y;
    const float dcross = lx * ly - c.old_luma.x * c.old_luma.y;
    float acc = 0;
    // (the packed f32 pipe -- v_pk_fma_f32 & co., two windows per instruction -- was tried here in
    // round 3: bit-identical, and no faster: 36.4 vs 37.2 ms per dense 30-pair 1080p pass; on gfx950 a
    // packed f32 instruction occupies the SIMD twice as long as a plain one)
#pragma unroll
    for (int j = 0; j < SMAX; ++j) {
        if (j * Lf < 25) { // uniform in the workgroup
            if (INTERIOR) {
                // the last slot of a lane may lie past the 25th neighbour: it then holds a copy
                // of the centre neighbour and is masked by the k < 25 test below
                const float val = ssim_core(window_mean(nb.A[j], dmx, 0.04f), window_mean(nb.B[j], dmy, 0.04f), nb.VX[j] + dvx,
                                            nb.VY[j] + dvy, nb.X[j] + dcross, 25.0f, P.ssim_clamp);
                const float d = nb.VAL[j] - val;
                acc += ((threadIdx.x & (Lf - 1)) + j * Lf < 25) ? d : 0.0f;
            } else {
                const bool valid = nb.N[j] != 0.0f;
                const float n = valid ? nb.N[j] : 25.0f;
                const float in = n == 25.0f ? 0.04f : __builtin_amdgcn_rcpf(n);
                const float val = ssim_core(window_mean(nb.A[j], dmx, in), window_mean(nb.B[j], dmy, in), nb.VX[j] + dvx,
                                            nb.VY[j] + dvy, nb.X[j] + dcross, n, P.ssim_clamp);
                acc += valid ? nb.VAL[j] - val : 0.0f;
            }
        }
    }
    const float change = group_sum(acc, Lf);
    const float dd = dx * dx + dy * dy;
    const float v_tps = fmaf(c.tps_axy, dd, fmaf(c.tps_b.x, dx, c.tps_b.y * dy));
    const float v_ui = fmaf(c.ui_axy, dd, fmaf(c.ui_b.x, dx, c.ui_b.y * dy));
    float e = P.w_ui * v_ui + P.w_ssim * change;
    if (L.temp_mask) // uniform in the launch
        e = fmaf(P.w_temp * c.tmask * L.factor_d, temp_change(c, dx, dy), e);
    return e * L.inv_wh + P.w_tps * v_tps;
}
#endif

// fover_update_isec_min, morph.cu:794-831
__device__ __forceinline__ void fover_isec(float cx, float cy, float gx, float gy, float e0x, float e0y,
                                           float e1x, float e1y, float &t_min)
{
    float dex = e1x - e0x, dey = e1y - e0y;
    float dcx = cx - e0x, dcy = cy - e0y;
    float d = dey * gx - dex * gy;
    float ud = gx * dcy - gy * dcx;
    int sign = signbit(d) ? 1 : 0;
    if (sign) {
        ud = -ud;
        d = -d;
    }
    if (ud >= 0 && ud <= d) {
        float td = dex * dcy - dey * dcx;
        td *= (float)(-sign * 2 + 1);
        if (td >= 0 && td < t_min * d)
            t_min = td / d; // one division per accepted crossing: IEEE in both modes
    }
}

// Where the halfway vectors of a pixel's 8 ring neighbours come from (ring position k =
// offsets (-1,-1) (0,-1) (1,-1) (1,0) (1,1) (0,1) (-1,1) (-1,0)): global memory, or -- in the
// PASS schedule, whose phases hand data over inside a launch -- lanes 0..7 of the pixel's
// 32-lane group, which fetched them with L1-bypassing loads after the tile barrier.
struct RingGlobal {
    const float2 *v;
    __device__ __forceinline__ float2 operator()(int, int gi) const { return v[gi]; }
};
struct RingLanes {
    float2 mine; // lane k < 8 of the group: v of ring neighbour k (anything where it is outside the image)
    __device__ __forceinline__ float2 operator()(int k, int) const
    {
        return make_float2(__shfl(mine.x, k, 32), __shfl(mine.y, k, 32));
    }
};

// fover_calc_isec_min (morph.cu:833-870) with fover_calc_vtx (:782-792, note the
// `p - off` of the original) for one sign
template <class Ring>
__device__ __forceinline__ void fover_ring(const VmLevelView &L, const Ring &ring, int px, int py, float sgn, float vx,
                                           float vy, float gx, float gy, float &t_min)
{
    const int rx[8] = {-1, 0, 1, 1, 1, 0, -1, -1};
    const int ry[8] = {-1, -1, -1, 0, 1, 1, 1, 0};
    float ex[8], ey[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float ux = vx, uy = vy;
        int qx = px + rx[k], qy = py + ry[k];
        const bool in = qx >= 0 && qx < L.w && qy >= 0 && qy < L.h;
        const float2 nv = ring(k, in ? qy * L.rs + qx : py * L.rs + px);
        if (in) {
            ux = sgn * nv.x;
            uy = sgn * nv.y;
        }
        ex[k] = ux + (float)(px - rx[k]);
        ey[k] = uy + (float)(py - ry[k]);
    }
    const float cx = (float)px + vx, cy = (float)py + vy;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        fover_isec(cx, cy, gx, gy, ex[k], ey[k], ex[(k + 1) & 7], ey[(k + 1) & 7], t_min);
}

// pixel_on_border, morph.cu:648-667 (BCOND_CORNER exactly as written there)
__device__ __forceinline__ bool pixel_locked(const VmLevelView &L, int bcond, int px, int py)
{
    if (bcond == VM_BCOND_CORNER)
        return (px == 0 && py == 0) || (px == 0 && py == L.h - 1) ||
               (px == L.w - 1 && py == 0 && px == L.w - 1 && py == L.h - 1);
    if (bcond == VM_BCOND_BORDER)
        return px == 0 || py == 0 || px == L.w - 1 || py == L.h - 1;
    return false;
}

// both halves' values of a wave-uniform-per-half float: lo = lanes 0-31's, hi = lanes 32-63's
__device__ __forceinline__ void halves(float v, float &lo, float &hi)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

// optimize_pixel (morph.cu:1030-1083) after the mask test: gradient, fold-over bound,
// golden-section search.  All L lanes of the pixel's group run it in lockstep and
// agree bit for bit.  Returns true and the accepted step when the energy drops.
template <class Energy, class Ring>
__device__ __forceinline__ bool decide_with(const VmLevelView &L, const VmKParams &P, const PixelCtx &c,
                                            const Energy &energy, const Ring &ring, float2 &step, uint32_t &n_eval VM_TS_ARG)
{
    VM_TS(4);
#define ENERGY(DX, DY) (++n_eval, energy((DX), (DY)))
    // The energy is evaluated at exactly two places of the instruction stream (not at
    // the reference's seven): the sweep kernels must stay inside the instruction cache.
    // compute_gradient, morph.cu:763-778: g = -(E(+eps x) - E(-eps x), E(+eps y) - E(-eps y))
    float gx = 0, gy = 0;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const float sgn = (k & 1) ? -1.0f : 1.0f;
        const float e = ENERGY(k < 2 ? sgn * P.eps : 0.0f, k < 2 ? 0.0f : sgn * P.eps) * sgn;
        if (k < 2)
            gx += e; // (0 + E+) + (-E-) == E+ - E- bit for bit
        else
            gy += e;
    }
    gx = -gx;
    gy = -gy;
    VM_TS(5);
    const float ng = fsqrt(gx * gx + gy * gy);
    if (ng == 0)
        return false;
    gx = fdiv(gx, ng);
    gy = fdiv(gy, ng);
    // prevent_foldover, morph.cu:872-883
    float t_min = 10;
    fover_ring(L, ring, c.px, c.py, -1.0f, -c.v.x, -c.v.y, -gx, -gy, t_min);
    fover_ring(L, ring, c.px, c.py, 1.0f, c.v.x, c.v.y, gx, gy, t_min);
    float cc = fmaxf(t_min - P.eps, 0.0f);
    VM_TS(6);
    // golden_section_search, morph.cu:885-947: step 0 and 1 evaluate the two initial
    // interior points b and x, later steps shrink the bracket
    const float R = 0.618033989f, C = 1.0f - R;
    float a = 0;
    float b = a * R + cc * C, x = b * R + cc * C;
    float fb = 0, fx = 0;
#pragma unroll 1
    for (int s = 0;; ++s) {
        float t = s == 0 ? b : x;
        bool lt = false;
        if (s >= 2) {
            if (!(cc - a > P.eps))
                break;
            lt = fx < fb;
            if (lt) {
                a = b;
                b = x;
                x = b * R + cc * C;
            } else {
                cc = x;
                x = b * R + a * C;
            }
            t = x;
        }
        const float f = ENERGY(gx * t, gy * t);
        if (s == 0) {
            fb = f;
        } else if (s == 1) {
            fx = f;
        } else if (lt) {
            fb = fx;
            fx = f;
        } else {
            const float tmp = b;
            b = x;
            x = tmp;
            fx = fb;
            fb = f;
        }
    }
#undef ENERGY
    VM_TS(7);
    const float tmin = fx < fb ? x : b, fmin = fx < fb ? fx : fb;
    if (!(fmin < 0))
        return false;
    step = make_float2(gx * tmin, gy * tmin);
    return true;
}

// decide_with for a 32-lane energy and a whole wave per pixel: the two halves of the wave evaluate
// two points of the line search at once -- +eps / -eps of a gradient axis, the two initial points
// of the golden section, then per round the point the search needs now and the one it needs next
// if the coming comparison repeats the last (the next point's position depends on that one bit
// only).  Every evaluation that is USED is the one decide_with makes at that step, with the same
// arguments: bit-identical results; n_eval counts the search's evaluations.  (The FAST lean path
// has its own copy, decide64, which also carries the lumas along.)
template <class Energy, class Ring>
__device__ __forceinline__ bool decide_with64(const VmLevelView &L, const VmKParams &P, const PixelCtx &c,
                                              const Energy &energy, const Ring &ring, bool hi, float2 &step,
                                              uint32_t &n_eval)
{
    float gx = 0, gy = 0;
    {
        const float sgn = hi ? -1.0f : 1.0f;
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const float e = energy(r == 0 ? sgn * P.eps : 0.0f, r == 0 ? 0.0f : sgn * P.eps) * sgn;
            float e_lo, e_hi;
            halves(e, e_lo, e_hi);
            if (r == 0) {
                gx += e_lo; // (0 + E+) + (-E-), as decide_with
                gx += e_hi;
            } else {
                gy += e_lo;
                gy += e_hi;
            }
        }
        n_eval += 4;
    }
    gx = -gx;
    gy = -gy;
    const float ng = fsqrt(gx * gx + gy * gy);
    if (ng == 0)
        return false;
    gx = fdiv(gx, ng);
    gy = fdiv(gy, ng);
    float t_min = 10;
    fover_ring(L, ring, c.px, c.py, -1.0f, -c.v.x, -c.v.y, -gx, -gy, t_min);
    fover_ring(L, ring, c.px, c.py, 1.0f, c.v.x, c.v.y, gx, gy, t_min);
    float cc = fmaxf(t_min - P.eps, 0.0f);
    const float R = 0.618033989f, C = 1.0f - R;
    float a = 0;
    float b = a * R + cc * C, x = b * R + cc * C;
    float fb, fx;
    {
        const float t = hi ? x : b;
        halves(energy(gx * t, gy * t), fb, fx);
        n_eval += 2;
    }
#pragma unroll 1
    for (;;) {
        if (!(cc - a > P.eps))
            break;
        const bool lt = fx < fb;
        // this step (decide_with: lt -> a = b, b = x, x = b R + cc C;  else cc = x, x = b R + a C, swap b, x)
        const float a1 = lt ? b : a, cc1 = lt ? cc : x;
        const float xn = lt ? x * R + cc * C : b * R + a * C;
        const float b1 = lt ? x : xn, x1 = lt ? xn : b;
        // the next one under the guess that the comparison repeats
        const bool G = lt;
        const float xn2 = G ? x1 * R + cc1 * C : b1 * R + a1 * C;
        const float t = hi ? xn2 : xn;
        float f1, f2;
        halves(energy(gx * t, gy * t), f1, f2);
        ++n_eval;
        {
            const float nfb = lt ? fx : f1, nfx = lt ? f1 : fb;
            a = a1;
            cc = cc1;
            b = b1;
            x = x1;
            fb = nfb;
            fx = nfx;
        }
        if (cc - a > P.eps && (fx < fb) == G) {
            const float ob = b, ofb = fb;
            a = G ? b : a;
            cc = G ? cc : x;
            b = G ? x : xn2;
            x = G ? xn2 : ob;
            fb = G ? fx : f2;
            fx = G ? f2 : ofb;
            ++n_eval;
        }
    }
    const float tmin = fx < fb ? x : b, fmin = fx < fb ? fx : fb;
    if (!(fmin < 0))
        return false;
    step = make_float2(gx * tmin, gy * tmin);
    return true;
}

// one lane (EXACT) or L lanes (FAST dense path) per pixel
template <bool INTERIOR, int SMAX, int LF = 0, class Src>
__device__ __forceinline__ bool decide(const VmLevelView &L, const VmKParams &P, const Src &src,
                                       const PixelCtx &c, int sub, int Lf, float2 &step, uint32_t &n_eval VM_TS_ARG)
{
    NbCacheT<SMAX> nb;
    nb_load<INTERIOR, SMAX, LF>(nb, L, src, c, sub, Lf);
    return decide_with(
        L, P, c, [&](float dx, float dy) { return energy_change<INTERIOR, SMAX, LF>(L, P, src, nb, c, dx, dy, Lf); },
        RingGlobal{L.v}, step, n_eval VM_TS_PASS);
}

#if VM_EXACT
// ---------------------------------------------------------------------------
// EXACT on 32 lanes per pixel, still bit-identical to the one-lane evaluation: lane k < 25
// computes the SSIM term of window neighbour k (the expensive part: IEEE divisions and square
// roots), every lane then adds the 25 differences in the reference's row-major order -- the
// terms of neighbours outside the image are +0, and x + (+0) == x for every x the running
// sum can hold (it starts at +0 and can never become -0) -- so all lanes end with the bits the
// sequential loop produces.  Taps, quadratic terms, fold-over test and the golden-section
// control run redundantly on every lane, exactly as written for one lane.
struct NbX {
    float2 m, q;
    float cr, val, counter;
    bool ok;
};

template <class Src>
__device__ __forceinline__ void nbx_load(NbX &nb, const VmLevelView &L, const Src &src, const PixelCtx &c, int sub)
{
    const int i = (sub * 13) >> 6, jj = sub - i * 5; // sub / 5, sub % 5 for sub < 32
    const int qx = c.px + jj - 2, qy = c.py + i - 2;
    nb.ok = sub < 25 && qx >= 0 && qx < L.w && qy >= 0 && qy < L.h;
    src.load(nb.ok ? i : 2, nb.ok ? jj : 2, nb.m, nb.q, nb.cr, nb.val);
    nb.counter = nb.ok ? (float)(window_count(qy, L.h) * window_count(qx, L.w)) : 25.0f;
}

__device__ __forceinline__ float energy_x32(const VmLevelView &L, const VmKParams &P, const NbX &nb,
                                            const PixelCtx &c, float dx, float dy)
{
    const float vx = c.v.x + dx, vy = c.v.y + dy;
    // one tap per lane (even lanes: image 0 at p - v, odd lanes: image 1 at p + v), swapped with
    // the neighbour lane: each value is produced by the expression the one-lane code uses
    const bool odd = threadIdx.x & 1;
    const float t = tap(odd ? L.img1 : L.img0, L.w, L.h, L.rs, odd ? c.px + vx + 0.5f : c.px - vx + 0.5f,
                        odd ? c.py + vy + 0.5f : c.py - vy + 0.5f);
    const float o = __shfl_xor(t, 1);
    const float lx = odd ? o : t, ly = odd ? t : o;
    const float dmx = lx - c.old_luma.x, dmy = ly - c.old_luma.y;
    const float dvx = lx * lx - c.old_luma.x * c.old_luma.x;
    const float dvy = ly * ly - c.old_luma.y * c.old_luma.y;
    const float dcross = lx * ly - c.old_luma.x * c.old_luma.y;
    float d = 0.0f;
    if (nb.ok) {
        const float ns = ssim_value(nb.m.x + dmx, nb.m.y + dmy, nb.q.x + dvx, nb.q.y + dvy, nb.cr + dcross,
                                    nb.counter, P.ssim_clamp);
        d = nb.val - ns;
    }
    float change = 0;
#pragma unroll
    for (int k = 0; k < 25; ++k)
        change += __shfl(d, k, 32);
    float v_tps = c.tps_axy * (dx * dx + dy * dy);
    v_tps += c.tps_b.x * dx;
    v_tps += c.tps_b.y * dy;
    float v_ui = c.ui_axy * (dx * dx + dy * dy);
    v_ui += c.ui_b.x * dx;
    v_ui += c.ui_b.y * dy;
    const float tt = L.temp_mask ? P.w_temp * temp_change(c, dx, dy) * c.tmask * L.factor_d : 0.0f;
    return (P.w_ui * v_ui + P.w_ssim * change + tt) * L.inv_wh + P.w_tps * v_tps;
}

template <class Src>
__device__ __forceinline__ bool decide_x32(const VmLevelView &L, const VmKParams &P, const Src &src,
                                           const PixelCtx &c, int sub, float2 &step, uint32_t &n_eval VM_TS_ARG)
{
    NbX nb;
    nbx_load(nb, L, src, c, sub);
    return decide_with(
        L, P, c, [&](float dx, float dy) { return energy_x32(L, P, nb, c, dx, dy); }, RingGlobal{L.v}, step,
        n_eval VM_TS_PASS);
}
// ... with a whole wave per pixel (both halves hold the same neighbour sums)
template <class Src>
__device__ __forceinline__ bool decide_x64(const VmLevelView &L, const VmKParams &P, const Src &src,
                                           const PixelCtx &c, int sub, bool hi, float2 &step, uint32_t &n_eval)
{
    NbX nb;
    nbx_load(nb, L, src, c, sub);
    return decide_with64(
        L, P, c, [&](float dx, float dy) { return energy_x32(L, P, nb, c, dx, dy); }, RingGlobal{L.v}, hi, step,
        n_eval);
}
#endif

#if !VM_EXACT
// ---------------------------------------------------------------------------
// The LEAN line search: exactly 32 lanes per pixel, lane k < 25 owns window neighbour k
// (one set of sums in registers, no slot loops, no run-time fan-out arithmetic).  It is
// the schedule of the latency-bound regime -- few candidates, each waiting on ~21
// dependent energy evaluations -- where the instruction count of ONE evaluation is the
// time: the two bilinear taps are spread over quads (one texel per lane) and exchanged by
// DPP, the 16 fold-over segment tests run on 16 lanes and meet in a min-butterfly, the
// golden-section loop is branch-free.  Same arithmetic per SSIM term as the generic FAST
// path (ssim_core); the tap sums and the energy along the search line are re-associated.
struct Nb1 {
    float A, B, VX, VY, X, VAL, N; // A, B: raw first-moment sums; N = window count, 0: lane owns no in-image neighbour
};

__device__ __forceinline__ float group_min32(float x)
{
    x = fminf(x, dpp_xor1(x));
    x = fminf(x, dpp_xor2(x));
    x = fminf(x, dpp_half_mirror(x));
    x = fminf(x, dpp_mirror(x));
    x = fminf(x, swz_xor16(x));
    return x;
}

// lane `sub` of a pixel's group owns window neighbour (px + sub % 5 - 2, py + sub / 5 - 2)
template <bool INTERIOR>
__device__ __forceinline__ bool nb1_cell(const VmLevelView &L, const PixelCtx &c, int sub, int &i, int &jj, int &qx,
                                         int &qy)
{
    i = (sub * 13) >> 6; // sub / 5 for sub < 32
    jj = sub - i * 5;
    qx = c.px + jj - 2;
    qy = c.py + i - 2;
    return sub < 25 && (INTERIOR || (qx >= 0 && qx < L.w && qy >= 0 && qy < L.h));
}

template <bool INTERIOR>
__device__ __forceinline__ void nb1_make(Nb1 &nb, const VmLevelView &L, bool ok, int qx, int qy, float2 m, float2 q,
                                         float cr, float val)
{
    float n = 25.0f, in = 0.04f;
    if (!INTERIOR) {
        n = ok ? (float)(window_count(qy, L.h) * window_count(qx, L.w)) : 25.0f;
        in = n == 25.0f ? 0.04f : __builtin_amdgcn_rcpf(n);
    }
    (void)in;
    nb.N = ok ? n : 0.0f;
    nb.A = m.x; // RAW first-moment sums: see window_mean()
    nb.B = m.y;
    nb.VX = q.x;
    nb.VY = q.y;
    nb.X = cr;
    nb.VAL = val;
}

template <bool INTERIOR, class Src>
__device__ __forceinline__ void nb1_load(Nb1 &nb, const VmLevelView &L, const Src &src, const PixelCtx &c, int sub)
{
    int i, jj, qx, qy;
    const bool ok = nb1_cell<INTERIOR>(L, c, sub, i, jj, qx, qy);
    float2 m, q;
    float cr, val;
    src.load(ok ? i : 2, ok ? jj : 2, m, q, cr, val);
    nb1_make<INTERIOR>(nb, L, ok, qx, qy, m, q, cr, val);
}

// Per-lane constants of the distributed bilinear taps.  The lanes of a group form 8 quads;
// even quads sample image 0 at p - v, odd quads image 1 at p + v; lane c of a quad owns
// corner (c & 1, c >> 1) of the 2x2 texel footprint: one texel load per lane, the quad sum
// (2 DPP adds) is the tap, the neighbour quad (row_half_mirror) holds the other image's.
struct TapLane {
    float sgn;              // -1: p - v, +1: p + v
    float wxs, wxo, wys, wyo; // corner weight = (wxs a + wxo)(wys b + wyo), a, b = fractions
    int cx, cy;             // corner
    int imgoff;             // element offset of the lane's image from img0 (same slab)
    bool odd;
};

__device__ __forceinline__ TapLane tap_lane_make(const VmLevelView &L, int sub)
{
    TapLane t;
    t.odd = (sub >> 2) & 1;
    t.sgn = t.odd ? 1.0f : -1.0f;
    t.cx = sub & 1;
    t.cy = (sub >> 1) & 1;
    t.wxs = t.cx ? 1.0f : -1.0f;
    t.wxo = t.cx ? 0.0f : 1.0f;
    t.wys = t.cy ? 1.0f : -1.0f;
    t.wyo = t.cy ? 0.0f : 1.0f;
    t.imgoff = t.odd ? (int)(L.img1 - L.img0) : 0;
    return t;
}

__device__ __forceinline__ int med3i(int x, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}

// tex2D(linear, clamp) of both images at p -+ (vx, vy) (the arithmetic of tap(), the four
// products summed in quad order)
// ... in two halves, so that the texel loads of independent evaluations (the four of the
// gradient) can be in flight together: the lane's texel and its corner weight,
__device__ __forceinline__ void taps32_issue(const VmLevelView &L, const TapLane &t, const PixelCtx &c, float vx,
                                             float vy, float &texel, float &wgt)
{
    const float xb = fmaf(t.sgn, vx, (float)c.px), yb = fmaf(t.sgn, vy, (float)c.py);
    float fi = floorf(xb), fj = floorf(yb);
    const float a = xb - fi, b = yb - fj;
    fi = __builtin_amdgcn_fmed3f(fi, -1.0f, (float)L.w);
    fj = __builtin_amdgcn_fmed3f(fj, -1.0f, (float)L.h);
    const int i = med3i((int)fi + t.cx, 0, L.w - 1), j = med3i((int)fj + t.cy, 0, L.h - 1);
    // (a global load off the image base + an unsigned 32-bit byte offset: the images are read-only,
    // both sit in the level's slab)
    texel = *(vm_g_cf32 *)((const char *)L.img0 + ((uint32_t)(t.imgoff + j * L.rs + i) << 2));
    wgt = fmaf(t.wxs, a, t.wxo) * fmaf(t.wys, b, t.wyo);
}
// ... then the quad sums and the exchange between the two images' quads
__device__ __forceinline__ void taps32_finish(const TapLane &t, float texel, float wgt, float &lx, float &ly)
{
    float r = texel * wgt;
    r += dpp_xor1(r);
    r += dpp_xor2(r);
    const float o = dpp_half_mirror(r);
    lx = t.odd ? o : r;
    ly = t.odd ? r : o;
}
__device__ __forceinline__ void taps32(const VmLevelView &L, const TapLane &t, const PixelCtx &c, float vx, float vy,
                                       float &lx, float &ly)
{
    float texel, wgt;
    taps32_issue(L, t, c, vx, vy, texel, wgt);
    taps32_finish(t, texel, wgt, lx, ly);
}

// sum over the window of (stored SSIM value - value with the pixel's lumas replaced by
// lx, ly): ssim_change (morph.cu:671-728) on 32 lanes
template <bool INTERIOR>
__device__ __forceinline__ float change32(const VmKParams &P, const Nb1 &nb, const PixelCtx &c, float lx, float ly)
{
    const float dmx = lx - c.old_luma.x, dmy = ly - c.old_luma.y;
    const float dvx = lx * lx - c.old_luma.x * c.old_luma.x;
    const float dvy = ly * ly - c.old_luma.y * c.old_luma.y;
    const float dcross = lx * ly - c.old_luma.x * c.old_luma.y;
    float acc;
    if (INTERIOR) {
        const float val = ssim_core(window_mean(nb.A, dmx, 0.04f), window_mean(nb.B, dmy, 0.04f), nb.VX + dvx, nb.VY + dvy,
                                    nb.X + dcross, 25.0f, P.ssim_clamp);
        acc = nb.N != 0.0f ? nb.VAL - val : 0.0f;
    } else {
        const bool valid = nb.N != 0.0f;
        const float n = valid ? nb.N : 25.0f;
        const float in = n == 25.0f ? 0.04f : __builtin_amdgcn_rcpf(n);
        const float val = ssim_core(window_mean(nb.A, dmx, in), window_mean(nb.B, dmy, in), nb.VX + dvx, nb.VY + dvy,
                                    nb.X + dcross, n, P.ssim_clamp);
        acc = valid ? nb.VAL - val : 0.0f;
    }
    return group_sum(acc, 32);
}

// prevent_foldover (morph.cu:872-883): lane s < 16 tests segment s & 7 of ring s >> 3
// (ring 0: sign -1 on (-v, -g); ring 1: sign +1 on (v, g)); the bound is the minimum of
// the 16 crossings (the reference's running minimum up to the rounding of `td < t_min d`)
template <class Ring>
__device__ __forceinline__ float fover32(const VmLevelView &L, const Ring &ring, const PixelCtx &c, float gx, float gy,
                                         int sub)
{
    const int k = sub & 7, k1 = (k + 1) & 7;
    const float sgn = (sub & 8) ? 1.0f : -1.0f;
    const float vx = sgn * c.v.x, vy = sgn * c.v.y;
    float ex[2], ey[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int kk = e ? k1 : k;
        // ring offsets (-1,-1) (0,-1) (1,-1) (1,0) (1,1) (0,1) (-1,1) (-1,0), two bits each
        const int rx = ((0x06A4 >> (2 * kk)) & 3) - 1, ry = ((0x6A40 >> (2 * kk)) & 3) - 1;
        const int qx = c.px + rx, qy = c.py + ry;
        float ux = vx, uy = vy;
        const bool in = qx >= 0 && qx < L.w && qy >= 0 && qy < L.h;
        const float2 nv = ring(kk, in ? qy * L.rs + qx : c.idx);
        if (in) {
            ux = sgn * nv.x;
            uy = sgn * nv.y;
        }
        ex[e] = ux + (float)(c.px - rx); // `p - off`, as fover_calc_vtx has it
        ey[e] = uy + (float)(c.py - ry);
    }
    float t_min = 10;
    fover_isec((float)c.px + vx, (float)c.py + vy, sgn * gx, sgn * gy, ex[0], ey[0], ex[1], ey[1], t_min);
    return group_min32(sub < 16 ? t_min : 10.0f);
}

// optimize_pixel after the mask test, 32 lanes in lockstep; on success also the lumas at
// the accepted point (what commit_pixel_motion would sample again).  The gradient uses the
// energy as energy_change (morph.cu:730-761) writes it; along the search line d = g t the
// quadratic terms collapse to t (Q2 t + Q1).
template <bool INTERIOR, class Ring>
__device__ __forceinline__ bool decide32(const VmLevelView &L, const VmKParams &P, const Nb1 &nb, const Ring &ring,
                                         const PixelCtx &c, int sub, float2 &step, float2 &luma,
                                         uint32_t &n_eval VM_TS_ARG)
{
    VM_TS(4);
    n_eval += 4;
    const bool has_temp = L.temp_mask != nullptr; // uniform in the launch
    const float WT = has_temp ? P.w_temp * c.tmask * L.factor_d * L.inv_wh : 0.0f;
    const TapLane tl = tap_lane_make(L, sub);
    float lx, ly;
    float gx = 0, gy = 0;
    // compute_gradient (morph.cu:763-778): the four evaluations do not depend on each other --
    // their texel loads are issued together (one memory latency instead of four), the rest
    // follows in the order of the reference
    float g_tex[4], g_wgt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sgn = (k & 1) ? -1.0f : 1.0f;
        const float dx = k < 2 ? sgn * P.eps : 0.0f, dy = k < 2 ? 0.0f : sgn * P.eps;
        taps32_issue(L, tl, c, c.v.x + dx, c.v.y + dy, g_tex[k], g_wgt[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float sgn = (k & 1) ? -1.0f : 1.0f;
        const float dx = k < 2 ? sgn * P.eps : 0.0f, dy = k < 2 ? 0.0f : sgn * P.eps;
        taps32_finish(tl, g_tex[k], g_wgt[k], lx, ly);
        const float change = change32<INTERIOR>(P, nb, c, lx, ly);
        const float dd = dx * dx + dy * dy;
        const float v_tps = fmaf(c.tps_axy, dd, fmaf(c.tps_b.x, dx, c.tps_b.y * dy));
        const float v_ui = fmaf(c.ui_axy, dd, fmaf(c.ui_b.x, dx, c.ui_b.y * dy));
        float e = (P.w_ui * v_ui + P.w_ssim * change) * L.inv_wh + P.w_tps * v_tps;
        if (has_temp)
            e = fmaf(WT, temp_change(c, dx, dy), e);
        e *= sgn;
        if (k < 2)
            gx += e;
        else
            gy += e;
    }
    gx = -gx;
    gy = -gy;
    VM_TS(5);
    const float ng = fsqrt(gx * gx + gy * gy);
    if (ng == 0)
        return false;
    gx = fdiv(gx, ng);
    gy = fdiv(gy, ng);
    float cc = fmaxf(fover32(L, ring, c, gx, gy, sub) - P.eps, 0.0f);
    VM_TS(6);
    // E(t) = WS change(t) + t (Q2 t + Q1)
    const float gg = gx * gx + gy * gy;
    const float WS = P.w_ssim * L.inv_wh, WU = P.w_ui * L.inv_wh;
    const float Q2 = (WU * c.ui_axy + P.w_tps * c.tps_axy) * gg;
    const float Q1 = WU * (c.ui_b.x * gx + c.ui_b.y * gy) + P.w_tps * (c.tps_b.x * gx + c.tps_b.y * gy);
    // temporal term along the line: WT (|v + g t - ref|_1 - |v - ref|_1)
    const float T0 = fabsf(c.v.x - c.tref.x) + fabsf(c.v.y - c.tref.y);
#define ELINE(T_, F_, LUM_)                                                            \
    {                                                                                  \
        const float nvx_ = fmaf(gx, (T_), c.v.x), nvy_ = fmaf(gy, (T_), c.v.y);        \
        ++n_eval;                                                                      \
        taps32(L, tl, c, nvx_, nvy_, lx, ly);                                          \
        (F_) = fmaf(WS, change32<INTERIOR>(P, nb, c, lx, ly), (T_) * fmaf(Q2, (T_), Q1)); \
        if (has_temp)                                                                  \
            (F_) = fmaf(WT, (fabsf(nvx_ - c.tref.x) + fabsf(nvy_ - c.tref.y)) - T0, (F_)); \
        (LUM_) = make_float2(lx, ly);                                                  \
    }
    // golden_section_search, morph.cu:885-947
    const float R = 0.618033989f, C = 1.0f - R;
    float a = 0;
    float b = cc * C, x = b * R + cc * C;
    float fb, fx;
    float2 lb, lq; // lumas at b and at x
    ELINE(b, fb, lb);
    ELINE(x, fx, lq);
#pragma unroll 1
    for (;;) {
        if (!(cc - a > P.eps))
            break;
        const bool lt = fx < fb;
        // lt: [a, cc] <- [b, cc], b <- x;   else: [a, cc] <- [a, x], x <- b;   one new point
        const float p = lt ? x : b, qv = lt ? cc : a;
        a = lt ? b : a;
        cc = lt ? cc : x;
        const float xn = p * R + qv * C;
        float f;
        float2 lf;
        ELINE(xn, f, lf);
        const float ob = b, ofb = fb;
        const float2 olb = lb;
        b = lt ? x : xn;
        x = lt ? xn : ob;
        fb = lt ? fx : f;
        fx = lt ? f : ofb;
        lb = lt ? lq : lf;
        lq = lt ? lf : olb;
    }
#undef ELINE
    VM_TS(7);
    const float tmin = fx < fb ? x : b, fmin = fx < fb ? fx : fb;
    if (!(fmin < 0))
        return false;
    step = make_float2(gx * tmin, gy * tmin);
    luma = fx < fb ? lq : lb;
    return true;
}

// decide32 with a whole wave per pixel: the line search is a chain of dependent energy
// evaluations (the only thing a launch-bound small level waits for), and the chip is idle there,
// so the two halves of the wave evaluate two points at once --
//   gradient: +eps and -eps of an axis side by side (2 rounds instead of 4),
//   the two initial points of the golden section side by side,
//   then per round the point the search needs now AND the point it will need next if the coming
//   comparison goes the way the last one went (the next point's position depends only on that one
//   bit).  When the guess holds the round completes two steps of the search.
// Every evaluation that is USED is the one decide32 makes at that step (same point, same
// arithmetic, same lanes' roles), so the result is bit-identical to decide32's; n_eval counts
// the search's evaluations, not the speculative ones.
// (Round 4 tried FOUR points per round -- the wave's four 16-lane rows, two window neighbours per lane, the two
// 16-lane trees of the 32-lane sum side by side: the gradient in one round, two golden-section steps per round
// for certain and a third under the same guess; bit-identical, all schedule-equality tests green.  Measured: no
// gain -- PASS 11.37 vs 11.15 us per phase, STEP 14.5 vs 14.1 (tools/dev_pass.py).  A round then costs ~255
// instructions (two SSIM terms, two trees, 12 v_readlane broadcasts, the bookkeeping of a three-step speculation)
// against ~150 here for ~2.7 instead of ~1.9 steps: 94 against 79 instructions per step of the search.  Removed.)
template <bool INTERIOR, class Ring>
__device__ __forceinline__ bool decide64(const VmLevelView &L, const VmKParams &P, const Nb1 &nb, const Ring &ring,
                                         const PixelCtx &c, int sub, bool hi, float2 &step, float2 &luma,
                                         uint32_t &n_eval)
{
    n_eval += 4;
    const bool has_temp = L.temp_mask != nullptr; // uniform in the launch
    const float WT = has_temp ? P.w_temp * c.tmask * L.factor_d * L.inv_wh : 0.0f;
    const TapLane tl = tap_lane_make(L, sub);
    float lx, ly;
    float gx = 0, gy = 0;
    {
        // compute_gradient (morph.cu:763-778): half 0 takes +eps (k = 0, 2), half 1 -eps (k = 1, 3)
        const float sgn = hi ? -1.0f : 1.0f;
        float g_tex[2], g_wgt[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float dx = r == 0 ? sgn * P.eps : 0.0f, dy = r == 0 ? 0.0f : sgn * P.eps;
            taps32_issue(L, tl, c, c.v.x + dx, c.v.y + dy, g_tex[r], g_wgt[r]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float dx = r == 0 ? sgn * P.eps : 0.0f, dy = r == 0 ? 0.0f : sgn * P.eps;
            taps32_finish(tl, g_tex[r], g_wgt[r], lx, ly);
            const float change = change32<INTERIOR>(P, nb, c, lx, ly);
            const float dd = dx * dx + dy * dy;
            const float v_tps = fmaf(c.tps_axy, dd, fmaf(c.tps_b.x, dx, c.tps_b.y * dy));
            const float v_ui = fmaf(c.ui_axy, dd, fmaf(c.ui_b.x, dx, c.ui_b.y * dy));
            float e = (P.w_ui * v_ui + P.w_ssim * change) * L.inv_wh + P.w_tps * v_tps;
            if (has_temp)
                e = fmaf(WT, temp_change(c, dx, dy), e);
            e *= sgn;
            float e_lo, e_hi;
            halves(e, e_lo, e_hi);
            if (r == 0) {
                gx += e_lo;
                gx += e_hi;
            } else {
                gy += e_lo;
                gy += e_hi;
            }
        }
    }
    gx = -gx;
    gy = -gy;
    const float ng = fsqrt(gx * gx + gy * gy);
    if (ng == 0)
        return false;
    gx = fdiv(gx, ng);
    gy = fdiv(gy, ng);
    float cc = fmaxf(fover32(L, ring, c, gx, gy, sub) - P.eps, 0.0f);
    const float gg = gx * gx + gy * gy;
    const float WS = P.w_ssim * L.inv_wh, WU = P.w_ui * L.inv_wh;
    const float Q2 = (WU * c.ui_axy + P.w_tps * c.tps_axy) * gg;
    const float Q1 = WU * (c.ui_b.x * gx + c.ui_b.y * gy) + P.w_tps * (c.tps_b.x * gx + c.tps_b.y * gy);
    const float T0 = fabsf(c.v.x - c.tref.x) + fabsf(c.v.y - c.tref.y);
    // the energy at t of this half; f and the lumas of both halves come back
#define ELINE2(T_, FLO_, FHI_, LLO_, LHI_)                                              \
    {                                                                                  \
        const float nvx_ = fmaf(gx, (T_), c.v.x), nvy_ = fmaf(gy, (T_), c.v.y);        \
        taps32(L, tl, c, nvx_, nvy_, lx, ly);                                          \
        float f_ = fmaf(WS, change32<INTERIOR>(P, nb, c, lx, ly), (T_) * fmaf(Q2, (T_), Q1)); \
        if (has_temp)                                                                  \
            f_ = fmaf(WT, (fabsf(nvx_ - c.tref.x) + fabsf(nvy_ - c.tref.y)) - T0, f_); \
        halves(f_, (FLO_), (FHI_));                                                    \
        halves(lx, (LLO_).x, (LHI_).x);                                                \
        halves(ly, (LLO_).y, (LHI_).y);                                                \
    }
    // golden_section_search, morph.cu:885-947
    const float R = 0.618033989f, C = 1.0f - R;
    float a = 0;
    float b = cc * C, x = b * R + cc * C;
    float fb, fx;
    float2 lb, lq; // lumas at b and at x
    ELINE2(hi ? x : b, fb, fx, lb, lq);
    n_eval += 2;
#pragma unroll 1
    for (;;) {
        if (!(cc - a > P.eps))
            break;
        const bool lt = fx < fb;
        // this step: [a, cc] <- [b, cc], b <- x  or  [a, cc] <- [a, x], x <- b;  one new point xn
        const float p = lt ? x : b, qv = lt ? cc : a;
        const float a1 = lt ? b : a, cc1 = lt ? cc : x;
        const float xn = p * R + qv * C;
        const float b1 = lt ? x : xn, x1 = lt ? xn : b;
        // the step after it, should its comparison fall like this one did
        // The guess: the comparison repeats.  Measured on the 120x68 level of the 1080p pair it
        // holds for 99 % of the steps (most accepted moves are small: the search keeps shrinking
        // towards 0); a parabola through the three known points predicted no better and cost more.
        const bool G = lt;
        const float p2 = G ? x1 : b1, q2 = G ? cc1 : a1;
        const float xn2 = p2 * R + q2 * C;
        float f1, f2;
        float2 l1, l2;
        ELINE2(hi ? xn2 : xn, f1, f2, l1, l2);
        ++n_eval;
        {
            const float nfb = lt ? fx : f1, nfx = lt ? f1 : fb;
            const float2 nlb = lt ? lq : l1, nlq = lt ? l1 : lb;
            a = a1;
            cc = cc1;
            b = b1;
            x = x1;
            fb = nfb;
            fx = nfx;
            lb = nlb;
            lq = nlq;
        }
        if (cc - a > P.eps && (fx < fb) == G) { // the guess held: f2 is the next step's evaluation
            const float ob = b, ofb = fb;
            const float2 olb = lb;
            a = G ? b : a;
            cc = G ? cc : x;
            b = G ? x : xn2;
            x = G ? xn2 : ob;
            fb = G ? fx : f2;
            fx = G ? f2 : ofb;
            lb = G ? lq : l2;
            lq = G ? l2 : olb;
            ++n_eval;
        }
    }
#undef ELINE2
    const float tmin = fx < fb ? x : b, fmin = fx < fb ? fx : fb;
    if (!(fmin < 0))
        return false;
    step = make_float2(gx * tmin, gy * tmin);
    luma = fx < fb ? lq : lb;
    return true;
}
#endif

// everything of a pixel that the energy needs besides the window sums
__device__ __forceinline__ void ctx_load(PixelCtx &c, const VmLevelView &L, const float *s_tps, int px, int py)
{
    c.px = px;
    c.py = py;
    c.idx = py * L.rs + px;
    c.v = L.v[c.idx];
    c.old_luma = L.luma[c.idx];
    c.ui_axy = L.ui_axy[c.idx];
    c.ui_b = L.ui_b[c.idx];
    c.tps_axy = s_tps[(border_class(py, L.h) * 5 + border_class(px, L.w)) * 25 + 12] / 2;
    c.tref = make_float2(0, 0);
    c.tmask = 0.0f;
    if (L.temp_mask) { // uniform in the launch
        c.tref = L.temp_ref[c.idx];
        c.tmask = L.temp_mask[c.idx];
    }
}

__device__ __forceinline__ bool is_interior(const VmLevelView &L, int px, int py)
{
    return px >= 4 && px < L.w - 4 && py >= 4 && py < L.h - 4;
}

// geometry of the improving-mask words around a tile
struct MaskGeom {
    int bx0, by0, nbx, nby;
};
__device__ __forceinline__ MaskGeom mask_geom(int w, int h, int ox, int oy)
{
    MaskGeom g;
    g.bx0 = ox / 5 - 1;
    g.by0 = oy / 5 - 1;
    const int bx1 = min(ox + VM_TILE_W - 1, w - 1) / 5 + 1;
    const int by1 = min(oy + VM_TILE_H - 1, h - 1) / 5 + 1;
    g.nbx = bx1 - g.bx0 + 1; // <= 16
    g.nby = by1 - g.by0 + 1; // <= 6
    return g;
}
__device__ __forceinline__ MaskGeom mask_geom(const VmLevelView &L, int ox, int oy) { return mask_geom(L.w, L.h, ox, oy); }

// get_improve_mask_idx, morph.cu:621-646, on the LDS copy of the mask words
__device__ __forceinline__ bool mask_hit(const uint32_t (*mask)[16], const uint32_t *imp, const MaskGeom &g,
                                         int px, int py)
{
    const int oxb = px % 5, oyb = py % 5;
    const int mcx = px / 5 - g.bx0, mcy = py / 5 - g.by0;
    const int begi = oyb >= 2 ? 1 : 0, begj = oxb >= 2 ? 1 : 0;
    const uint32_t *ib = imp + (oyb * 5 + oxb) * 9;
    bool hit = false;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (mask[mcy + begi + i - 1][mcx + begj + j - 1] & ib[(begi + i) * 3 + begj + j])
                hit = true;
    return hit;
}

// bits of mask word (bx, by) whose pixels lie inside [x0, x1] x [y0, y1]
__device__ __forceinline__ uint32_t block_bits_in(int bx, int by, int x0, int x1, int y0, int y1)
{
    uint32_t cols = 0, m = 0;
#pragma unroll
    for (int r = 0; r < 5; ++r)
        if (5 * bx + r >= x0 && 5 * bx + r <= x1)
            cols |= 1u << r;
#pragma unroll
    for (int r = 0; r < 5; ++r)
        if (5 * by + r >= y0 && 5 * by + r <= y1)
            m |= cols << (5 * r);
    return m;
}

// Can the tile at (ox, oy) have a candidate at all?  A pixel is one iff a set mask bit lies within +-2 of it
// (get_improve_mask_idx), so only the bits of the positions inside the tile's rectangle + 2 count -- pixels of this
// tile or of the gaps around it, which no other tile of the pass owns: a race-free test, the same in every
// schedule.  (Positions past the right / bottom image edge count: init_improving_mask sets whole words and
// nothing clears the bits of pixels that do not exist, see k_pass.)
__device__ __forceinline__ uint32_t tile_reach_bits(const VmLevelView &L, int ox, int oy, int bx, int by)
{
    return block_bits_in(bx, by, ox - 2, min(ox + VM_TILE_W - 1, L.w - 1) + 2, oy - 2, min(oy + VM_TILE_H - 1, L.h - 1) + 2);
}

// ordered compaction of the (at most 256) flagged phase pixels of a workgroup into
// list[]: slot order, hence identical in every workgroup that looks at the same tile.
// Every thread of the workgroup calls it; returns the number of entries.
// (`also`: a second flag of the same threads; *any_also = it is set somewhere -- bit 16 of the counts)
__device__ __forceinline__ int compact256(bool flag, int tid, int *list, int *wave_cnt, bool also = false,
                                          bool *any_also = nullptr)
{
    const unsigned long long b = __ballot(flag);
    const unsigned long long b2 = __ballot(also);
    const int lane = tid & 63, wave = tid >> 6;
    if (tid < 256 && lane == 0)
        wave_cnt[wave] = __popcll(b) | (b2 ? 1 << 16 : 0);
    __syncthreads();
    const int w0 = wave_cnt[0], w1 = wave_cnt[1], w2 = wave_cnt[2], w3 = wave_cnt[3];
    const int c0 = w0 & 0xFFFF, c1 = w1 & 0xFFFF, c2 = w2 & 0xFFFF, c3 = w3 & 0xFFFF;
    if (any_also)
        *any_also = ((w0 | w1 | w2 | w3) >> 16) != 0;
    if (tid < 256 && flag) {
        const int off = (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
        list[off + __popcll(b & ((1ull << lane) - 1ull))] = tid;
    }
    __syncthreads();
    return c0 + c1 + c2 + c3;
}

// commit_pixel_motion (morph.cu:990-1026) for the pixel of slot `tid`: own-pixel state,
// the record the per-cell gather reads, the mask bit.  Returns true for a commit.
// gather_cell with the committed slots given as bitmap rows (rowbits[t]: bit sx of slot row sy0 + t,
// already cut down to the slots within +-2 of the cell): no d_ok reads, no loop over empty slots;
// the same records in the same row-major order
template <class LdsT>
__device__ __forceinline__ void gather_cell_bits(const LdsT &S, const VmLevelView &L, int ox, int oy, int rx, int ry,
                                                 int pi, int pj, int sy0, int sx0, const uint32_t (&rowbits)[3], float2 &m,
                                                 float2 &q, float &cr, float2 &tb, int order)
{
    auto add = [&](int t, int sx) {
        const int y = 2 * (sy0 + t) + pi, x = 2 * sx + pj;
        const int rec = (sy0 + t) * 32 + sx;
        const float2 dm = S.d_mean[rec], dv = S.d_var[rec], st = S.d_step[rec];
        m.x += dm.x;
        m.y += dm.y;
        q.x += dv.x;
        q.y += dv.y;
        cr += S.d_cross[rec];
        const int By = border_class(oy + y, L.h), Bx = border_class(ox + x, L.w);
        const float k = S.tps[(By * 5 + Bx) * 25 + (ry - y + 2) * 5 + (rx - x + 2)];
        tb.x += st.x * k;
        tb.y += st.y * k;
    };
#if VM_EXACT
    // order (vm_set_commit_order): bit 0 = reversed, bit 1 = column-major over the committing pixels --
    // equally legal orders of the commits the reference leaves to atomics
    if (order) {
        for (int k0 = 0; k0 < 9; ++k0) {
            const int k = (order & 1) ? 8 - k0 : k0;
            const int t = (order & 2) ? k % 3 : k / 3, a = (order & 2) ? k / 3 : k % 3;
            const int sx = sx0 + a;
            if (sx < 32 && ((rowbits[t] >> sx) & 1u))
                add(t, sx);
        }
        return;
    }
#else
    (void)order;
#endif
    (void)sx0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        uint32_t bits = rowbits[t];
        while (bits) {
            const int sx = __ffs(bits) - 1;
            bits &= ~(1u << sx);
            add(t, sx);
        }
    }
}

template <class LdsT>
__device__ __forceinline__ bool commit_own(LdsT &S, const VmLevelView &L, const MaskGeom &g, int tid, int ox,
                                           int oy, int pi, int pj)
{
    const int state = S.d_ok[tid];
    if (state == 0)
        return false;
    const int tx = tid & 31, ty = tid >> 5;
    const int px = ox + tx * 2 + pj, py = oy + ty * 2 + pi;
    const int mcx = px / 5 - g.bx0, mcy = py / 5 - g.by0;
    const uint32_t bit = 1u << ((px % 5) + (py % 5) * 5);
    if (state == 2) {
        atomicAnd(&S.mask[mcy][mcx], ~bit);
        return false;
    }
    if (state == 3) { // the lean line search wrote the pixel's own state and its record already
        S.d_ok[tid] = 1;
        atomicOr(&S.mask[mcy][mcx], bit);
        return true;
    }
    const int idx = py * L.rs + px;
    const float2 v = L.v[idx], ol = L.luma[idx], st = S.d_step[tid];
    const float2 newv = make_float2(v.x + st.x, v.y + st.y);
    const float lx = tap(L.img0, L.w, L.h, L.rs, px - newv.x + 0.5f, py - newv.y + 0.5f);
    const float ly = tap(L.img1, L.w, L.h, L.rs, px + newv.x + 0.5f, py + newv.y + 0.5f);
    L.luma[idx] = make_float2(lx, ly);
    S.d_mean[tid] = make_float2(lx - ol.x, ly - ol.y);
    S.d_var[tid] = make_float2(lx * lx - ol.x * ol.x, ly * ly - ol.y * ol.y);
    S.d_cross[tid] = lx * ly - ol.x * ol.y;
    const float axy = L.ui_axy[idx];
    const float2 ub = L.ui_b[idx];
    L.ui_b[idx] = make_float2(ub.x + 2 * st.x * axy, ub.y + 2 * st.y * axy);
    L.v[idx] = newv;
    atomicOr(&S.mask[mcy][mcx], bit);
    return true;
}

// ssim_update (morph.cu:951-988) + the tps.b scatter (:1006-1015) as a gather: the
// tile+halo cell (rx, ry) adds the records of the committed pixels of this phase
// whose 5x5 window contains it, in row-major order of those pixels.  Returns whether
// any record touched the cell; the caller then recomputes the SSIM value
// (UpdateSSIM, :1258-1279).
template <class LdsT>
__device__ __forceinline__ bool gather_cell(const LdsT &S, const VmLevelView &L, int ox, int oy, int rx, int ry,
                                            int pi, int pj, float2 &m, float2 &q, float &cr, float2 &tb, int order)
{
    int ylo = max(ry - 2, 0), yhi = min(ry + 2, VM_TILE_H - 1);
    int xlo = max(rx - 2, 0), xhi = min(rx + 2, VM_TILE_W - 1);
    ylo += (ylo & 1) ^ pi;
    xlo += (xlo & 1) ^ pj;
    bool touched = false;
#if VM_EXACT
    // diagnostic (vm_set_commit_order): the same records in another order -- bit 0 = reversed, bit 1 =
    // column-major over the committing pixels; equally legal orders of the commits the reference
    // leaves to atomics
    if (order) {
        const int ny = yhi >= ylo ? ((yhi - ylo) >> 1) + 1 : 0, nx = xhi >= xlo ? ((xhi - xlo) >> 1) + 1 : 0;
        const int n = ny * nx;
        for (int k0 = 0; k0 < n; ++k0) {
            const int k = (order & 1) ? n - 1 - k0 : k0;
            const int iy = (order & 2) ? k % ny : k / nx, ix = (order & 2) ? k / ny : k % nx;
            const int y = ylo + 2 * iy, x = xlo + 2 * ix;
            const int rec = (y >> 1) * 32 + (x >> 1);
            if (S.d_ok[rec] != 1)
                continue;
            touched = true;
            const float2 dm = S.d_mean[rec], dv = S.d_var[rec], st = S.d_step[rec];
            m.x += dm.x;
            m.y += dm.y;
            q.x += dv.x;
            q.y += dv.y;
            cr += S.d_cross[rec];
            const int By = border_class(oy + y, L.h), Bx = border_class(ox + x, L.w);
            const float k2 = S.tps[(By * 5 + Bx) * 25 + (ry - y + 2) * 5 + (rx - x + 2)];
            tb.x += st.x * k2;
            tb.y += st.y * k2;
        }
        return touched;
    }
#else
    (void)order;
#endif
    for (int y = ylo; y <= yhi; y += 2)
        for (int x = xlo; x <= xhi; x += 2) {
            const int rec = (y >> 1) * 32 + (x >> 1);
            if (S.d_ok[rec] != 1)
                continue;
            touched = true;
            const float2 dm = S.d_mean[rec], dv = S.d_var[rec], st = S.d_step[rec];
            m.x += dm.x;
            m.y += dm.y;
            q.x += dv.x;
            q.y += dv.y;
            cr += S.d_cross[rec];
            const int By = border_class(oy + y, L.h), Bx = border_class(ox + x, L.w);
            const float k = S.tps[(By * 5 + Bx) * 25 + (ry - y + 2) * 5 + (rx - x + 2)];
            tb.x += st.x * k;
            tb.y += st.y * k;
        }
    return touched;
}

// ===========================================================================
// TILE schedule
// One tile of one pass (one thread block of kernel_optimize_level, morph.cu:1281-1345): mask
// test, LoadSSIM, four Jacobi phases, SaveSSIM.  Shared by the TILE kernel (one workgroup per
// tile, one launch per pass) and the SPARSE kernel (one workgroup per frame pair walking the
// few active tiles of a pruned level).  Returns false when no mask word near the tile is set
// (nothing read, nothing written).
// DENSE = false (FAST only): the variant for pruned sweeps -- every phase runs the lean line
// search, 16 candidates per round; without the dense path the kernel needs 134 instead of
// 256 VGPRs (measured: pruned sweeps 5-8 % faster).
// sp_list / sp_cnt / sp_cap (SPARSE kernel, list kept in LDS): the set words this tile owns are appended to
// the pass's new word list as they are written back (entries past sp_cap are only counted: overflow).
template <bool DENSE, int SMAX = VM_SMAX, int MINF = VM_MIN_FANOUT, bool INTV = true>
__device__ __forceinline__ bool tile_sweep(TileLds &S, const VmLevelView &L, const VmKParams &P,
                                           const uint32_t *__restrict__ tables, bool tables_staged, int ox, int oy,
                                           int tid, int T, bool &improving, uint32_t &st_cand, uint32_t &st_commit,
                                           uint32_t *sp_list = nullptr, uint32_t *sp_val = nullptr, uint32_t *sp_cnt = nullptr,
                                           uint32_t sp_cap = 0)
{
    VM_TTSF(0);
    // --- improving-mask words of the tile and its ring of neighbour blocks ---
    const MaskGeom g = mask_geom(L, ox, oy);
    uint32_t mymask = 0;
    if (tid < g.nbx * g.nby) {
        int mx = tid % g.nbx, my = tid / g.nbx;
        mymask = L.impmask[(g.by0 + my + 1) * L.imp_rs + (g.bx0 + mx + 1)];
        S.mask[my][mx] = mymask;
    }
    // tile-level early out: no set bit within +-2 of the tile means no pixel of it is a candidate in
    // phase 0, hence no commit, hence none in the later phases; and re-deriving the SSIM values from
    // unchanged sums reproduces them bit for bit.  (Rounds 1-3 tested whole words of the window: a tile
    // beside an active one was staged and walked through four empty phases, ~5 us a time -- a third of a
    // cycling level's iteration.)
    {
        uint32_t reach = 0;
        if (tid < g.nbx * g.nby)
            reach = mymask & tile_reach_bits(L, ox, oy, g.bx0 + tid % g.nbx, g.by0 + tid / g.nbx);
        if (!__syncthreads_or(reach != 0))
            return false;
    }

    if (!DENSE && tid < (VM_NCELL + 31) / 32)
        S.dirty[tid] = 0; // (ordered before the first gather by the barriers below)
    if (!tables_staged) { // TILE: per launch, after the early out; SPARSE: once per kernel
        for (int k = tid; k < 625; k += T)
            S.tps[k] = __uint_as_float(tables[VM_TAB_TPS + k]);
        for (int k = tid; k < 225; k += T)
            S.imp[k] = tables[VM_TAB_IMP + k];
    }

    // --- LoadSSIM (morph.cu:1214-1234) + the tile's tps.b ---
    // (three cells per thread in flight: one HBM/L2 round trip for a 512-thread workgroup)
    for (int c0 = tid; c0 < VM_NCELL; c0 += 3 * T) {
        float2 m[3], q[3], tb[3];
        float cr[3], val[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int c = c0 + e * T;
            const int gx = ox - 2 + c % VM_HALO_W, gy = oy - 2 + c / VM_HALO_W;
            const bool in = c < VM_NCELL && gx >= 0 && gx < L.w && gy >= 0 && gy < L.h;
            const int gi = in ? gy * L.rs + gx : 0;
            m[e] = L.mean[gi];
            q[e] = L.var[gi];
            tb[e] = L.tps_b[gi];
            cr[e] = L.cross[gi];
            val[e] = L.value[gi];
            if (!in) {
                m[e] = q[e] = tb[e] = make_float2(0, 0);
                cr[e] = val[e] = 0.0f;
            }
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int c = c0 + e * T;
            if (c < VM_NCELL) {
                S.mean[c] = m[e];
                S.var[c] = q[e];
                S.tpsb[c] = tb[e];
                S.cross[c] = cr[e];
                S.value[c] = val[e];
            }
        }
    }
    __syncthreads();
    VM_TTSF(1);

    bool tile_improving = false;

    for (int pi = 0; pi < 2; ++pi) {
        for (int pj = 0; pj < 2; ++pj) {
            VM_TTS(pi * 2 + pj, 0);
            // ---- 1. candidates of this phase ----
            bool cand = false, hit = false;
            if (tid < 256) {
                const int px = ox + (tid & 31) * 2 + pj, py = oy + (tid >> 5) * 2 + pi;
                int state = 0;
                if (px < L.w && py < L.h && mask_hit(S.mask, S.imp, g, px, py)) {
                    state = 2; // in the mask: its bit is cleared unless it commits
                    hit = true;
                    cand = !pixel_locked(L, P.bcond, px, py);
                }
                S.d_ok[tid] = state;
            }
            bool any_hit;
            const int n_act = compact256(cand, tid, S.list, S.wave_cnt, hit, &any_hit);
            // no pixel of this phase in the mask: nothing to search, nothing to commit, no bit to clear
            // (a pruned tile visit is mostly such phases: 1.6 -> 0.9 us each)
            if (!any_hit) {
                VM_TTS(pi * 2 + pj, 1);
                VM_TTS(pi * 2 + pj, 2);
                VM_TTS(pi * 2 + pj, 3);
                VM_TTS(pi * 2 + pj, 4);
                VM_TTS(pi * 2 + pj, 5);
                continue;
            }
            VM_TTS(pi * 2 + pj, 1);

            if (n_act > 0) {
                st_cand += n_act;
                // ---- 2. line searches on the pre-phase state, L lanes per candidate ----
#if !VM_EXACT
                // (the 128-VGPR dense kernel keeps to the dense path: fewer live ranges; so does the kernel of
                // the small levels, INTV == false: their phases are full for hundreds of iterations, and the
                // four lean line-search bodies beside the dense one cost 2-3 % per dense pass and 6 % of the
                // 60-pair job even though they never run -- r03, 82.1 -> 87.1 G pixel*iters/s)
                if (!DENSE || (SMAX > 7 && INTV && n_act * 32 <= T)) {
                    // sparse phase: the lean 32-lane line search; with <= T / 64 candidates a whole
                    // wave each, two points of the search per round (decide64)
                    const bool wide = n_act * 64 <= T;
                    for (int base = 0; base < n_act; base += wide ? T / 64 : T / 32) {
                    const int li = base + (wide ? tid >> 6 : tid >> 5), sub = tid & 31;
                    const bool writer = wide ? (tid & 63) == 0 : sub == 0;
                    const int slot = S.list[min(li, n_act - 1)];
                    const int tx = slot & 31, ty = slot >> 5;
                    const int px = ox + tx * 2 + pj, py = oy + ty * 2 + pi;
                    const bool wave_interior = __all(li >= n_act || is_interior(L, px, py));
                    if (li < n_act) {
                        PixelCtx c;
                        ctx_load(c, L, S.tps, px, py);
                        LdsSrc src{&S, (ty * 2 + pi) * VM_HALO_W + (tx * 2 + pj)};
                        c.tps_b = S.tpsb[src.hc + 2 * VM_HALO_W + 2];
                        float2 step, luma;
#ifdef VM_PROF
                        unsigned long long ts[16];
#endif
                        Nb1 nb;
                        bool ok;
                        uint32_t n_eval = 0;
                        if (wave_interior) {
                            nb1_load<true>(nb, L, src, c, sub);
                            ok = wide ? decide64<true>(L, P, nb, RingGlobal{L.v}, c, sub, (tid & 32) != 0, step, luma, n_eval)
                                      : decide32<true>(L, P, nb, RingGlobal{L.v}, c, sub, step, luma, n_eval VM_TS_PASS);
                        } else {
                            nb1_load<false>(nb, L, src, c, sub);
                            ok = wide ? decide64<false>(L, P, nb, RingGlobal{L.v}, c, sub, (tid & 32) != 0, step, luma, n_eval)
                                      : decide32<false>(L, P, nb, RingGlobal{L.v}, c, sub, step, luma, n_eval VM_TS_PASS);
                        }
                        if (writer)
                            atomicAdd(&S.n_eval, n_eval);
                        if (ok && writer) {
                            // commit_pixel_motion (morph.cu:990-1026), the pixel's own part, at once:
                            // nothing else of this phase reads its v, luma or ui.b (state 3)
                            const float2 ol = c.old_luma;
                            S.d_step[slot] = step;
                            S.d_mean[slot] = make_float2(luma.x - ol.x, luma.y - ol.y);
                            S.d_var[slot] = make_float2(luma.x * luma.x - ol.x * ol.x, luma.y * luma.y - ol.y * ol.y);
                            S.d_cross[slot] = luma.x * luma.y - ol.x * ol.y;
                            L.luma[c.idx] = luma;
                            L.ui_b[c.idx] = make_float2(c.ui_b.x + 2 * step.x * c.ui_axy, c.ui_b.y + 2 * step.y * c.ui_axy);
                            L.v[c.idx] = make_float2(c.v.x + step.x, c.v.y + step.y);
                            S.d_ok[slot] = 3;
                        }
                    }
                    }
                } else
#else
                if (n_act * 32 <= 4 * T) {
                    // up to four rounds of T / 32 candidates on 32 lanes each (decide_x32); up to
                    // T / 64 candidates get a whole wave each (decide_x64)
                    const bool wide = n_act * 64 <= T;
                    for (int base = 0; base < n_act; base += wide ? T / 64 : T / 32) {
                        const int li = base + (wide ? tid >> 6 : tid >> 5), sub = tid & 31;
                        const bool writer = wide ? (tid & 63) == 0 : sub == 0;
                        const int slot = S.list[min(li, n_act - 1)];
                        const int tx = slot & 31, ty = slot >> 5;
                        const int px = ox + tx * 2 + pj, py = oy + ty * 2 + pi;
                        if (li < n_act) {
                            PixelCtx c;
                            ctx_load(c, L, S.tps, px, py);
                            LdsSrc src{&S, (ty * 2 + pi) * VM_HALO_W + (tx * 2 + pj)};
                            c.tps_b = S.tpsb[src.hc + 2 * VM_HALO_W + 2];
                            float2 step;
#ifdef VM_PROF
                            unsigned long long ts[16];
#endif
                            uint32_t n_eval = 0;
                            const bool ok = wide ? decide_x64(L, P, src, c, sub, (tid & 32) != 0, step, n_eval)
                                                 : decide_x32(L, P, src, c, sub, step, n_eval VM_TS_PASS);
                            if (writer)
                                atomicAdd(&S.n_eval, n_eval);
                            if (ok && writer) {
                                S.d_step[slot] = step;
                                S.d_ok[slot] = 1;
                            }
                        }
                    }
                } else
#endif
                if (DENSE) {
                // A FIXED fan-out of MINF lanes per candidate (2 in the 256-VGPR kernel, 4 in the 128-VGPR one),
                // known at compile time.  Rounds 1-2 grew it with the free lanes (2 ... 16: "sparse phases
                // fill the CU with neighbours"); measured in round 3 that was a loss everywhere: the
                // generic code carries a uniform branch per window slot and per reduction stage, every
                // extra lane of a candidate repeats the taps and the line-search bookkeeping, and a phase
                // of few candidates is the lean path's anyway.  Same workloads, us per dense pass, adaptive
                // vs fixed: 30 x 120x68 258 -> 184, 30 x 240x135 881 -> 619, 30 x 480x270 2504 -> 1912,
                // 8 x 960x540 2358 -> 2026, 8 x 1080p 36.6 -> 31.1 ms (tools/dev_dense.py); 60 pairs on
                // one GPU 68.1 -> 82.2 G pixel*iters/s.
                constexpr int Lf = MINF;
                const int slots = T / Lf;
                const int sub = tid & (Lf - 1), grp = tid / Lf;
                for (int base = 0; base < n_act; base += slots) {
                    const int li = base + grp;
                    const int slot = S.list[min(li, n_act - 1)];
                    const int tx = slot & 31, ty = slot >> 5;
                    const int px = ox + tx * 2 + pj, py = oy + ty * 2 + pi;
                    // the constant-count fast path is taken per WAVE (all its pixels interior):
                    // a per-pixel choice would make mixed waves run both line searches
                    const bool wave_interior = INTV && __all(li >= n_act || is_interior(L, px, py));
                    if (li < n_act) {
                        PixelCtx c;
                        ctx_load(c, L, S.tps, px, py);
                        LdsSrc src{&S, (ty * 2 + pi) * VM_HALO_W + (tx * 2 + pj)};
                        c.tps_b = S.tpsb[src.hc + 2 * VM_HALO_W + 2];
                        float2 step;
#ifdef VM_PROF
                        unsigned long long ts[16];
#endif
                        uint32_t n_eval = 0;
                        const bool ok = wave_interior ? decide<true, SMAX, MINF>(L, P, src, c, sub, Lf, step, n_eval VM_TS_PASS)
                                                      : decide<false, SMAX, MINF>(L, P, src, c, sub, Lf, step, n_eval VM_TS_PASS);
                        if (sub == 0)
                            atomicAdd(&S.n_eval, n_eval);
                        if (ok && sub == 0) {
                            S.d_step[slot] = step;
                            S.d_ok[slot] = 1;
                        }
                    }
                }
                }
            }
            VM_TTS(pi * 2 + pj, 2);
            __syncthreads();
            VM_TTS(pi * 2 + pj, 3);

            // ---- 3. commits ----
            const bool ok = tid < 256 && commit_own(S, L, g, tid, ox, oy, pi, pj);
            {
                const unsigned long long cb = __ballot(ok);
                if (tid < 256 && (tid & 63) == 0) {
                    S.cbits[(tid >> 6) * 2] = (uint32_t)cb;
                    S.cbits[(tid >> 6) * 2 + 1] = (uint32_t)(cb >> 32);
                }
            }
            const int ncommit = __syncthreads_count(ok);
            VM_TTS(pi * 2 + pj, 4);
            if (ncommit) {
                tile_improving = true;
                st_commit += ncommit;
                for (int cell = tid; cell < VM_NCELL; cell += T) {
                    const int ry = cell / VM_HALO_W - 2, rx = cell % VM_HALO_W - 2; // tile-relative
                    const int qx = ox + rx, qy = oy + ry;
                    if (qx < 0 || qx >= L.w || qy < 0 || qy >= L.h)
                        continue;
                    // the committed pixels of this phase within +-2 of the cell, straight from the commit bitmap
                    // (bit tx of word ty): at most 3 x 3 slots, visited in gather_cell's order
                    int ylo = max(ry - 2, 0), xlo = max(rx - 2, 0);
                    const int yhi = min(ry + 2, VM_TILE_H - 1), xhi = min(rx + 2, VM_TILE_W - 1);
                    ylo += (ylo & 1) ^ pi;
                    xlo += (xlo & 1) ^ pj;
                    if (ylo > yhi || xlo > xhi)
                        continue;
                    const int sx0 = xlo >> 1, nx = ((xhi - xlo) >> 1) + 1, sy0 = ylo >> 1, ny = ((yhi - ylo) >> 1) + 1;
                    const uint32_t colmask = ((1u << nx) - 1u) << sx0; // nx <= 3
                    uint32_t rowbits[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        rowbits[t] = t < ny ? S.cbits[sy0 + t] & colmask : 0u;
                    if (!(rowbits[0] | rowbits[1] | rowbits[2]))
                        continue;
                    float2 m = S.mean[cell], q = S.var[cell], tb = S.tpsb[cell];
                    float cr = S.cross[cell];
                    gather_cell_bits(S, L, ox, oy, rx, ry, pi, pj, sy0, sx0, rowbits, m, q, cr, tb, P.commit_order);
                    if (!DENSE)
                        atomicOr(&S.dirty[cell >> 5], 1u << (cell & 31));
                    {
                        S.mean[cell] = m;
                        S.var[cell] = q;
                        S.cross[cell] = cr;
                        S.tpsb[cell] = tb;
                        const float counter = (float)(window_count(qy, L.h) * window_count(qx, L.w));
                        S.value[cell] = ssim_value(m.x, m.y, q.x, q.y, cr, counter, P.ssim_clamp);
                    }
                }
            }
            __syncthreads();
            VM_TTS(pi * 2 + pj, 5);
        }
    }
    VM_TTSF(2);

    // ---- SaveSSIM (morph.cu:1236-1256), tps.b and the owned mask words ----
    // (a pruned visit commits a pixel or two: only the cells those commits reached have changed)
    if (tile_improving) {
        for (int c = tid; c < VM_NCELL; c += T) {
            if (!DENSE && !((S.dirty[c >> 5] >> (c & 31)) & 1u))
                continue;
            int gx = ox - 2 + c % VM_HALO_W, gy = oy - 2 + c / VM_HALO_W;
            if (gx < 0 || gx >= L.w || gy < 0 || gy >= L.h)
                continue;
            int gi = gy * L.rs + gx;
            L.mean[gi] = S.mean[c];
            L.var[gi] = S.var[c];
            L.tps_b[gi] = S.tpsb[c];
            L.cross[gi] = S.cross[c];
            L.value[gi] = S.value[c];
        }
    }
    if (tid < g.nbx * g.nby) {
        int mx = tid % g.nbx, my = tid / g.nbx;
        // words owned by this tile: blocks that contain one of its pixels
        if (mx >= 1 && mx <= g.nbx - 2 && my >= 1 && my <= g.nby - 2) {
            const uint32_t wi = (uint32_t)((g.by0 + my + 1) * L.imp_rs + (g.bx0 + mx + 1)), wv = S.mask[my][mx];
            L.impmask[wi] = wv;
            if (sp_list && wv) {
                const uint32_t k = atomicAdd(sp_cnt, 1u);
                if (k < sp_cap) {
                    sp_list[k] = wi;
                    sp_val[k] = wv;
                }
            }
        }
    }
    improving = improving || tile_improving;
    VM_TTSF(3);
    return true;
}

// One stamp of the in-kernel clock probe: words [6] / [7] of an iteration's activity counters collect
// sum(exit - entry) of s_memtime (shader cycles) / s_memrealtime (constant 100 MHz) over the launches of the
// iteration, modulo 2^32 (MI355X_MICROARCH.md: in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz).
__device__ __forceinline__ void clock_probe(uint32_t *st, bool exit_stamp)
{
    const uint32_t c = (uint32_t)__builtin_amdgcn_s_memtime(), w = (uint32_t)__builtin_amdgcn_s_memrealtime();
    atomicAdd(&st[6], exit_stamp ? c : 0u - c);
    atomicAdd(&st[7], exit_stamp ? w : 0u - w);
}

// DENSE, SMAX = VM_SMAX: the 256-VGPR kernel of dense sweeps (one workgroup per CU);
// DENSE, SMAX = 7, MINF = 4: its 128-VGPR form -- every candidate gets >= 4 lanes, a phase of more than
// T / 4 candidates takes two rounds, two workgroups share a CU (four waves per SIMD instead of two: the
// dense line search is bound by the issue rate of a single wave, one VALU instruction per 4 cycles);
// !DENSE: the lean kernel of pruned sweeps.
// (The same 128-VGPR form as ONE 1024-thread workgroup per tile -- 4 lanes per candidate in a single
// round, for batches over levels of few tiles -- was measured slower than the 256-VGPR kernel at
// every batch size: 30 x 120x68 310 vs 280 us per pass, 30 x 240x135 970 vs 905; removed.)
template <bool DENSE, int SMAX = VM_SMAX, int MINF = VM_MIN_FANOUT, bool INTV = true>
__global__ __launch_bounds__(VM_SWEEP_T) __attribute__((amdgpu_waves_per_eu(DENSE && SMAX > 7 ? 1 : 4))) void SUF(k_optimize)(const VmLevelView *__restrict__ views, int cap,
                                                        VmKParams P, const uint32_t *__restrict__ tables,
                                                        int offx, int offy, uint32_t *__restrict__ flags,
                                                        uint32_t *__restrict__ stats, int iter_idx, int fixed_work,
                                                        const int *__restrict__ iter_dev)
{
    __shared__ TileLds S;
    // replayed from a hipGraph the launch can only carry the iteration's position inside the
    // graph: the base comes from a device counter that k_next_iter advances once per replay
    if (iter_dev)
        iter_idx += *iter_dev;
    const int tid = threadIdx.x, T = blockDim.x;
    // blockIdx.z = frame pair of the batch: same geometry, own state, own flags
    const VmLevelView L = views[blockIdx.z];
    flags += (size_t)blockIdx.z * cap;
    stats += (size_t)blockIdx.z * cap * VM_STAT_WORDS;

    // converged in the previous iteration: nothing left to do (sticky)
    if (!fixed_work && iter_idx > 0 && flags[iter_idx - 1] == 0)
        return;

    const int ox = blockIdx.x * VM_PITCH_X + offx, oy = blockIdx.y * VM_PITCH_Y + offy;
    if (ox >= L.w || oy >= L.h)
        return;
    if (tid == 0)
        S.n_eval = 0; // ordered before its first use by the barriers of tile_sweep
    // shader clock actually held under this kernel (bench.py: sclk_mhz_observed): the first tile of the first pair
    // brackets its sweep with s_memtime (shader cycles) and s_memrealtime (100 MHz); the differences accumulate in
    // stats words 6 / 7 -- minus the entry stamp now, plus the exit stamp later, so nothing stays live across the sweep
    const bool probe = DENSE && tid == 0 && (blockIdx.x | blockIdx.y | blockIdx.z) == 0;
    if (probe)
        clock_probe(stats + iter_idx * VM_STAT_WORDS, false);

    bool improving = false;
    uint32_t st_cand = 0, st_commit = 0;
    const bool swept = tile_sweep<DENSE, SMAX, MINF, INTV>(S, L, P, tables, false, ox, oy, tid, T, improving, st_cand, st_commit);
    if (probe)
        clock_probe(stats + iter_idx * VM_STAT_WORDS, true);
    if (!swept)
        return;
    if (tid == 0) {
        if (improving)
            flags[iter_idx] = 1u; // every writer stores the same 1: no atomic needed (see k_step)
        // per-iteration activity counters: active tiles, line searches, commits
        atomicAdd(&stats[iter_idx * VM_STAT_WORDS + 0], 1u);
        atomicAdd(&stats[iter_idx * VM_STAT_WORDS + 1], st_cand);
        atomicAdd(&stats[iter_idx * VM_STAT_WORDS + 2], st_commit);
        atomicAdd(&stats[iter_idx * VM_STAT_WORDS + 4], S.n_eval);
    }
}

// does the mask window of the tile at (ox, oy) hold block (bx, by)?  (mask_geom: the words
// k_optimize / tile_sweep test before doing anything)
__device__ __forceinline__ bool tile_window_has(const VmLevelView &L, int ox, int oy, int bx, int by)
{
    const MaskGeom g = mask_geom(L, ox, oy);
    return bx >= g.bx0 && bx < g.bx0 + g.nbx && by >= g.by0 && by < g.by0 + g.nby;
}

// ---------------------------------------------------------------------------
// The LISTED form of a pruned TILE pass (FAST, lean kernel, big batches).  A pass over a pruned level of a batch
// launches tiles x pairs workgroups of which a handful find a set mask bit near their tile; the rest cost nothing
// but their dispatch -- and that is the cost: measured ~4.7 ns per empty 512-thread workgroup, whatever it does
// before it returns (the chip starts about one wave per cycle): 30 pairs x 832 tiles of the 1080p level = 118 us
// per pass, 30 us at 960x540 (tools/dev_empty_tile_cost.py).  Here a scan (a few thousand waves) walks the mask
// words of every pair, marks the tiles of this pass a set bit reaches -- tile_sweep's own early-out test, which
// depends on nothing a tile of the same pass writes -- and appends each once to a list; the sweep then runs as a
// fixed grid of workgroups that take list entries in turn.  Same tiles, same state, any order: bit-identical.
// tl: [0 .. 4 cap) entry counters, one per iteration and pass of the call, then the stamps (one per pair and tile: the
// epoch of the scan that listed it last), then the entries (pair << 16 | tile); the host zeroes counters and stamps
// before every call (epochs restart there).
__global__ __launch_bounds__(256) void SUF(k_tile_scan)(const VmLevelView *__restrict__ views, int cap, int offx, int offy,
                                                        const uint32_t *__restrict__ flags, int iter_idx, int fixed_work,
                                                        const int *__restrict__ iter_dev, uint32_t *__restrict__ tl,
                                                        int tiles_stride)
{
    if (iter_dev)
        iter_idx += *iter_dev;
    const int pass = (offx ? 1 : 0) + (offy ? 2 : 0);
    const uint32_t epoch = (uint32_t)(iter_idx * 4 + pass) + 1u;
    const int slot = iter_idx * 4 + pass;
    const int z = blockIdx.z;
    if (!fixed_work && iter_idx > 0 && flags[(size_t)z * cap + iter_idx - 1] == 0)
        return; // converged in the previous iteration (sticky)
    const VmLevelView L = views[z];
    const int gx = (L.w + VM_PITCH_X - 1) / VM_PITCH_X, gy = (L.h + VM_PITCH_Y - 1) / VM_PITCH_Y;
    const int nwords = L.imp_rs * L.imp_rows;
    uint32_t *const stamp = tl + 4 * (size_t)cap + (size_t)z * tiles_stride;
    uint32_t *const entries = tl + 4 * (size_t)cap + (size_t)gridDim.z * tiles_stride;
    for (int wi = blockIdx.x * 256 + threadIdx.x; wi < nwords; wi += gridDim.x * 256) {
        const uint32_t wv = L.impmask[wi];
        if (!wv)
            continue;
        const int bx = wi % L.imp_rs - 1, by = wi / L.imp_rs - 1;
        const int ce = (5 * bx - offx) / VM_PITCH_X, re = (5 * by - offy) / VM_PITCH_Y;
        for (int r = max(re - 1, 0); r <= re + 1 && r < gy; ++r)
            for (int c = max(ce - 1, 0); c <= ce + 1 && c < gx; ++c) {
                const int ox = c * VM_PITCH_X + offx, oy = r * VM_PITCH_Y + offy;
                if (ox < L.w && oy < L.h && tile_window_has(L, ox, oy, bx, by) && (wv & tile_reach_bits(L, ox, oy, bx, by)) &&
                    atomicExch(&stamp[r * gx + c], epoch) != epoch) // first to name this tile
                    entries[atomicAdd(&tl[slot], 1u)] = (uint32_t)z << 16 | (uint32_t)(r * gx + c);
            }
    }
}

#if !VM_EXACT
__global__ __launch_bounds__(VM_SWEEP_T) __attribute__((amdgpu_waves_per_eu(4))) void SUF(k_optimize_listed)(
    const VmLevelView *__restrict__ views, int cap, VmKParams P, const uint32_t *__restrict__ tables, int offx, int offy,
    uint32_t *__restrict__ flags, uint32_t *__restrict__ stats, int iter_idx, const int *__restrict__ iter_dev,
    const uint32_t *__restrict__ tl, int tiles_stride, int nbatch)
{
    __shared__ TileLds S;
    if (iter_dev)
        iter_idx += *iter_dev;
    const int tid = threadIdx.x, T = blockDim.x;
    const int pass = (offx ? 1 : 0) + (offy ? 2 : 0);
    const uint32_t cnt = tl[iter_idx * 4 + pass];
    if (blockIdx.x >= cnt)
        return;
    const uint32_t *const entries = tl + 4 * (size_t)cap + (size_t)nbatch * tiles_stride;
    for (int k = tid; k < 625; k += T)
        S.tps[k] = __uint_as_float(tables[VM_TAB_TPS + k]);
    for (int k = tid; k < 225; k += T)
        S.imp[k] = tables[VM_TAB_IMP + k];
    for (uint32_t e = blockIdx.x; e < cnt; e += gridDim.x) {
        const uint32_t ent = entries[e];
        const int z = (int)(ent >> 16), t = (int)(ent & 0xFFFFu);
        const VmLevelView L = views[z];
        const int gx = (L.w + VM_PITCH_X - 1) / VM_PITCH_X;
        const int ox = (t % gx) * VM_PITCH_X + offx, oy = (t / gx) * VM_PITCH_Y + offy;
        if (tid == 0)
            S.n_eval = 0; // ordered before its first use by the barriers of tile_sweep
        bool improving = false;
        uint32_t st_cand = 0, st_commit = 0;
        if (tile_sweep<false>(S, L, P, tables, true, ox, oy, tid, T, improving, st_cand, st_commit) && tid == 0) {
            uint32_t *const st = stats + ((size_t)z * cap + iter_idx) * VM_STAT_WORDS;
            if (improving)
                flags[(size_t)z * cap + iter_idx] = 1u; // every writer stores the same 1
            atomicAdd(&st[0], 1u);
            atomicAdd(&st[1], st_cand);
            atomicAdd(&st[2], st_commit);
            atomicAdd(&st[4], S.n_eval);
        }
        __syncthreads(); // the tile's LDS is done with before the next entry stages its own
    }
}
#endif

// ===========================================================================
// SPARSE schedule: the pruned regime without kernel boundaries.  After the first sweeps of a
// level the improving mask leaves a handful of active tiles (none once the level has
// converged), yet the TILE schedule still pays four launches per iteration -- 3.4 us each over
// a 1080p level just to find every mask word zero -- and, in a batch, every pass waits for the
// slowest tile of ALL pairs.  Here ONE workgroup per frame pair walks a whole batch of
// iterations on the device: it keeps a list of the non-zero mask words, derives from it the
// tiles of the current pass whose mask window holds one (exactly the TILE kernel's early-out
// test), sweeps them one after the other with the same tile_sweep, and updates the list from
// the words those tiles own.  A converged level costs a few barriers per pass; pairs of a batch
// advance independently.  Tiles of one pass touch disjoint state, so their order is free: the
// results are bit-identical to the TILE schedule.
#define VM_SPARSE_LDS_CAP 1024
struct SparseLds {
    uint32_t tilebits[256]; // tiles of the current pass with a set mask word in their window
    int done[128];          // (list in memory) tiles swept in this pass, for the list update
    int ndone;
    uint32_t nnew;
    uint32_t wl[2][VM_SPARSE_LDS_CAP]; // (list in LDS) the non-zero mask words: current list and the next pass's
    uint32_t wv[2][VM_SPARSE_LDS_CAP]; // ... and their values
    int tl[64];                        // the tiles selected for this pass, in the order they were found
    uint32_t ntl;                      // how many (entries past 64 are only counted: the bitmap is walked instead)
    // resident visits (lean kernel): bounding box of the set mask bits (x0, x1, y0, y1); per pass parity and per
    // wave 0 / 1, which of the <= 4 real tiles of the pass a set bit reaches (bit 4: any set bit at all); 1 = a
    // commit left the safe rectangle
    int bb[4];
    uint32_t rw[2][2];
    uint32_t unsafe;
    // sv_phases, per phase and in two buffers (phases alternate: a phase without a mask hit has ONE barrier, so the
    // next phase's writers must not touch what a slow wave may still be reading): candidates and hits per slot wave,
    // the candidate bitmap and the commit bitmap (bit tx of word ty)
    int ph_cnt[2][4];
    uint32_t ph_cand[2][8], ph_commit[2][8];
};

// every non-zero mask word of the level into the pair's list 0
__global__ __launch_bounds__(256) void SUF(k_sparse_scan)(const VmLevelView *__restrict__ views)
{
    const VmLevelView L = views[blockIdx.z];
    const int nwords = L.imp_rs * L.imp_rows;
    const int wi = blockIdx.x * 256 + threadIdx.x;
    if (wi < nwords && L.impmask[wi] != 0) {
        const uint32_t idx = atomicAdd(&L.sp_cnt[0], 1u);
        L.sp_wl[idx] = (uint32_t)wi;
    }
}

#if !VM_EXACT
// ---------------------------------------------------------------------------
// The lean tile visit of the SPARSE schedule in pieces (round 4): load, four phases, store -- tile_sweep<false>
// taken apart so that the window sums of a small active region can STAY in LDS across the four passes of an
// iteration and across iterations (a RESIDENT visit).  A finest level that never converges is two or three
// pixels of one border row trading rounding-level moves (profiles/r04_notes.md): every pass has one tile over them,
// and each visit staged 1360 cells, walked four phases and wrote back, 2000 times.
//
// The LDS copy is a VIRTUAL tile: VM_TILE_W x VM_TILE_H pixels + halo at an origin (vx, vy) that is not on any
// pass's tile grid, placed around the bounding box of the set mask bits.  A real tile (ox, oy) of a pass is then
// swept inside it: its candidates are the mask hits of the virtual tile's slots that lie in the real tile's
// rectangle, its phase (pi, pj) is the virtual tile's phase (pi ^ (oy - vy) & 1, pj ^ (ox - vx) & 1), and everything
// a phase does -- line searches on the pre-phase sums, own-pixel commits, the per-cell gather in row-major
// order of the committing pixels, the mask bits -- is position-relative, so the bits are those of the real
// tile's own visit.  What makes it legal: (1) every candidate of any real tile (a pixel within +-2 of a set bit)
// lies inside the virtual tile -- all set bits sit in a SAFE rectangle, two pixels inside the virtual tile's edges
// (or at the level's edge), checked on entry and after every commit (a commit sets the bit of its pixel); a
// commit outside re-centres the virtual tile (store, load) before the next phase, and if the bits no longer fit
// one tile the visit goes on in the real tile's own frame and the kernel returns to list-driven visits;
// (2) nothing else touches the level meanwhile (one workgroup per pair).
struct SvTile {
    int vx, vy;
    MaskGeom g;
};
// the pixels' own state beside the window sums (tile + halo, as TileLds): a line search reads its pixel's v,
// lumas and UI terms and the v of its 8 ring neighbours from here, a commit writes here, and sv_store takes the
// committed pixels to memory -- no global load ahead of a search, no store acknowledgement ahead of a phase's
// barrier (measured on a cycling 1080p level: ~0.4 us and, in phases with a commit, ~1.4 us of every phase)
struct SvPix {
    float2 v[VM_NCELL], luma[VM_NCELL], uib[VM_NCELL];
    float uiaxy[VM_NCELL];
    uint32_t dirty[(VM_NCELL + 31) / 32];
};
struct SvNone {};
struct RingLds {
    const float2 *v;
    int pc; // the pixel's cell
    __device__ __forceinline__ float2 operator()(int k, int) const
    {
        // ring offsets (-1,-1) (0,-1) (1,-1) (1,0) (1,1) (0,1) (-1,1) (-1,0), two bits each (fover32)
        const int rx = ((0x06A4 >> (2 * k)) & 3) - 1, ry = ((0x6A40 >> (2 * k)) & 3) - 1;
        return v[pc + ry * VM_HALO_W + rx];
    }
};

__device__ __forceinline__ void sv_load_mask(TileLds &S, const VmLevelView &L, const MaskGeom &g, int tid)
{
    if (tid < g.nbx * g.nby) {
        const int mx = tid % g.nbx, my = tid / g.nbx;
        S.mask[my][mx] = L.impmask[(g.by0 + my + 1) * L.imp_rs + (g.bx0 + mx + 1)];
    }
}

// LoadSSIM (morph.cu:1214-1234) + the tile's tps.b, as in tile_sweep, + the pixels' own state; ends with a barrier
__device__ __forceinline__ void sv_load_state(TileLds &S, SvPix &X, const VmLevelView &L, int vx, int vy, int tid, int T)
{
    if (tid < (VM_NCELL + 31) / 32) {
        S.dirty[tid] = 0;
        X.dirty[tid] = 0;
    }
    for (int c0 = tid; c0 < VM_NCELL; c0 += 2 * T) {
        float2 m[2], q[2], tb[2], pv[2], pl[2], pu[2];
        float cr[2], val[2], pa[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = c0 + e * T;
            const int gx = vx - 2 + c % VM_HALO_W, gy = vy - 2 + c / VM_HALO_W;
            const bool in = c < VM_NCELL && gx >= 0 && gx < L.w && gy >= 0 && gy < L.h;
            const int gi = in ? gy * L.rs + gx : 0;
            m[e] = L.mean[gi];
            q[e] = L.var[gi];
            tb[e] = L.tps_b[gi];
            cr[e] = L.cross[gi];
            val[e] = L.value[gi];
            pv[e] = L.v[gi];
            pl[e] = L.luma[gi];
            pu[e] = L.ui_b[gi];
            pa[e] = L.ui_axy[gi];
            if (!in) {
                m[e] = q[e] = tb[e] = make_float2(0, 0);
                cr[e] = val[e] = 0.0f;
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = c0 + e * T;
            if (c < VM_NCELL) {
                S.mean[c] = m[e];
                S.var[c] = q[e];
                S.tpsb[c] = tb[e];
                S.cross[c] = cr[e];
                S.value[c] = val[e];
                X.v[c] = pv[e];
                X.luma[c] = pl[e];
                X.uib[c] = pu[e];
                X.uiaxy[c] = pa[e];
            }
        }
    }
    __syncthreads();
}

// SaveSSIM (morph.cu:1236-1256) of the cells a commit reached, tps.b, and the mask words the tile owns (blocks
// that contain one of its pixels); set words are appended to the pass's new word list when one is given
__device__ __forceinline__ void sv_store(const TileLds &S, const SvPix &X, const VmLevelView &L, const MaskGeom &g, int vx, int vy,
                                         int tid, int T, uint32_t *sp_list, uint32_t *sp_val, uint32_t *sp_cnt, uint32_t sp_cap)
{
    for (int c = tid; c < VM_NCELL; c += T) {
        const bool dc = (S.dirty[c >> 5] >> (c & 31)) & 1u, dp = (X.dirty[c >> 5] >> (c & 31)) & 1u;
        if (!dc && !dp)
            continue;
        const int gx = vx - 2 + c % VM_HALO_W, gy = vy - 2 + c / VM_HALO_W;
        if (gx < 0 || gx >= L.w || gy < 0 || gy >= L.h)
            continue;
        const int gi = gy * L.rs + gx;
        if (dc) {
            L.mean[gi] = S.mean[c];
            L.var[gi] = S.var[c];
            L.tps_b[gi] = S.tpsb[c];
            L.cross[gi] = S.cross[c];
            L.value[gi] = S.value[c];
        }
        if (dp) { // a committed pixel: commit_pixel_motion's own-pixel part (morph.cu:990-1026)
            L.v[gi] = X.v[c];
            L.luma[gi] = X.luma[c];
            L.ui_b[gi] = X.uib[c];
        }
    }
    if (tid < g.nbx * g.nby) {
        const int mx = tid % g.nbx, my = tid / g.nbx;
        if (mx >= 1 && mx <= g.nbx - 2 && my >= 1 && my <= g.nby - 2) {
            const uint32_t wi = (uint32_t)((g.by0 + my + 1) * L.imp_rs + (g.bx0 + mx + 1)), wv = S.mask[my][mx];
            L.impmask[wi] = wv;
            if (sp_list && wv) {
                const uint32_t k = atomicAdd(sp_cnt, 1u);
                if (k < sp_cap) {
                    sp_list[k] = wi;
                    sp_val[k] = wv;
                }
            }
        }
    }
}

// where a virtual tile goes for set bits in [x0, x1] x [y0, y1]: false if they do not fit one (every pixel within
// +-2 of a bit must be a pixel of the tile).  Margins are split evenly; the tile is kept inside the level, whose
// edges need no margin.
__device__ __forceinline__ bool sv_place(const VmLevelView &L, int x0, int x1, int y0, int y1, int &vx, int &vy)
{
    const int sx = VM_TILE_W - 5 - (x1 - x0), sy = VM_TILE_H - 5 - (y1 - y0);
    if (x0 > x1 || y0 > y1 || sx < 0 || sy < 0)
        return false;
    vx = min(max(x0 - 2 - sx / 2, 0), max(L.w - VM_TILE_W, 0));
    vy = min(max(y0 - 2 - sy / 2, 0), max(L.h - VM_TILE_H, 0));
    return true;
}

// is a set bit at (px, py) inside the safe rectangle of the virtual tile?
__device__ __forceinline__ bool sv_safe(const VmLevelView &L, const SvTile &V, int px, int py)
{
    return max(px - 2, 0) >= V.vx && min(px + 2, L.w - 1) <= min(V.vx + VM_TILE_W - 1, L.w - 1) && max(py - 2, 0) >= V.vy &&
           min(py + 2, L.h - 1) <= min(V.vy + VM_TILE_H - 1, L.h - 1);
}

// bounding box of the set bits of mask word `w` of block (bx, by) into Q.bb (LDS atomics)
__device__ __forceinline__ void sv_bbox_add(SparseLds &Q, uint32_t w, int bx, int by)
{
    if (!w)
        return;
    const uint32_t cols = (w | (w >> 5) | (w >> 10) | (w >> 15) | (w >> 20)) & 31u;
    uint32_t rows = 0;
#pragma unroll
    for (int r = 0; r < 5; ++r)
        rows |= ((w >> (5 * r)) & 31u) ? 1u << r : 0u;
    atomicMin(&Q.bb[0], 5 * bx + __ffs(cols) - 1);
    atomicMax(&Q.bb[1], 5 * bx + 31 - __clz(cols));
    atomicMin(&Q.bb[2], 5 * by + __ffs(rows) - 1);
    atomicMax(&Q.bb[3], 5 * by + 31 - __clz(rows));
}

// slot number of the k-th set bit (k from 0; it exists) of the phase's candidate bitmap: bit tx of word ty
__device__ __forceinline__ int sv_kth_slot(const uint32_t *cb, int k)
{
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        w[i] = cb[i]; // (one round trip)
    int sel = 0;
    uint32_t x = w[0];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int c = __popc(w[i]);
        if (k >= c && sel == i) {
            k -= c;
            sel = i + 1;
            x = w[i + 1];
        }
    }
    for (; k > 0; --k)
        x &= x - 1;
    return sel * 32 + __ffs(x) - 1;
}

// The four Jacobi phases of the real tile (ox, oy) inside the LDS copy at V (V = the real tile itself: a plain
// visit).  res: 0 = plain visit; 1 = resident; 2, 3 = resident, tests: every commit counts as unsafe, and (3) no
// re-centred tile is accepted.  Returns whether a pixel committed; `left` = the virtual tile was given up
// (V is now the real tile's own frame, stored by the caller like a plain visit).
// Three barriers per phase with a mask hit, one without (tile_sweep: five and two): the candidates are a
// bitmap (the four ballots of the slot threads; a searching wave picks its k-th set bit -- no compacted list),
// and the wave that decides a pixel also does its commit bookkeeping (mask bit, commit bitmap), so the cells
// gather right after the searches' barrier.  All mask reads of a phase (candidates) come before its first
// barrier, all mask writes after it.  The per-phase words (counts, candidate and commit bitmaps) exist twice and
// phases alternate between them: a phase without a mask hit is a single barrier, and the next phase's
// writers must not run into a wave that has not read this phase's counts yet.
__device__ __forceinline__ bool sv_phases(TileLds &S, SvPix &X, SparseLds &Q, const VmLevelView &L, const VmKParams &P, SvTile &V,
                                          int ox, int oy, int tid, int T, uint32_t &st_cand, uint32_t &st_commit, int res,
                                          bool &left)
{
    const int rx1 = min(ox + VM_TILE_W - 1, L.w - 1), ry1 = min(oy + VM_TILE_H - 1, L.h - 1);
    bool tile_improving = false;
    for (int pi = 0; pi < 2; ++pi) {
        for (int pj = 0; pj < 2; ++pj) {
            const int vx = V.vx, vy = V.vy;
            const int vpi = pi ^ ((oy - vy) & 1), vpj = pj ^ ((ox - vx) & 1);
            int *const ph_cnt = Q.ph_cnt[pj];
            uint32_t *const ph_cand = Q.ph_cand[pj], *const ph_commit = Q.ph_commit[pj];
            VM_TTS(pi * 2 + pj, 0);
            // ---- 1. candidates of this phase ----
            bool cand = false, hit = false;
            uint32_t *mword = nullptr;
            uint32_t mbit = 0;
            if (tid < 256) {
                const int px = vx + (tid & 31) * 2 + vpj, py = vy + (tid >> 5) * 2 + vpi;
                if (px >= ox && px <= rx1 && py >= oy && py <= ry1 && mask_hit(S.mask, S.imp, V.g, px, py)) {
                    hit = true; // in the mask: its bit is cleared unless it commits
                    cand = !pixel_locked(L, P.bcond, px, py);
                    mword = &S.mask[py / 5 - V.g.by0][px / 5 - V.g.bx0];
                    mbit = 1u << ((px % 5) + (py % 5) * 5);
                }
                const unsigned long long bc = __ballot(cand), bh = __ballot(hit);
                if ((tid & 63) == 0) {
                    ph_cand[(tid >> 6) * 2] = (uint32_t)bc; // the candidate bitmap: bit tx of word ty
                    ph_cand[(tid >> 6) * 2 + 1] = (uint32_t)(bc >> 32);
                    ph_cnt[tid >> 6] = __popcll(bc) | (bh ? 1 << 16 : 0);
                    ph_commit[(tid >> 6) * 2] = 0; // the commit bitmap of this phase
                    ph_commit[(tid >> 6) * 2 + 1] = 0;
                }
            }
            __syncthreads();
            const int w0 = ph_cnt[0], w1 = ph_cnt[1], w2 = ph_cnt[2], w3 = ph_cnt[3];
            // no pixel of this phase in the mask: nothing to search, nothing to commit, no bit to clear
            if (!((w0 | w1 | w2 | w3) >> 16))
                continue;
            const int n_act = (w0 & 0xFFFF) + (w1 & 0xFFFF) + (w2 & 0xFFFF) + (w3 & 0xFFFF);
            if (hit && !cand)
                atomicAnd(mword, ~mbit); // a locked pixel in the mask
            VM_TTS(pi * 2 + pj, 1);
            if (n_act > 0) {
                st_cand += n_act;
                // ---- 2. line searches on the pre-phase state: the lean search, a whole wave per candidate
                // (two points of the search per round) while the workgroup has that many waves ----
                const bool wide = n_act * 64 <= T;
                for (int base = 0; base < n_act; base += wide ? T / 64 : T / 32) {
                    const int li = base + (wide ? tid >> 6 : tid >> 5), sub = tid & 31;
                    const bool writer = wide ? (tid & 63) == 0 : sub == 0;
                    const int slot = sv_kth_slot(ph_cand, min(li, n_act - 1));
                    const int tx = slot & 31, ty = slot >> 5;
                    const int px = vx + tx * 2 + vpj, py = vy + ty * 2 + vpi;
                    const bool wave_interior = __all(li >= n_act || is_interior(L, px, py));
                    if (li < n_act) {
                        LdsSrc src{&S, (ty * 2 + vpi) * VM_HALO_W + (tx * 2 + vpj)};
                        const int pc = src.hc + 2 * VM_HALO_W + 2; // the pixel's own cell
                        PixelCtx c; // (ctx_load, from the LDS copy)
                        c.px = px;
                        c.py = py;
                        c.idx = py * L.rs + px;
                        c.v = X.v[pc];
                        c.old_luma = X.luma[pc];
                        c.ui_axy = X.uiaxy[pc];
                        c.ui_b = X.uib[pc];
                        c.tps_axy = S.tps[(border_class(py, L.h) * 5 + border_class(px, L.w)) * 25 + 12] / 2;
                        c.tref = make_float2(0, 0);
                        c.tmask = 0.0f;
                        if (L.temp_mask) { // uniform in the launch
                            c.tref = L.temp_ref[c.idx];
                            c.tmask = L.temp_mask[c.idx];
                        }
                        c.tps_b = S.tpsb[pc];
                        const RingLds ring{X.v, pc};
                        VM_TTS(pi * 2 + pj, 6);
                        float2 step, luma;
#ifdef VM_PROF
                        unsigned long long ts[16];
#endif
                        Nb1 nb;
                        bool ok;
                        uint32_t n_eval = 0;
                        if (wave_interior) {
                            nb1_load<true>(nb, L, src, c, sub);
                            ok = wide ? decide64<true>(L, P, nb, ring, c, sub, (tid & 32) != 0, step, luma, n_eval)
                                      : decide32<true>(L, P, nb, ring, c, sub, step, luma, n_eval VM_TS_PASS);
                        } else {
                            nb1_load<false>(nb, L, src, c, sub);
                            ok = wide ? decide64<false>(L, P, nb, ring, c, sub, (tid & 32) != 0, step, luma, n_eval)
                                      : decide32<false>(L, P, nb, ring, c, sub, step, luma, n_eval VM_TS_PASS);
                        }
                        if (writer) {
                            atomicAdd(&S.n_eval, n_eval);
                            uint32_t *const mw = &S.mask[py / 5 - V.g.by0][px / 5 - V.g.bx0];
                            const uint32_t mb = 1u << ((px % 5) + (py % 5) * 5);
                            if (ok) {
                                // commit_pixel_motion (morph.cu:990-1026), the pixel's own part, at once: nothing
                                // else of this phase reads its v, luma or ui.b; its record for the cells' gather;
                                // its mask bit
                                const float2 ol = c.old_luma;
                                S.d_step[slot] = step;
                                S.d_mean[slot] = make_float2(luma.x - ol.x, luma.y - ol.y);
                                S.d_var[slot] = make_float2(luma.x * luma.x - ol.x * ol.x, luma.y * luma.y - ol.y * ol.y);
                                S.d_cross[slot] = luma.x * luma.y - ol.x * ol.y;
                                X.luma[pc] = luma;
                                X.uib[pc] = make_float2(c.ui_b.x + 2 * step.x * c.ui_axy, c.ui_b.y + 2 * step.y * c.ui_axy);
                                X.v[pc] = make_float2(c.v.x + step.x, c.v.y + step.y);
                                atomicOr(&X.dirty[pc >> 5], 1u << (pc & 31));
                                atomicOr(&ph_commit[ty], 1u << tx);
                                atomicOr(mw, mb);
                                if (res && (res > 1 || !sv_safe(L, V, px, py)))
                                    Q.unsafe = 1u;
                            } else
                                atomicAnd(mw, ~mb);
                        }
                    }
                }
            }
            VM_TTS(pi * 2 + pj, 2);
            __syncthreads();
            VM_TTS(pi * 2 + pj, 3);

            // ---- 3. the cells gather the commits of the phase ----
            uint32_t cb[8];
            int ncommit = 0, ty0 = 8, ty1 = -1;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                cb[k] = ph_commit[k];
                ncommit += __popc(cb[k]);
                if (cb[k]) {
                    ty0 = min(ty0, k);
                    ty1 = k;
                }
            }
            VM_TTS(pi * 2 + pj, 4);
            if (ncommit) {
                tile_improving = true;
                st_commit += ncommit;
                // (rows of the halo grid within +-2 of a committing pixel: 2 ty + vpi - 2 ... + 2, shifted by the halo's 2)
                for (int cell = (2 * ty0 + vpi) * VM_HALO_W + tid; cell < (2 * ty1 + vpi + 5) * VM_HALO_W; cell += T) {
                    const int ry = cell / VM_HALO_W - 2, rx = cell % VM_HALO_W - 2; // tile-relative
                    const int qx = vx + rx, qy = vy + ry;
                    if (qx < 0 || qx >= L.w || qy < 0 || qy >= L.h)
                        continue;
                    int ylo = max(ry - 2, 0), xlo = max(rx - 2, 0);
                    const int yhi = min(ry + 2, VM_TILE_H - 1), xhi = min(rx + 2, VM_TILE_W - 1);
                    ylo += (ylo & 1) ^ vpi;
                    xlo += (xlo & 1) ^ vpj;
                    if (ylo > yhi || xlo > xhi)
                        continue;
                    const int sx0 = xlo >> 1, nx = ((xhi - xlo) >> 1) + 1, sy0 = ylo >> 1, ny = ((yhi - ylo) >> 1) + 1;
                    const uint32_t colmask = ((1u << nx) - 1u) << sx0; // nx <= 3
                    uint32_t rowbits[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        rowbits[t] = t < ny ? ph_commit[sy0 + t] & colmask : 0u;
                    if (!(rowbits[0] | rowbits[1] | rowbits[2]))
                        continue;
                    float2 m = S.mean[cell], q = S.var[cell], tb = S.tpsb[cell];
                    float cr = S.cross[cell];
                    gather_cell_bits(S, L, vx, vy, rx, ry, vpi, vpj, sy0, sx0, rowbits, m, q, cr, tb, P.commit_order);
                    atomicOr(&S.dirty[cell >> 5], 1u << (cell & 31));
                    S.mean[cell] = m;
                    S.var[cell] = q;
                    S.cross[cell] = cr;
                    S.tpsb[cell] = tb;
                    const float counter = (float)(window_count(qy, L.h) * window_count(qx, L.w));
                    S.value[cell] = ssim_value(m.x, m.y, q.x, q.y, cr, counter, P.ssim_clamp);
                }
            }
            __syncthreads();
            VM_TTS(pi * 2 + pj, 5);
            if (!res || !(Q.unsafe & 1u)) // (uniform: written before the barrier above)
                continue;
            // ---- a commit left the safe rectangle: move the virtual tile before the next phase ----
            sv_store(S, X, L, V.g, vx, vy, tid, T, nullptr, nullptr, nullptr, 0);
            if (tid == 0) {
                Q.bb[0] = Q.bb[2] = 0x7fffffff;
                Q.bb[1] = Q.bb[3] = -0x7fffffff;
            }
            __syncthreads(); // the cells and words are out; nobody still reads Q.unsafe
            if (tid < V.g.nbx * V.g.nby)
                sv_bbox_add(Q, S.mask[tid / V.g.nbx][tid % V.g.nbx], V.g.bx0 + tid % V.g.nbx, V.g.by0 + tid / V.g.nbx);
            if (tid == 0)
                Q.unsafe = 0;
            __syncthreads();
            int nvx, nvy;
            if (res == 3 || !sv_place(L, Q.bb[0], Q.bb[1], Q.bb[2], Q.bb[3], nvx, nvy)) {
                nvx = ox; // the rest of this visit in the real tile's own frame, then back to list-driven visits
                nvy = oy;
                left = true;
                res = 0;
            }
            V.vx = nvx;
            V.vy = nvy;
            V.g = mask_geom(L, nvx, nvy);
            __syncthreads(); // S.mask and Q.bb are read before they are overwritten
            sv_load_mask(S, L, V.g, tid);
            sv_load_state(S, X, L, nvx, nvy, tid, T);
        }
    }
    return tile_improving;
}
#endif // !VM_EXACT

// The word list lives in LDS while it fits (lds_cap <= VM_SPARSE_LDS_CAP entries; round 4): building the
// tile set of a pass, carrying the list over and adding the words of the swept tiles then touch no memory
// at all.  (In memory the same steps were three chains of dependent loads -- list entry -> mask word ->
// stamp -- per pass: ~10 us of the ~25 us a pass of a cycling level took.)  A word of the list can only
// have changed if a swept tile owns it (at most one tile of a pass owns a word: the 5-pixel gaps are as
// wide as a block); such entries are dropped and come back from tile_sweep, which appends the set words a
// tile owns as it writes them back.  Should the list outgrow LDS, the kernel rescans the level into the
// lists in memory and goes on there (lds_cap = 0 forces that path: tests).
// res_mode (lean kernel): 0 = resident visits where the set bits fit one tile, 1 = never; 2, 3 = tests (sv_phases)
template <bool DENSE>
__global__ __launch_bounds__(VM_SWEEP_T) __attribute__((amdgpu_waves_per_eu(1))) void SUF(k_sparse)(
    const VmLevelView *__restrict__ views, int cap, VmKParams P, const uint32_t *__restrict__ tables,
    uint32_t *__restrict__ flags, uint32_t *__restrict__ stats, int it0, int nit, int fixed_work, int lds_cap, int res_mode)
{
#if VM_EXACT
    constexpr bool LEAN = false;
#else
    constexpr bool LEAN = !DENSE; // the visit from its pieces (sv_*): plain or resident
#endif
    __shared__ TileLds S;
    __shared__ SparseLds Q;
#if !VM_EXACT
    __shared__ typename std::conditional<LEAN, SvPix, SvNone>::type X;
#endif
    const int tid = threadIdx.x, T = blockDim.x;
    const VmLevelView L = views[blockIdx.z];
    flags += (size_t)blockIdx.z * cap;
    stats += (size_t)blockIdx.z * cap * VM_STAT_WORDS;
    if (!fixed_work && it0 > 0 && flags[it0 - 1] == 0)
        return; // converged in the previous iteration (sticky)
    for (int k = tid; k < 625; k += T)
        S.tps[k] = __uint_as_float(tables[VM_TAB_TPS + k]);
    for (int k = tid; k < 225; k += T)
        S.imp[k] = tables[VM_TAB_IMP + k];
    const int gx = (L.w + VM_PITCH_X - 1) / VM_PITCH_X, gy = (L.h + VM_PITCH_Y - 1) / VM_PITCH_Y;
    const int ntw = (gx * gy + 31) / 32; // <= 256 (the host checks)
    const int nwords = L.imp_rs * L.imp_rows;
    uint32_t *const lists[2] = {L.sp_wl, L.sp_wl + nwords};
    const uint32_t lcap = (uint32_t)min(max(lds_cap, 0), VM_SPARSE_LDS_CAP);
    int cur = 0;
    uint32_t nw = L.sp_cnt[0]; // length of the current list (from k_sparse_scan); then carried in registers
    bool in_lds = nw <= lcap;   // uniform
    if (in_lds)
        for (uint32_t k = tid; k < nw; k += T) {
            const uint32_t wi = L.sp_wl[k];
            Q.wl[0][k] = wi;
            Q.wv[0][k] = L.impmask[wi];
        }
    __syncthreads();
    // every set word of the level into list 0 in memory, and on from there (the LDS list has been outgrown, or
    // resident visits were given up)
    auto rescan_into_memory = [&]() {
        __syncthreads();
        if (tid == 0)
            Q.nnew = 0;
        __syncthreads();
        for (int wi = tid; wi < nwords; wi += T)
            if (L.impmask[wi] != 0)
                lists[0][atomicAdd(&Q.nnew, 1u)] = (uint32_t)wi;
        __syncthreads();
        nw = Q.nnew;
        in_lds = false;
        for (int wi = tid; wi < nwords; wi += T)
            L.sp_stamp[wi] = 0; // stamps are epochs of the passes walked in memory
    };
#if !VM_EXACT
    bool resident = false; // the window sums around the set bits stay in LDS (sv_phases); the word list rests
    SvTile V{0, 0, MaskGeom{0, 0, 0, 0}};
#else
    constexpr bool resident = false;
#endif
    for (int it = it0; it < it0 + nit; ++it) {
        // no set mask word anywhere: no tile of any pass of any later iteration can be active --
        // every remaining sweep of this batch is a no-op (its flags and counters stay zero)
        if (nw == 0)
            break;
        bool improving = false, emptied = false;
        uint32_t st_cand = 0, st_commit = 0, st_tiles = 0, st_res = 0;
        if (tid == 0)
            S.n_eval = 0;
#if !VM_EXACT
        if constexpr (LEAN) {
            // a few set words whose bits fit one tile: go resident
            if (!resident && in_lds && res_mode != 1 && nw <= 64) {
                if (tid == 0) {
                    Q.bb[0] = Q.bb[2] = 0x7fffffff;
                    Q.bb[1] = Q.bb[3] = -0x7fffffff;
                    Q.unsafe = 0;
                }
                __syncthreads();
                if ((uint32_t)tid < nw) {
                    const uint32_t wi = Q.wl[cur][tid];
                    sv_bbox_add(Q, Q.wv[cur][tid], (int)(wi % (uint32_t)L.imp_rs) - 1, (int)(wi / (uint32_t)L.imp_rs) - 1);
                }
                __syncthreads();
                if (sv_place(L, Q.bb[0], Q.bb[1], Q.bb[2], Q.bb[3], V.vx, V.vy)) {
                    V.g = mask_geom(L, V.vx, V.vy);
                    sv_load_mask(S, L, V.g, tid);
                    sv_load_state(S, X, L, V.vx, V.vy, tid, T);
                    resident = true;
                }
            }
        }
#endif
        for (int pass = 0; pass < 4; ++pass) {
            const int offx = (pass & 1) ? VM_TILE_W : 0, offy = (pass & 2) ? VM_TILE_H : 0; // morph.cu:1382-1385
            const uint32_t *list = in_lds ? Q.wl[cur] : lists[cur];
            const bool pass_resident = resident; // (residency can end inside a pass, never begin)
            VM_TTSF(4);
            // ---- 1. the tiles of this pass that a set mask bit reaches ----
            int ntl = 0;
            unsigned long long rtl = 0; // (resident) the <= 4 tiles, 16 bits each
            bool selected = false;
#if !VM_EXACT
            if constexpr (LEAN) if (pass_resident) {
                selected = true;
                // every set bit of the level is in the LDS copy of the words: the <= 2 x 2 tiles of this pass whose
                // rectangle + 2 meets the virtual tile, tested as tile_sweep's early out tests them
                const int ax = V.vx - (VM_TILE_W + 1) - offx, bx = V.vx + (VM_TILE_W + 1) - offx;
                const int ay = V.vy - (VM_TILE_H + 1) - offy, by = V.vy + (VM_TILE_H + 1) - offy;
                const int c_lo = ax > 0 ? (ax + VM_PITCH_X - 1) / VM_PITCH_X : 0, r_lo = ay > 0 ? (ay + VM_PITCH_Y - 1) / VM_PITCH_Y : 0;
                const int c_hi = bx >= 0 ? min(bx / VM_PITCH_X, gx - 1) : -1, r_hi = by >= 0 ? min(by / VM_PITCH_Y, gy - 1) : -1;
                if (tid < 128) { // the <= 96 words of a window sit in waves 0 and 1
                    uint32_t mine = 0;
                    if (tid < V.g.nbx * V.g.nby) {
                        const int mx = tid % V.g.nbx, my = tid / V.g.nbx;
                        const uint32_t w = S.mask[my][mx];
                        if (w) {
                            mine = 16u; // a set bit exists
                            for (int r = r_lo; r <= r_hi; ++r)
                                for (int c = c_lo; c <= c_hi; ++c) {
                                    const int ox = c * VM_PITCH_X + offx, oy = r * VM_PITCH_Y + offy;
                                    if (ox < L.w && oy < L.h && (w & tile_reach_bits(L, ox, oy, V.g.bx0 + mx, V.g.by0 + my)))
                                        mine |= 1u << ((r - r_lo) * 2 + (c - c_lo));
                                }
                        }
                    }
                    uint32_t m = 0;
#pragma unroll
                    for (int bnum = 0; bnum < 5; ++bnum)
                        m |= __ballot((mine >> bnum) & 1u) ? 1u << bnum : 0u;
                    if ((tid & 63) == 0)
                        Q.rw[pass & 1][tid >> 6] = m; // (two buffers: a pass without a visit has no barrier after its read)
                }
                __syncthreads();
                const uint32_t reach = Q.rw[pass & 1][0] | Q.rw[pass & 1][1];
                if (!(reach & 16u)) {
                    // no set bit left: the level has converged (what `nw == 0` says to list-driven visits)
                    if (pass == 0) {
                        sv_store(S, X, L, V.g, V.vx, V.vy, tid, T, nullptr, nullptr, nullptr, 0);
                        resident = false;
                        nw = 0;
                        emptied = true;
                        break;
                    }
                    continue;
                }
                ntl = 0;
                for (int r = r_lo; r <= r_hi; ++r)
                    for (int c = c_lo; c <= c_hi; ++c)
                        if ((reach >> ((r - r_lo) * 2 + (c - c_lo))) & 1u)
                            rtl |= (unsigned long long)(r * gx + c) << (16 * ntl++);
            }
#endif
            if (!selected) {
                for (int k = tid; k < ntw; k += T)
                    Q.tilebits[k] = 0;
                if (tid == 0) {
                    Q.ndone = 0;
                    Q.nnew = 0;
                    Q.ntl = 0;
                }
                __syncthreads();
                for (uint32_t k = tid; k < nw; k += T) {
                    const int wi = (int)list[k];
                    const uint32_t wv = in_lds ? Q.wv[cur][k] : L.impmask[wi];
                    const int bx = wi % L.imp_rs - 1, by = wi / L.imp_rs - 1;
                    const int ce = (5 * bx - offx) / VM_PITCH_X, re = (5 * by - offy) / VM_PITCH_Y;
                    for (int r = max(re - 1, 0); r <= re + 1 && r < gy; ++r)
                        for (int c = max(ce - 1, 0); c <= ce + 1 && c < gx; ++c) {
                            const int ox = c * VM_PITCH_X + offx, oy = r * VM_PITCH_Y + offy;
                            // (tile_sweep's early-out test: a set bit within +-2 of the tile)
                            if (ox < L.w && oy < L.h && tile_window_has(L, ox, oy, bx, by) && (wv & tile_reach_bits(L, ox, oy, bx, by))) {
                                const uint32_t bit = 1u << ((r * gx + c) & 31);
                                if (!(atomicOr(&Q.tilebits[(r * gx + c) >> 5], bit) & bit)) { // first to name this tile
                                    const uint32_t q = atomicAdd(&Q.ntl, 1u);
                                    if (q < 64)
                                        Q.tl[q] = r * gx + c;
                                }
                            }
                        }
                }
                __syncthreads();
                ntl = (int)Q.ntl;
            }
            VM_TTSF(5);
            // ---- 2. sweep them, one after the other (tiles of a pass touch disjoint state: any order) ----
            // (a few tiles: straight from the list; many: the bitmap, word by word -- walking all of its up to 46
            // words for the one tile of a cycling level was ~3 us of every pass)
            for (int wd = 0; wd < (ntl <= 64 ? ntl : ntw); ++wd) {
                uint32_t bits = ntl <= 64 ? 1u : Q.tilebits[wd];
                while (bits) {
                    const int t = pass_resident ? (int)((rtl >> (16 * wd)) & 0xFFFFu) : (ntl <= 64 ? Q.tl[wd] : wd * 32 + __ffs(bits) - 1);
                    bits &= bits - 1;
                    const int ox = (t % gx) * VM_PITCH_X + offx, oy = (t / gx) * VM_PITCH_Y + offy;
                    bool visited;
#if !VM_EXACT
                    if constexpr (LEAN) {
                        // plain visit: tile_sweep from its pieces; resident visit: the phases only
                        visited = true;
                        if (!resident) {
                            V = SvTile{ox, oy, mask_geom(L, ox, oy)};
                            uint32_t reach = 0;
                            if (tid < V.g.nbx * V.g.nby) {
                                const int mx = tid % V.g.nbx, my = tid / V.g.nbx;
                                const uint32_t w = L.impmask[(V.g.by0 + my + 1) * L.imp_rs + (V.g.bx0 + mx + 1)];
                                S.mask[my][mx] = w;
                                reach = w & tile_reach_bits(L, ox, oy, V.g.bx0 + mx, V.g.by0 + my);
                            }
                            visited = __syncthreads_or(reach != 0); // tile_sweep's early out
                            if (visited)
                                sv_load_state(S, X, L, ox, oy, tid, T);
                        }
                        if (visited) {
                            bool left = false;
                            st_res += resident ? 1u : 0u;
                            if (sv_phases(S, X, Q, L, P, V, ox, oy, tid, T, st_cand, st_commit, resident ? (res_mode ? res_mode : 1) : 0, left))
                                improving = true;
                            if (!resident || left) {
                                const bool listed = !pass_resident && in_lds;
                                sv_store(S, X, L, V.g, V.vx, V.vy, tid, T, listed ? Q.wl[cur ^ 1] : nullptr, Q.wv[cur ^ 1], &Q.nnew, lcap);
                                resident = false;
                            }
                        }
                    } else
#endif
                        visited = tile_sweep<DENSE>(S, L, P, tables, true, ox, oy, tid, T, improving, st_cand, st_commit,
                                                    in_lds ? Q.wl[cur ^ 1] : nullptr, Q.wv[cur ^ 1], &Q.nnew, lcap);
                    if (visited) {
                        ++st_tiles;
                        if (tid == 0 && !pass_resident) {
                            if (Q.ndone < 128)
                                Q.done[Q.ndone] = t;
                            ++Q.ndone;
                        }
                    }
                    __syncthreads(); // the tile's state and mask words are out before anything reads them
                }
            }
            VM_TTSF(6);
            if (resident)
                continue; // the LDS copy of the words is the list
            if (pass_resident) {
                // residency was given up in this pass: the list again, from a scan of the level
                rescan_into_memory();
                cur = 0;
                __syncthreads();
                continue;
            }
            // ---- 3. the list for the next pass: old entries that are still set, plus the set
            // words the swept tiles own (no other word can have changed) ----
            if (in_lds) {
                // an entry owned by a swept tile has been dealt with by that tile (appended again if still
                // set); every other entry is unchanged, hence still set
                uint32_t *nl = Q.wl[cur ^ 1];
                for (uint32_t k = tid; k < nw; k += T) {
                    const uint32_t wi = list[k];
                    const int bx = (int)(wi % (uint32_t)L.imp_rs) - 1, by = (int)(wi / (uint32_t)L.imp_rs) - 1;
                    // the tile of this pass that holds a pixel of block (bx, by), if any
                    bool swept = false;
                    const int nx = 5 * bx + 4 - offx, ny = 5 * by + 4 - offy;
                    if (nx >= 0 && ny >= 0) {
                        const int c = nx / VM_PITCH_X, r = ny / VM_PITCH_Y;
                        const int ox = c * VM_PITCH_X + offx, oy = r * VM_PITCH_Y + offy;
                        if (c < gx && r < gy && ox < L.w && oy < L.h && 5 * bx <= min(ox + VM_TILE_W - 1, L.w - 1) &&
                            5 * by <= min(oy + VM_TILE_H - 1, L.h - 1))
                            swept = (Q.tilebits[(r * gx + c) >> 5] >> ((r * gx + c) & 31)) & 1u;
                    }
                    if (!swept) {
                        const uint32_t q = atomicAdd(&Q.nnew, 1u);
                        if (q < lcap) {
                            nl[q] = wi;
                            Q.wv[cur ^ 1][q] = Q.wv[cur][k];
                        }
                    }
                }
                __syncthreads();
                nw = Q.nnew;
                if (nw > lcap) { // outgrown
                    rescan_into_memory();
                    cur = 1; // (flipped to 0 below)
                }
                cur ^= 1;
                __syncthreads(); // Q.nnew is reset at the top of the next pass
                VM_TTSF(7);
                continue;
            }
            uint32_t *nlist = lists[cur ^ 1];
            const uint32_t epoch = (uint32_t)(it * 4 + pass) + 1u;
            for (uint32_t k = tid; k < nw; k += T) {
                const uint32_t wi = list[k];
                if (L.impmask[wi] != 0) {
                    L.sp_stamp[wi] = epoch;
                    nlist[atomicAdd(&Q.nnew, 1u)] = wi;
                }
            }
            __syncthreads();
            const int nd = Q.ndone;
            if (nd > 128) { // more tiles than the list of swept ones holds: rescan the level
                for (int wi = tid; wi < nwords; wi += T)
                    if (L.impmask[wi] != 0 && L.sp_stamp[wi] != epoch) {
                        L.sp_stamp[wi] = epoch;
                        nlist[atomicAdd(&Q.nnew, 1u)] = (uint32_t)wi;
                    }
            } else {
                for (int d = 0; d < nd; ++d) {
                    const int t = Q.done[d];
                    const int ox = (t % gx) * VM_PITCH_X + offx, oy = (t / gx) * VM_PITCH_Y + offy;
                    const MaskGeom g = mask_geom(L, ox, oy);
                    if (tid < g.nbx * g.nby) {
                        const int mx = tid % g.nbx, my = tid / g.nbx;
                        if (mx >= 1 && mx <= g.nbx - 2 && my >= 1 && my <= g.nby - 2) { // words the tile owns
                            const int wi = (g.by0 + my + 1) * L.imp_rs + (g.bx0 + mx + 1);
                            if (L.impmask[wi] != 0 && L.sp_stamp[wi] != epoch) {
                                L.sp_stamp[wi] = epoch;
                                nlist[atomicAdd(&Q.nnew, 1u)] = (uint32_t)wi;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            nw = Q.nnew;
            cur ^= 1;
            __syncthreads(); // Q.nnew is reset at the top of the next pass
        }
        if (emptied)
            break;
        if (tid == 0) { // one writer per pair and iteration; the host zeroed the arrays
            flags[it] = improving ? 1u : 0u;
            stats[it * VM_STAT_WORDS + 0] = st_tiles;
            stats[it * VM_STAT_WORDS + 1] = st_cand;
            stats[it * VM_STAT_WORDS + 2] = st_commit;
            stats[it * VM_STAT_WORDS + 4] = S.n_eval;
            stats[it * VM_STAT_WORDS + 5] = st_res; // tile visits served from the resident LDS copy (diagnostic)
        }
        if (!fixed_work && !improving)
            break; // reference semantics: the level stops here (the following flags stay 0)
        __syncthreads();
    }
#if !VM_EXACT
    if constexpr (LEAN) if (resident)
        sv_store(S, X, L, V.g, V.vx, V.vy, tid, T, nullptr, nullptr, nullptr, 0);
#endif
}

// ===========================================================================
// SPLIT schedule.  k_decide runs the line searches of one phase and, for an accepted
// step, writes the pixel's own state at once (v, luma, ui.b: nothing else of the same
// phase reads them -- the fold-over ring of a pixel has no pixel of its own parity) plus
// a decision record; k_commit folds the records into the window sums of the cells they
// reach.  Records, two float4 per pixel:
//   rec_tag = epoch << 2 | state      (state 1: commit, 2: mask hit that did not move)
//   rec_a = (d mean.x, d mean.y, d var.x, d var.y),  rec_b = (d cross, step.x, step.y, -)
// valid only for the current epoch (one epoch per phase, so nothing is ever cleared).

__global__ __launch_bounds__(VM_SWEEP_T) void SUF(k_decide)(const VmLevelView *__restrict__ views, int cap, VmKParams P,
                                                      const uint32_t *__restrict__ tables, int offx, int offy,
                                                      int pi, int pj, int parts, uint32_t epoch,
                                                      const uint32_t *__restrict__ flags,
                                                      uint32_t *__restrict__ stats, int iter_idx,
                                                      int fixed_work)
{
    __shared__ SplitLds S;
    const int tid = threadIdx.x, T = blockDim.x;
#ifdef VM_PROF
    unsigned long long ts[16];
    for (int k = 0; k < 16; ++k) ts[k] = 0;
    VM_TS(0);
#endif
    const VmLevelView L = views[blockIdx.z];
    flags += (size_t)blockIdx.z * cap;
    stats += (size_t)blockIdx.z * cap * VM_STAT_WORDS;
    if (!fixed_work && iter_idx > 0 && flags[iter_idx - 1] == 0)
        return;
    const int part = blockIdx.x % parts, tile = blockIdx.x / parts;
    const int gxn = (L.w + VM_PITCH_X - 1) / VM_PITCH_X;
    const int ox = (tile % gxn) * VM_PITCH_X + offx, oy = (tile / gxn) * VM_PITCH_Y + offy;
    if (ox >= L.w || oy >= L.h)
        return;
    const MaskGeom g = mask_geom(L, ox, oy);
    // the mask words and the two small tables travel together: one round trip
    uint32_t mymask = 0;
    if (tid < g.nbx * g.nby) {
        int mx = tid % g.nbx, my = tid / g.nbx;
        mymask = L.impmask[(g.by0 + my + 1) * L.imp_rs + (g.bx0 + mx + 1)];
        S.mask[my][mx] = mymask;
    }
    for (int k = tid; k < 225; k += T)
        S.imp[k] = tables[VM_TAB_IMP + k];
    if (tid >= 256 && tid < 256 + 25) // ctx_load only needs the centre weight of each border class
        S.tps[(tid - 256) * 25 + 12] = __uint_as_float(tables[VM_TAB_TPS + (tid - 256) * 25 + 12]);
    if (T < 512 && tid < 25)
        S.tps[tid * 25 + 12] = __uint_as_float(tables[VM_TAB_TPS + tid * 25 + 12]);
    if (tid == 0)
        S.n_eval = 0;
    if (!__syncthreads_or(mymask != 0))
        return;
    VM_TS(1);
    VM_TS(2);

    // every mask hit of the phase, in slot order: the list is the same in all `parts`
    // workgroups of the tile, entry i belongs to workgroup i % parts
    bool hit = false;
    if (tid < 256) {
        const int px = ox + (tid & 31) * 2 + pj, py = oy + (tid >> 5) * 2 + pi;
        hit = px < L.w && py < L.h && mask_hit(S.mask, S.imp, g, px, py);
    }
    const int n_hit = compact256(hit, tid, S.list, S.wave_cnt);
    const int n_mine = (n_hit - part + parts - 1) / parts;
    if (n_mine <= 0)
        return;
    const int Lf = 32; // SPLIT is the latency-bound regime: always 32 lanes per candidate
    const int slots = T / Lf;
    const int sub = tid & (Lf - 1), grp = tid / Lf;
    for (int base = 0; base < n_mine; base += slots) {
        const int m = base + grp;
        const int slot = S.list[part + parts * min(m, n_mine - 1)];
        const int px = ox + (slot & 31) * 2 + pj, py = oy + (slot >> 5) * 2 + pi;
        const bool wave_interior = __all(m >= n_mine || is_interior(L, px, py));
        if (m < n_mine) {
            uint32_t state = 2;
            uint32_t n_eval = 0;
            float2 step = make_float2(0, 0), luma = make_float2(0, 0);
            PixelCtx c;
            c.idx = py * L.rs + px;
            if (!pixel_locked(L, P.bcond, px, py)) {
                VM_TS(3);
                ctx_load(c, L, S.tps, px, py);
                c.tps_b = L.tps_b[c.idx];
                GlbSrc src{&L, (py - 2) * L.rs + (px - 2)};
#if VM_EXACT
                (void)wave_interior;
                const bool ok = decide_x32(L, P, src, c, sub, step, n_eval VM_TS_PASS);
                if (ok) { // the lumas commit_pixel_motion samples (morph.cu:997-1003)
                    const float nvx = c.v.x + step.x, nvy = c.v.y + step.y;
                    luma.x = tap(L.img0, L.w, L.h, L.rs, px - nvx + 0.5f, py - nvy + 0.5f);
                    luma.y = tap(L.img1, L.w, L.h, L.rs, px + nvx + 0.5f, py + nvy + 0.5f);
                }
#else
                Nb1 nb;
                bool ok;
                if (wave_interior) {
                    nb1_load<true>(nb, L, src, c, sub);
                    ok = decide32<true>(L, P, nb, RingGlobal{L.v}, c, sub, step, luma, n_eval VM_TS_PASS);
                } else {
                    nb1_load<false>(nb, L, src, c, sub);
                    ok = decide32<false>(L, P, nb, RingGlobal{L.v}, c, sub, step, luma, n_eval VM_TS_PASS);
                }
#endif
                if (ok)
                    state = 1;
            }
            if (sub == 0) {
                if (n_eval)
                    atomicAdd(&S.n_eval, n_eval);
                float4 ra = make_float4(0, 0, 0, 0), rb = make_float4(0, 0, 0, 0);
                if (state == 1) {
                    // commit_pixel_motion (morph.cu:990-1026), the pixel's own part
                    const float2 ol = c.old_luma;
                    ra = make_float4(luma.x - ol.x, luma.y - ol.y, luma.x * luma.x - ol.x * ol.x,
                                     luma.y * luma.y - ol.y * ol.y);
                    rb.x = luma.x * luma.y - ol.x * ol.y;
                    rb.y = step.x;
                    rb.z = step.y;
                    L.luma[c.idx] = luma;
                    L.ui_b[c.idx] = make_float2(c.ui_b.x + 2 * step.x * c.ui_axy, c.ui_b.y + 2 * step.y * c.ui_axy);
                    L.v[c.idx] = make_float2(c.v.x + step.x, c.v.y + step.y);
                    L.rec_a[c.idx] = ra;
                }
                if (state == 1)
                    L.rec_b[c.idx] = rb;
                L.rec_tag[c.idx] = (epoch << 2) | state;
            }
        }
    }
    __syncthreads();
    if (tid == 0 && S.n_eval)
        atomicAdd(&stats[iter_idx * VM_STAT_WORDS + 4], S.n_eval);
#ifdef VM_PROF
    VM_TS(8);
    if ((tid & 63) == 0 && blockIdx.x < 512 && (tid >> 6) < 1) {
        ts[9] = (unsigned long long)Lf;
        ts[10] = (unsigned long long)n_mine;
        for (int k = 0; k < 16; ++k) vm_prof_buf[blockIdx.x * 16 + k] = ts[k];
    }
#endif
}

__global__ __launch_bounds__(1024) void SUF(k_commit)(const VmLevelView *__restrict__ views, int cap, VmKParams P,
                                                      const uint32_t *__restrict__ tables, int offx, int offy,
                                                      int pi, int pj, uint32_t epoch,
                                                      uint32_t *__restrict__ flags, uint32_t *__restrict__ stats,
                                                      int iter_idx, int fixed_work)
{
    __shared__ SplitLds S;
    const int tid = threadIdx.x, T = blockDim.x;
    const VmLevelView L = views[blockIdx.z];
    flags += (size_t)blockIdx.z * cap;
    stats += (size_t)blockIdx.z * cap * VM_STAT_WORDS;
    if (!fixed_work && iter_idx > 0 && flags[iter_idx - 1] == 0)
        return;
    const int ox = blockIdx.x * VM_PITCH_X + offx, oy = blockIdx.y * VM_PITCH_Y + offy;
    if (ox >= L.w || oy >= L.h)
        return;
    // One round trip: the records of this tile's phase pixels, the mask words, the stencil
    // table and (speculatively) the sums of the two cells each thread may have to update.
    const MaskGeom g = mask_geom(L, ox, oy);
    int state = 0;
    uint32_t bit = 0;
    int mcx = 0, mcy = 0;
    if (tid < 256) {
        const int px = ox + (tid & 31) * 2 + pj, py = oy + (tid >> 5) * 2 + pi;
        if (px < L.w && py < L.h) {
            const uint32_t r = L.rec_tag[py * L.rs + px];
            const float4 rb = L.rec_b[py * L.rs + px];
            const float4 ra = L.rec_a[py * L.rs + px];
            if ((r >> 2) == epoch) {
                state = (int)(r & 3u);
                if (state == 1) {
                    S.d_mean[tid] = make_float2(ra.x, ra.y);
                    S.d_var[tid] = make_float2(ra.z, ra.w);
                    S.d_cross[tid] = rb.x;
                    S.d_step[tid] = make_float2(rb.y, rb.z);
                }
            }
            mcx = px / 5 - g.bx0;
            mcy = py / 5 - g.by0;
            bit = 1u << ((px % 5) + (py % 5) * 5);
        }
        S.d_ok[tid] = state;
    } else if (tid < 256 + g.nbx * g.nby) {
        const int k = tid - 256, mx = k % g.nbx, my = k / g.nbx;
        S.mask[my][mx] = L.impmask[(g.by0 + my + 1) * L.imp_rs + (g.bx0 + mx + 1)];
    }
    for (int k = tid; k < 625; k += T)
        S.tps[k] = __uint_as_float(tables[VM_TAB_TPS + k]);
    float2 cm[2], cq[2], ctb[2];
    float ccr[2];
    int cgi[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int cell = tid + e * 1024;
        const int ry = cell / VM_HALO_W - 2, rx = cell % VM_HALO_W - 2;
        const int qx = ox + rx, qy = oy + ry;
        cgi[e] = -1;
        if (cell < VM_NCELL && qx >= 0 && qx < L.w && qy >= 0 && qy < L.h) {
            const int gi = qy * L.rs + qx;
            cgi[e] = gi;
            cm[e] = L.mean[gi];
            cq[e] = L.var[gi];
            ctb[e] = L.tps_b[gi];
            ccr[e] = L.cross[gi];
        }
    }
    const int n_rec = __syncthreads_count(state != 0);
    if (n_rec == 0)
        return;
    // the improving mask: a committed pixel sets its bit, a hit that did not move clears it
    if (state == 1)
        atomicOr(&S.mask[mcy][mcx], bit);
    else if (state == 2)
        atomicAnd(&S.mask[mcy][mcx], ~bit);
    const int ncommit = __syncthreads_count(state == 1);
    if (ncommit) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (cgi[e] < 0)
                continue;
            const int cell = tid + e * 1024;
            const int ry = cell / VM_HALO_W - 2, rx = cell % VM_HALO_W - 2;
            float2 m = cm[e], q = cq[e], tb = ctb[e];
            float cr = ccr[e];
            if (!gather_cell(S, L, ox, oy, rx, ry, pi, pj, m, q, cr, tb, P.commit_order))
                continue;
            const int gi = cgi[e];
            L.mean[gi] = m;
            L.var[gi] = q;
            L.cross[gi] = cr;
            L.tps_b[gi] = tb;
            const int qx = ox + rx, qy = oy + ry;
            const float counter = (float)(window_count(qy, L.h) * window_count(qx, L.w));
            L.value[gi] = ssim_value(m.x, m.y, q.x, q.y, cr, counter, P.ssim_clamp);
        }
    }
    if (tid < g.nbx * g.nby) {
        int mx = tid % g.nbx, my = tid / g.nbx;
        if (mx >= 1 && mx <= g.nbx - 2 && my >= 1 && my <= g.nby - 2)
            L.impmask[(g.by0 + my + 1) * L.imp_rs + (g.bx0 + mx + 1)] = S.mask[my][mx];
    }
    if (tid == 0) {
        if (ncommit)
            flags[iter_idx] = 1u; // every writer stores the same 1: no atomic needed (see k_step)
        atomicAdd(&stats[iter_idx * VM_STAT_WORDS + 3], 1u); // tile-phases with records
        atomicAdd(&stats[iter_idx * VM_STAT_WORDS + 1], (uint32_t)n_rec);
        atomicAdd(&stats[iter_idx * VM_STAT_WORDS + 2], (uint32_t)ncommit);
    }
}


// ===========================================================================
// STEP schedule: ONE launch per phase.  The commit of phase s-1 and the line
// searches of phase s share a launch: the window sums and the mask live in two copies;
// a step reads copy `src` (state before the records of phase s-1 were folded in) and
//  - its FOLD workgroups (one per 64x16 block of the level, geometry-free) write
//    sums + records(s-1) -> copy `dst`, whole level;
//  - its DECIDE workgroups (tiles x parts, as k_decide) fold the same records privately,
//    per lane, into the one window cell the lane owns, run the 32-lane line search and
//    write the pixel's own state + records(s).
// Folding is defined by image coordinates only -- a record of epoch s-1 reaches every
// cell within +-2 of its pixel, row-major order -- which is what gather_cell computes
// tile-relative; so the STEP schedule is bit-identical to the SPLIT schedule (same
// arithmetic mode).  A batch of steps ends with a fold-only launch in place on copy 0.
// staged window of the record tags: the tile +-4 (decide: cells +-2 of a tile pixel, their
// records +-2 more) or a 64x16 block +-2 (fold); kept as one commit bit per pixel
#define VM_STEP_BW (VM_TILE_W + 8)
#define VM_STEP_BH (VM_TILE_H + 8)
struct StepLds {
    uint32_t bits[VM_STEP_BH][4]; // bit x of row y: pixel (x0 + x, y0 + y) committed in the last phase
    int list[256];
    int wave_cnt[4];
    float tps[625];
    uint32_t imp[225];
    uint32_t mask[6][16];
    uint32_t n_commit;
    uint32_t n_eval;
};

// the 25 commit bits of the 5x5 window around cell (qx, qy); (bx0, by0) = origin of the staged window
__device__ __forceinline__ uint32_t cell_hits(const uint32_t (*bits)[4], int bx0, int by0, int qx, int qy)
{
    const int sx = qx - 2 - bx0, sy = qy - 2 - by0;
    const int wi = sx >> 5, sh = sx & 31;
    uint32_t hits = 0;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const uint32_t lo = bits[sy + dy][wi], hi = bits[sy + dy][wi + 1];
        hits |= (__funnelshift_r(lo, hi, sh) & 31u) << (5 * dy);
    }
    return hits;
}

// sums of cell (qx, qy) + the committed records flagged in `hits`, row-major (ssim_update,
// morph.cu:951-988, and the tps.b scatter, :1006-1015, as a gather).  The (at most 9) records
// are fetched in two batches of 5 and 4 -- all loads of a batch are issued before the first is
// used, and a batch is skipped when no lane of the wave has a record left; one batch of 9 would
// cost 72 VGPRs and with them a workgroup per CU.

// COH: the records were written by other workgroups of THIS launch (PASS schedule): L1-bypassing loads.
template <bool COH>
__device__ __forceinline__ float4 rec_load(const float4 *p)
{
    if (!COH)
        return *p;
    vm_g_cu64 *u = (vm_g_cu64 *)p;
    const unsigned long long a = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)b),
                       __uint_as_float((uint32_t)(b >> 32)));
}

// bit dy * 5 + dx of a 5 x 5 window field -> bit dx * 5 + dy
__device__ __forceinline__ uint32_t transpose5(uint32_t h)
{
    uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 25; ++k)
        t |= ((h >> k) & 1u) << ((k % 5) * 5 + k / 5);
    return t;
}

template <bool COH = false>
__device__ __forceinline__ bool fold_cell(const VmLevelView &L, const float4 *__restrict__ r_a,
                                          const float4 *__restrict__ r_b, const float *s_tps, uint32_t hits, int qx,
                                          int qy, float2 &m, float2 &q, float &cr, float2 &tb, int order)
{
#if !VM_EXACT
    (void)order;
#endif
    const bool touched = hits != 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (!__any(hits != 0))
            break;
        float4 ra[5], rb[5];
        int bidx[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (half == 1 && k == 4) { // 5 + 4: at most 9 pixels of one phase lie in a 5x5 window
                bidx[k] = -1;
                continue;
            }
#if VM_EXACT
            // order (vm_set_commit_order): bit 0 = the last window position first, bit 1 = positions in
            // column-major order (picked from the transposed 5 x 5 bit field)
            int b = -1;
            if (hits) {
                if (!(order & 2)) {
                    b = (order & 1) ? 31 - __clz(hits) : __ffs(hits) - 1;
                } else {
                    const uint32_t ht = transpose5(hits);
                    const int bt = (order & 1) ? 31 - __clz(ht) : __ffs(ht) - 1;
                    b = (bt % 5) * 5 + bt / 5;
                }
                hits &= ~(1u << b);
            }
#else
            const int b = hits ? __ffs(hits) - 1 : -1;
            hits &= hits - 1;
#endif
            bidx[k] = b;
            const int bb = max(b, 0);
            const int dy = (bb * 13) >> 6, dx = bb - dy * 5;
            const int rsafe = b >= 0 ? (qy + dy - 2) * L.rs + (qx + dx - 2) : qy * L.rs + qx;
            ra[k] = rec_load<COH>(r_a + rsafe);
            rb[k] = rec_load<COH>(r_b + rsafe);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int b = bidx[k];
            if (b < 0)
                continue;
            const int dy = (b * 13) >> 6, dx = b - dy * 5;
            const int x = qx + dx - 2, y = qy + dy - 2;
            m.x += ra[k].x;
            m.y += ra[k].y;
            q.x += ra[k].z;
            q.y += ra[k].w;
            cr += rb[k].x;
            const float kk = s_tps[(border_class(y, L.h) * 5 + border_class(x, L.w)) * 25 + (4 - dy) * 5 + (4 - dx)];
            tb.x += rb[k].y * kk;
            tb.y += rb[k].z * kk;
        }
    }
    return touched;
}

template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(T / 128))) void SUF(k_step)(const VmLevelView *__restrict__ views, int cap, VmKParams P,
                                                   const uint32_t *__restrict__ tables, int offx, int offy, int pi,
                                                   int pj, int parts, uint32_t epoch, uint32_t pe, int srcbuf,
                                                   int n_fold, uint32_t *__restrict__ flags,
                                                   uint32_t *__restrict__ stats, int iter_idx, int fixed_work,
                                                   uint32_t *__restrict__ slots_cur, const uint32_t *__restrict__ slots_prev,
                                                   int prev_iter_idx, int nslot)
{
    __shared__ StepLds S;
    const int tid = threadIdx.x;
#ifdef VM_PROF
    unsigned long long ts[16];
    for (int k = 0; k < 16; ++k) ts[k] = 0;
    VM_TS(0);
#endif
    const VmLevelView L = views[blockIdx.z];
    flags += (size_t)blockIdx.z * cap;
    stats += (size_t)blockIdx.z * cap * VM_STAT_WORDS;
    // copy 0 = the canonical arrays; srcbuf 2 = fold in place on copy 0 (end of a batch)
    const bool s1 = srcbuf == 1, d1 = srcbuf == 0;
    const float2 *s_mean = s1 ? L.mean2 : L.mean, *s_var = s1 ? L.var2 : L.var, *s_tpsb = s1 ? L.tps_b2 : L.tps_b;
    const float *s_cross = s1 ? L.cross2 : L.cross, *s_value = s1 ? L.value2 : L.value;
    const uint32_t *s_imp = s1 ? L.impmask2 : L.impmask;
    float2 *d_mean = d1 ? L.mean2 : L.mean, *d_var = d1 ? L.var2 : L.var, *d_tpsb = d1 ? L.tps_b2 : L.tps_b;
    float *d_cross = d1 ? L.cross2 : L.cross, *d_value = d1 ? L.value2 : L.value;
    uint32_t *d_imp = d1 ? L.impmask2 : L.impmask;
    const bool in_place = srcbuf == 2;
    // records: read the set the previous phase wrote, write the other one
    const bool r1 = pe & 1u, w1 = epoch & 1u;
    const uint32_t *r_tag = r1 ? L.rec_tag2 : L.rec_tag;
    const float4 *r_a = r1 ? L.rec_a2 : L.rec_a, *r_b = r1 ? L.rec_b2 : L.rec_b;
    uint32_t *w_tag = w1 ? L.rec_tag2 : L.rec_tag;
    float4 *w_a = w1 ? L.rec_a2 : L.rec_a, *w_b = w1 ? L.rec_b2 : L.rec_b;
    const uint32_t want = (pe << 2) | 1u;

    for (int k = tid; k < 625; k += T)
        S.tps[k] = __uint_as_float(tables[VM_TAB_TPS + k]);
    if (tid < VM_STEP_BH * 4)
        S.bits[tid >> 2][tid & 3] = 0;

    if ((int)blockIdx.x < n_fold) {
        // ------------------------------------------------------------- FOLD
        const int nfx = (L.w + 63) / 64;
        const int x0 = ((int)blockIdx.x % nfx) * 64, y0 = ((int)blockIdx.x / nfx) * 16;
        const int bx0 = x0 - 2, by0 = y0 - 2; // staged window: 68 x 20
        constexpr int NTF = (68 * 20 + T - 1) / T, NCF = 1024 / T;
        uint32_t tg[NTF];
#pragma unroll
        for (int e = 0; e < NTF; ++e) {
            const int k = tid + e * T;
            const int x = bx0 + k % 68, y = by0 + k / 68;
            tg[e] = (k < 68 * 20 && x >= 0 && x < L.w && y >= 0 && y < L.h) ? r_tag[y * L.rs + x] : 0u;
        }
        float2 cm[NCF], cq[NCF], ctb[NCF];
        float ccr[NCF], cval[NCF];
        int cgi[NCF];
#pragma unroll
        for (int e = 0; e < NCF; ++e) {
            const int cell = tid + e * T;
            const int qx = x0 + (cell & 63), qy = y0 + (cell >> 6);
            cgi[e] = -1;
            if (qx < L.w && qy < L.h) {
                const int gi = qy * L.rs + qx;
                cgi[e] = gi;
                cm[e] = s_mean[gi];
                cq[e] = s_var[gi];
                ctb[e] = s_tpsb[gi];
                ccr[e] = s_cross[gi];
                cval[e] = s_value[gi];
            }
        }
        // mask words, by index range over the padded word array
        const int nw = L.imp_rs * L.imp_rows, per = (nw + n_fold - 1) / n_fold;
        const int wi = (int)blockIdx.x * per + tid;
        if (tid < per && wi < nw) {
            const int my = wi / L.imp_rs, mx = wi - my * L.imp_rs;
            uint32_t word = s_imp[wi];
            if (mx >= 1 && my >= 1 && (mx - 1) * 5 < L.w && (my - 1) * 5 < L.h) {
                // a committed pixel sets its bit, a hit that did not move clears it
                uint32_t tw[25];
#pragma unroll
                for (int k = 0; k < 25; ++k) {
                    const int dy = (k * 13) >> 6, dx = k - dy * 5;
                    const int x = (mx - 1) * 5 + dx, y = (my - 1) * 5 + dy;
                    tw[k] = (x < L.w && y < L.h) ? r_tag[y * L.rs + x] : 0u;
                }
#pragma unroll
                for (int k = 0; k < 25; ++k)
                    if ((tw[k] >> 2) == pe) {
                        if ((tw[k] & 3u) == 1u)
                            word |= 1u << k;
                        else if ((tw[k] & 3u) == 2u)
                            word &= ~(1u << k);
                    }
            }
            d_imp[wi] = word;
        }
        __syncthreads(); // bits zeroed
#pragma unroll
        for (int e = 0; e < NTF; ++e) {
            const int k = tid + e * T;
            if (tg[e] == want)
                atomicOr(&S.bits[k / 68][(k % 68) >> 5], 1u << ((k % 68) & 31));
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NCF; ++e) {
            const int cell = tid + e * T;
            const int qx = x0 + (cell & 63), qy = y0 + (cell >> 6);
            const bool in = cgi[e] >= 0;
            float2 m = cm[e], q = cq[e], tb = ctb[e];
            float cr = ccr[e], val = cval[e];
            const uint32_t hits = in ? cell_hits(S.bits, bx0, by0, qx, qy) : 0u;
            const bool touched = fold_cell(L, r_a, r_b, S.tps, hits, in ? qx : 0, in ? qy : 0, m, q, cr, tb, P.commit_order);
            if (!in)
                continue;
            if (touched) {
                const float counter = (float)(window_count(qy, L.h) * window_count(qx, L.w));
                val = ssim_value(m.x, m.y, q.x, q.y, cr, counter, P.ssim_clamp);
            } else if (in_place) {
                continue;
            }
            const int gi = cgi[e];
            d_mean[gi] = m;
            d_var[gi] = q;
            d_cross[gi] = cr;
            d_tpsb[gi] = tb;
            d_value[gi] = val;
        }
        // Activity counters: a decide workgroup leaves its counts in a slot (a plain store); the
        // first fold workgroup of the NEXT launch adds them up.  Five agent-scope atomics per
        // workgroup on one line were 0.8 us of every launch (they drain at memory, and the launch
        // cannot end before they have).
        if (blockIdx.x == 0 && prev_iter_idx >= 0) {
            __syncthreads();
            if (tid < 4)
                S.wave_cnt[tid] = 0;
            __syncthreads();
            uint32_t acc[4] = {0, 0, 0, 0};
            const uint4 *sl = (const uint4 *)slots_prev + (size_t)blockIdx.z * nslot;
            for (int k = tid; k < nslot; k += T) {
                const uint4 q = sl[k];
                acc[0] += q.x;
                acc[1] += q.y;
                acc[2] += q.z;
                acc[3] += q.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (acc[k])
                    atomicAdd((uint32_t *)&S.wave_cnt[k], acc[k]);
            __syncthreads();
            if (tid == 0) {
                uint32_t *st = stats + (size_t)prev_iter_idx * VM_STAT_WORDS;
                st[1] += (uint32_t)S.wave_cnt[0]; // line searches
                st[2] += (uint32_t)S.wave_cnt[1]; // commits
                st[4] += (uint32_t)S.wave_cnt[2]; // energy evaluations
                st[3] += (uint32_t)S.wave_cnt[3]; // tile-phases with records
            }
        }
        return;
    }

    // ------------------------------------------------------------- DECIDE
    // converged in the previous iteration: nothing can be a candidate (sticky); the folds go on
    const int bid = (int)blockIdx.x - n_fold;
    uint4 *my_slot = (uint4 *)slots_cur + (size_t)blockIdx.z * nslot + bid; // read by the next launch: always written
    if (!fixed_work && iter_idx > 0 && flags[iter_idx - 1] == 0) {
        if (tid == 0)
            *my_slot = make_uint4(0, 0, 0, 0);
        return;
    }
    const int part = bid % parts, tile = bid / parts;
    const int gxn = (L.w + VM_PITCH_X - 1) / VM_PITCH_X;
    const int ox = (tile % gxn) * VM_PITCH_X + offx, oy = (tile / gxn) * VM_PITCH_Y + offy;
    if (ox >= L.w || oy >= L.h) {
        if (tid == 0)
            *my_slot = make_uint4(0, 0, 0, 0);
        return;
    }
    const MaskGeom g = mask_geom(L, ox, oy);
    const int bx0 = ox - 4, by0 = oy - 4; // staged window: 72 x 24
    constexpr int NTD = (VM_STEP_BW * VM_STEP_BH + T - 1) / T;
    uint32_t tg[NTD];
#pragma unroll
    for (int e = 0; e < NTD; ++e) {
        const int k = tid + e * T;
        const int x = bx0 + k % VM_STEP_BW, y = by0 + k / VM_STEP_BW;
        tg[e] = (k < VM_STEP_BW * VM_STEP_BH && x >= 0 && x < L.w && y >= 0 && y < L.h) ? r_tag[y * L.rs + x] : 0u;
    }
    if (tid < g.nbx * g.nby)
        S.mask[tid / g.nbx][tid % g.nbx] = s_imp[(g.by0 + tid / g.nbx + 1) * L.imp_rs + (g.bx0 + tid % g.nbx + 1)];
    for (int k = tid; k < 225; k += T)
        S.imp[k] = tables[VM_TAB_IMP + k];
    if (tid == 0) {
        S.n_commit = 0;
        S.n_eval = 0;
    }
    __syncthreads();
    VM_TS(1);
    // the last phase's records: commit bits for the cell folds, and the mask words -- a committed
    // pixel sets its bit, a hit that did not move clears it.  (Mask bits further than 2 pixels
    // from the tile stay unfolded here: mask_hit never looks at them.)
#pragma unroll
    for (int e = 0; e < NTD; ++e) {
        const int k = tid + e * T;
        const uint32_t t = tg[e];
        if ((t >> 2) == pe && (t & 3u) != 0u) {
            const int sx = k % VM_STEP_BW, sy = k / VM_STEP_BW;
            const int x = bx0 + sx, y = by0 + sy;
            const uint32_t bit = 1u << ((x % 5) + (y % 5) * 5);
            uint32_t *word = &S.mask[y / 5 - g.by0][x / 5 - g.bx0];
            if ((t & 3u) == 1u) {
                atomicOr(&S.bits[sy][sx >> 5], 1u << (sx & 31));
                atomicOr(word, bit);
            } else {
                atomicAnd(word, ~bit);
            }
        }
    }
    __syncthreads();
    VM_TS(2);

    bool hit = false;
    if (tid < 256) {
        const int px = ox + (tid & 31) * 2 + pj, py = oy + (tid >> 5) * 2 + pi;
        hit = px < L.w && py < L.h && mask_hit(S.mask, S.imp, g, px, py);
    }
    const int n_hit = compact256(hit, tid, S.list, S.wave_cnt);
    const int n_mine = (n_hit - part + parts - 1) / parts;
    if (n_mine <= 0) {
        if (tid == 0)
            *my_slot = make_uint4(0, 0, 0, 0);
        return;
    }
    // few candidates (the launch-bound small levels): a whole wave per pixel, decide64
    const bool wide = n_mine * 64 <= T;
    const int slots = wide ? T / 64 : T / 32;
    const int sub = tid & 31, grp = wide ? tid >> 6 : tid >> 5;
    const bool writer = wide ? (tid & 63) == 0 : sub == 0;
    uint32_t my_commits = 0;
    for (int base = 0; base < n_mine; base += slots) {
        const int mi = base + grp;
        const int slot = S.list[part + parts * min(mi, n_mine - 1)];
        const int px = ox + (slot & 31) * 2 + pj, py = oy + (slot >> 5) * 2 + pi;
        const bool wave_interior = __all(mi >= n_mine || is_interior(L, px, py));
        const bool live = mi < n_mine && !pixel_locked(L, P.bcond, px, py);
        uint32_t state = 2;
        uint32_t n_eval = 0;
        float2 step = make_float2(0, 0), luma = make_float2(0, 0);
        PixelCtx c;
        c.px = px;
        c.py = py;
        c.idx = py * L.rs + px;
        VM_TS(3);
        // every lane takes part in the fold (its slot loops are wave-uniform); idle lanes fold
        // the pixel's own cell with no records
        ctx_load(c, L, S.tps, px, py);
        // the lane's window cell: sums of copy `src` + the records of the last phase
        const int i = (sub * 13) >> 6, jj = sub - i * 5; // lane sub < 25 owns neighbour (sub % 5 - 2, sub / 5 - 2)
        const int qx = px + jj - 2, qy = py + i - 2;
        const bool okc = sub < 25 && qx >= 0 && qx < L.w && qy >= 0 && qy < L.h;
        const int cx = okc ? qx : px, cy = okc ? qy : py, gi = cy * L.rs + cx;
        float2 m = s_mean[gi], q = s_var[gi], tb = s_tpsb[gi];
        float cr = s_cross[gi], val = s_value[gi];
        const uint32_t hits = live && okc ? cell_hits(S.bits, bx0, by0, cx, cy) : 0u;
        if (fold_cell(L, r_a, r_b, S.tps, hits, cx, cy, m, q, cr, tb, P.commit_order)) {
            const float counter = (float)(window_count(cy, L.h) * window_count(cx, L.w));
            val = ssim_value(m.x, m.y, q.x, q.y, cr, counter, P.ssim_clamp);
        }
        // tps.b of the pixel itself is the folded value of its own cell (lane 12)
        c.tps_b.x = __shfl(tb.x, 12, 32);
        c.tps_b.y = __shfl(tb.y, 12, 32);
        if (live) {
            bool ok;
#if VM_EXACT
            (void)wave_interior;
            NbX nb;
            nb.ok = okc;
            nb.m = m;
            nb.q = q;
            nb.cr = cr;
            nb.val = val;
            nb.counter = okc ? (float)(window_count(qy, L.h) * window_count(qx, L.w)) : 25.0f;
            if (wide)
                ok = decide_with64(
                    L, P, c, [&](float dx, float dy) { return energy_x32(L, P, nb, c, dx, dy); }, RingGlobal{L.v},
                    (tid & 32) != 0, step, n_eval);
            else
                ok = decide_with(
                    L, P, c, [&](float dx, float dy) { return energy_x32(L, P, nb, c, dx, dy); }, RingGlobal{L.v}, step,
                    n_eval VM_TS_PASS);
            if (ok) { // the lumas commit_pixel_motion samples (morph.cu:997-1003)
                const float nvx = c.v.x + step.x, nvy = c.v.y + step.y;
                luma.x = tap(L.img0, L.w, L.h, L.rs, px - nvx + 0.5f, py - nvy + 0.5f);
                luma.y = tap(L.img1, L.w, L.h, L.rs, px + nvx + 0.5f, py + nvy + 0.5f);
            }
#else
            Nb1 nb;
            if (wide) {
                const bool hi = (tid & 32) != 0;
                if (wave_interior) {
                    nb1_make<true>(nb, L, okc, qx, qy, m, q, cr, val);
                    ok = decide64<true>(L, P, nb, RingGlobal{L.v}, c, sub, hi, step, luma, n_eval);
                } else {
                    nb1_make<false>(nb, L, okc, qx, qy, m, q, cr, val);
                    ok = decide64<false>(L, P, nb, RingGlobal{L.v}, c, sub, hi, step, luma, n_eval);
                }
            } else if (wave_interior) {
                nb1_make<true>(nb, L, okc, qx, qy, m, q, cr, val);
                ok = decide32<true>(L, P, nb, RingGlobal{L.v}, c, sub, step, luma, n_eval VM_TS_PASS);
            } else {
                nb1_make<false>(nb, L, okc, qx, qy, m, q, cr, val);
                ok = decide32<false>(L, P, nb, RingGlobal{L.v}, c, sub, step, luma, n_eval VM_TS_PASS);
            }
#endif
            if (ok)
                state = 1;
        }
        if (mi < n_mine && writer) {
            if (n_eval)
                atomicAdd(&S.n_eval, n_eval);
            if (state == 1) {
                const float2 ol = c.old_luma;
                w_a[c.idx] = make_float4(luma.x - ol.x, luma.y - ol.y, luma.x * luma.x - ol.x * ol.x,
                                         luma.y * luma.y - ol.y * ol.y);
                w_b[c.idx] = make_float4(luma.x * luma.y - ol.x * ol.y, step.x, step.y, 0.0f);
                L.luma[c.idx] = luma;
                L.ui_b[c.idx] = make_float2(c.ui_b.x + 2 * step.x * c.ui_axy, c.ui_b.y + 2 * step.y * c.ui_axy);
                L.v[c.idx] = make_float2(c.v.x + step.x, c.v.y + step.y);
                ++my_commits;
            }
            w_tag[c.idx] = (epoch << 2) | state;
        }
    }
    if (my_commits)
        atomicAdd(&S.n_commit, my_commits);
#ifdef VM_PROF
    VM_TS(8);
    if ((tid & 63) == 0 && bid < 512 && (tid >> 6) < 1) {
        ts[9] = 32;
        ts[10] = (unsigned long long)n_mine;
        for (int k = 0; k < 16; ++k) vm_prof_buf[bid * 16 + k] = ts[k];
    }
#endif
    __syncthreads();
    if (tid == 0) {
        const uint32_t nc = S.n_commit;
        // the next iteration's first launch reads this flag, so it cannot wait for the slot fold;
        // every writer stores the same 1 into a word that was cleared before the batch: a plain
        // store does (L2 write-back is byte-masked), no atomic has to drain at memory
        if (nc)
            flags[iter_idx] = 1u;
        *my_slot = make_uint4((uint32_t)n_mine, nc, S.n_eval, part == 0 ? 1u : 0u);
    }
}


// ===========================================================================
// PASS schedule: ONE launch per pass for the small levels a single frame pair is bound by
// (120x68 of a 1080p pyramid: 8 tiles per pass, every pixel active for all 500 iterations,
// 8000 dependent phases).  STEP pays a kernel boundary per phase: launch gap, three cold
// dependent round trips at entry, the level re-staged into eight cold L2s.  Here the four
// Jacobi phases of a tile stay inside one launch behind a TILE-LOCAL barrier:
//  - a tile is worked by a GROUP of 32 workgroups of 8 waves, one wave per phase pixel
//    (slot = part * 8 + wave): no candidate list, no workgroup-wide staging.  Per phase a wave
//    fetches everything it needs in ONE round trip (mask words, the last phase's tags and
//    records around its pixel, its pixel's state, the window cell each lane owns, the ring
//    neighbours' v), passes the records through a wave-private LDS area, folds them into its
//    window cells, tests its pixel's mask bits and runs the wave-wide line search (decide64 /
//    decide_with64);
//  - the fold of phase s-1's records into the window sums is a by-product of that: every cell
//    of the tile + halo is OWNED by one slot (the 2 x 2 cells at its pixel, edge slots the
//    halo beside them), whose wave stores the folded cell into the other copy of the sums:
//        phase 0 reads C | 1: C + rec(0) -> T | 2: T + rec(1) -> C | 3: C + rec(2) -> T |
//        closing step after the last barrier: T + rec(3) -> C
//    so a pass leaves the canonical state complete and the next pass (other tile geometry)
//    starts from it like any other schedule;
//  - workgroup ids b, b + 8, b + 16, ... belong to one group: under the observed round-robin
//    dispatch they share an XCD, so a tile's state stays in ONE L2.  That is checked, not
//    assumed: every workgroup reports its XCC id with its first arrival; a group found on one
//    XCD hands data over with plain stores (the XCD's L2 is the point of coherence of its
//    CUs) + L1-bypassing loads and meets at flags in that L2; a group found spread over XCDs
//    writes its phase-0 stores back (agent-scope release), meets once more and from then on
//    stores write-through (sc1) and meets at an agent-scope counter -- the acquire-free form
//    of MI355X_MICROARCH.md "Valid forms";
//  - every wait is bounded: a timeout raises an error word instead of hanging the device.
// Same arithmetic, same fold order, same records as STEP: bit-identical to it.
struct PassLds {
    float tps[625];
    uint32_t imp[225];
    float4 rec[8][49][2]; // per wave: the last phase's records around the wave's pixel (rec_a, rec_b), the
                          // 5 x 5 grid of positions inside a border of one: entry (j + 1) * 7 + (i + 1)
    uint32_t n_cand, n_commit, n_eval;
    uint32_t go;          // 1: go on, 0: a barrier timed out
    uint32_t wt;          // 1: the group spans XCDs: write-through stores, counter barrier
};

__device__ __forceinline__ uint32_t ldc(const uint32_t *p)
{
    return __hip_atomic_load((vm_g_cu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ldc(const float *p) { return __uint_as_float(ldc((const uint32_t *)p)); }
__device__ __forceinline__ float2 ldc(const float2 *p)
{
    const unsigned long long u = __hip_atomic_load((vm_g_cu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((uint32_t)u), __uint_as_float((uint32_t)(u >> 32)));
}
__device__ __forceinline__ float4 ldc(const float4 *p) { return rec_load<true>(p); }
// stores of handed-off data: write-through (sc1) when the group spans XCDs, plain otherwise
__device__ __forceinline__ void sth(uint32_t *p, uint32_t v, bool wt)
{
    if (wt)
        __hip_atomic_store((vm_g_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
__device__ __forceinline__ void sth(float *p, float v, bool wt) { sth((uint32_t *)p, __float_as_uint(v), wt); }
__device__ __forceinline__ void sth(float2 *p, float2 v, bool wt)
{
    if (wt) {
        const unsigned long long u = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
        __hip_atomic_store((vm_g_u64 *)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *p = v;
    }
}
__device__ __forceinline__ void sth(float4 *p, float4 v, bool wt)
{
    if (wt) {
        sth((float2 *)p, make_float2(v.x, v.y), true);
        sth((float2 *)p + 1, make_float2(v.z, v.w), true);
    } else {
        *p = v;
    }
}

// Addressing of k_pass: a uniform base (SGPR pair) + an unsigned 32-bit byte offset per lane -- the
// global_load / global_store form "vdst, voffset, s[base]": one v_lshl_add_u32 per address instead
// of a sign extension + a 64-bit shift-add, and half the SGPRs per array (an offset instead of a
// pointer; with 28 pointers of the level view live the kernel spilled 118 SGPRs).
template <class T> __device__ __forceinline__ T ldo(const char *base, uint32_t byte_off) { return ldc((const T *)(base + byte_off)); }
template <class T> __device__ __forceinline__ void sto(const char *base, uint32_t byte_off, T v, bool wt)
{
    sth((T *)const_cast<char *>(base + byte_off), v, wt);
}

#define VM_PASS_PARTS 32
#define VM_PASS_T 512
#define VM_PASS_TIMEOUT_TICKS 200000000ull // 2 s of the 100 MHz wall clock

// one wave adds the per-workgroup count slots of a finished PASS launch into the counters of
// iteration `it`, pair by pair (slot k belongs to group (k >> 8) * 8 + (k & 7))
__device__ __forceinline__ void pass_sum_slots(uint32_t *stats0, const uint32_t *slots, int it, int nslot, int ntiles,
                                               int ngroups, int cap, int lane, bool spread)
{
    const uint4 *sl = (const uint4 *)slots;
    const int npairs = ngroups / ntiles;
    for (int p = 0; p < npairs; ++p) {
        uint32_t acc[4] = {0, 0, 0, 0};
        for (int k = lane; k < nslot; k += 64) {
            const int grp = (k >> 8) * 8 + (spread ? (k & 255) >> 5 : k & 7);
            if (grp >= ngroups || grp / ntiles != p)
                continue;
            const uint4 q = sl[k];
            acc[0] += q.x;
            acc[1] += q.y;
            acc[2] += q.z;
            acc[3] += q.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            for (int o = 32; o > 0; o >>= 1)
                acc[k] += __shfl_xor(acc[k], o);
        if (lane == 0) {
            uint32_t *st = stats0 + ((size_t)p * cap + it) * VM_STAT_WORDS;
            st[1] += acc[0]; // line searches
            st[2] += acc[1]; // commits
            st[4] += acc[2]; // energy evaluations
            st[0] += acc[3]; // tile visits
        }
    }
}

__global__ __launch_bounds__(VM_PASS_T) __attribute__((amdgpu_waves_per_eu(2, 2))) void SUF(k_pass)(
    const VmLevelView *__restrict__ views, int cap, VmKParams P, const uint32_t *__restrict__ tables, int offx, int offy,
    uint32_t epoch0, int ngroups, int ntiles, uint32_t *__restrict__ sync, uint32_t *__restrict__ flags,
    uint32_t *__restrict__ stats, int iter_idx, int fixed_work, uint32_t *__restrict__ slots_cur,
    const uint32_t *__restrict__ slots_prev, int prev_iter_idx, int nslot_prev, uint32_t *__restrict__ err,
    uint32_t *__restrict__ dbg, int force_wt)
{
    __shared__ PassLds S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = tid & 31;
    const bool hi = (tid & 32) != 0;
    // workgroup -> (group, part): ids b, b + 8, ... of a 256-block chunk form one group
    const int b = (int)blockIdx.x, within = b & 255;
#ifdef VM_PROF
    if (tid == 0 && b < 256)
        vm_prof_buf[8192 + b * 2] = wall_clock64();
#endif
    // (force_wt & 4, a test switch: groups of 32 CONSECUTIVE ids instead -- every group then spans all eight
    // XCDs, which is what exercises the census, the write-back fence and the write-through hand-offs)
    const bool spread = (force_wt & 4) != 0;
    const int grp = (b >> 8) * 8 + (spread ? within >> 5 : within & 7), part = spread ? within & 31 : within >> 3;
    uint4 *my_slot = (uint4 *)slots_cur + b; // read by the next launch: always written
    if (grp >= ngroups) {
        if (tid == 0)
            *my_slot = make_uint4(0, 0, 0, 0);
        return;
    }
    const int pair = grp / ntiles, tile = grp - pair * ntiles;
    const VmLevelView L = views[pair];
    VM_PTSF(0);
    flags += (size_t)pair * cap;
    uint32_t *const stats0 = stats;
    unsigned long long *const bar = (unsigned long long *)(sync + (size_t)grp * VM_PASS_SYNC_WORDS);
    uint32_t *const flg = sync + (size_t)grp * VM_PASS_SYNC_WORDS + 32;
    const int gxn = (L.w + VM_PITCH_X - 1) / VM_PITCH_X;
    const int ox = (tile % gxn) * VM_PITCH_X + offx, oy = (tile / gxn) * VM_PITCH_Y + offy;
    const MaskGeom g = mask_geom(L, ox, oy);
    // (requested before the tables are copied: one round trip for both)
    uint32_t early_word = 0;
    if (ox < L.w && oy < L.h && tid < g.nbx * g.nby)
        early_word = L.impmask[(g.by0 + tid / g.nbx + 1) * L.imp_rs + (g.bx0 + tid % g.nbx + 1)];
    const uint32_t prev_flag = (!fixed_work && iter_idx > 0) ? flags[iter_idx - 1] : 1u;
    // a barrier of an earlier launch of this batch timed out: the host will discard the batch -- do not wait again
    const uint32_t err_before = ldc(err);
    for (int k = tid; k < 625; k += VM_PASS_T)
        S.tps[k] = __uint_as_float(tables[VM_TAB_TPS + k]);
    for (int k = tid; k < 225; k += VM_PASS_T)
        S.imp[k] = tables[VM_TAB_IMP + k];
    if (tid == 0) {
        S.n_cand = S.n_commit = S.n_eval = 0;
        S.go = 1;
        S.wt = 0;
    }
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; // HW_REG_XCC_ID
    if (dbg && tid == 0 && b < 2048) // diagnostic: which XCD the workgroup runs on
        dbg[b] = xcc;
    // Group-wide early outs.  Every workgroup of the group must take the same decision from data
    // no workgroup of this launch can have changed yet: the flag of the previous iteration, and
    // the mask bits of the positions within +-2 of the tile (pixels of this tile or of the gaps: no
    // other tile of the pass owns them).  No set bit there = no candidate in phase 0, hence no
    // commit, hence none in the later phases.  Positions past the right / bottom image edge
    // count: init_improving_mask sets whole words, nothing ever clears the bits of pixels that
    // do not exist, and the reference's test (get_improve_mask_idx) sees them -- a pixel within 2
    // of such an edge stays a candidate for ever (measured: 684 line searches per iteration of a
    // converged 120x68 level, in the oracle and in every schedule).
    VM_PTSF(1);
    bool live = ox < L.w && oy < L.h && prev_flag != 0 && err_before == 0;
    {
        uint32_t mine = 0;
        if (live && tid < g.nbx * g.nby) {
            const int mx = tid % g.nbx, my = tid / g.nbx;
            mine = early_word &
                   block_bits_in(g.bx0 + mx, g.by0 + my, ox - 2, min(ox + VM_TILE_W - 1, L.w - 1) + 2, oy - 2,
                                 min(oy + VM_TILE_H - 1, L.h - 1) + 2);
        }
        if (!__syncthreads_or(mine != 0))
            live = false;
    }
    VM_PTSF(2);
    // the activity counts of the previous launch, added up by one wave of workgroup 0 (a plain
    // read-modify-write by one thread: nothing else touches that iteration's counters now)
    if (b == 0 && wave == 7 && prev_iter_idx >= 0)
        pass_sum_slots(stats0, slots_prev, prev_iter_idx, nslot_prev, ntiles, ngroups, cap, lane, spread);
    if (!live) {
        if (tid == 0)
            *my_slot = make_uint4(0, 0, 0, 0);
        return;
    }
    // in-kernel clock probe (clock_probe above): workgroup 0 brackets its four phases
    if (b == 0 && tid == 0)
        clock_probe(stats0 + (size_t)iter_idx * VM_STAT_WORDS, false);

    // this wave's pixel slot of the tile: (tx, ty), pixel (ox + 2 tx + pj, oy + 2 ty + pi) in phase (pi, pj)
    const int slot = part * 8 + wave, tx = slot & 31, ty = slot >> 5;
    const int tx1 = min(ox + VM_TILE_W, L.w), ty1 = min(oy + VM_TILE_H, L.h); // tile pixels: [ox, tx1) x [oy, ty1)
    // mask words the tile owns (blocks holding one of its pixels); word `slot` is folded by this wave
    const int own_bx0 = ox / 5, own_by0 = oy / 5;
    const int own_nx = (tx1 - 1) / 5 - own_bx0 + 1, own_ny = (ty1 - 1) / 5 - own_by0 + 1;
    const bool has_word = slot < own_nx * own_ny;
    const int wbx = own_bx0 + (has_word ? slot % own_nx : 0), wby = own_by0 + (has_word ? slot / own_nx : 0);
    const int wword = (wby + 1) * L.imp_rs + (wbx + 1);

    // addressing (ldo / sto): the level's slab (vm_level_alloc: v first) and the schedule workspace
    // (level_ensure_ws: rec_tag first) as bases, the arrays as byte offsets from them (both
    // allocations are far below 4 GB: the host admits PASS only then)
    const char *const sb = (const char *)L.v, *const wb = (const char *)L.rec_tag;
#define VM_SO(p) ((uint32_t)((const char *)(p) - sb))
#define VM_WO(p) ((uint32_t)((const char *)(p) - wb))
    const uint32_t o_luma = VM_SO(L.luma), o_uib = VM_SO(L.ui_b), o_uiaxy = VM_SO(L.ui_axy), o_imp = VM_SO(L.impmask);
    const uint32_t o_mean = VM_SO(L.mean), o_var = VM_SO(L.var), o_tpsb = VM_SO(L.tps_b), o_cross = VM_SO(L.cross),
                   o_value = VM_SO(L.value);
    const uint32_t o_mean2 = VM_WO(L.mean2), o_var2 = VM_WO(L.var2), o_tpsb2 = VM_WO(L.tps_b2), o_cross2 = VM_WO(L.cross2),
                   o_value2 = VM_WO(L.value2), o_imp2 = VM_WO(L.impmask2);
    const uint32_t o_tag2 = VM_WO(L.rec_tag2), o_a = VM_WO(L.rec_a), o_b = VM_WO(L.rec_b), o_a2 = VM_WO(L.rec_a2),
                   o_b2 = VM_WO(L.rec_b2);
#undef VM_SO
#undef VM_WO

    uint32_t my_cand = 0, my_commit = 0, my_eval = 0; // of this wave, over the four phases
    bool wt = (force_wt & 1) != 0;                     // write-through stores, counter barrier
    bool pure_known = false;                           // after the first barrier: the group's XCD census is in
    uint32_t rounds = 0;                               // barrier rounds behind us
    bool timed_out = false;
    int prof_ph = 0;
    (void)prof_ph;

    // Tile barrier: every store of this workgroup has left (vmcnt(0) per wave, then the workgroup
    // barrier), then one arrival per workgroup.  First round (and every round of a group spread
    // over XCDs): an agent-scope add on the group's counter, polled with L1-bypassing loads; the
    // first arrival also carries the workgroup's XCD (a 6-bit arrival count per XCD above the
    // total).  Later rounds of a group that sits on one XCD: a flag word per workgroup in that
    // XCD's L2 (plain store, L1-bypassing polls of the 32 flags by one wave).
    auto tile_barrier = [&](bool first) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        VM_PTS(prof_ph, 5);
        if (wave == 0) {
            const bool by_flags = pure_known && !wt;
            const unsigned long long t0 = wall_clock64();
            if (by_flags) {
                if (lane == 0)
                    flg[part] = rounds + 1u;
                bool done = false;
                for (uint32_t spin = 0; !done; ++spin) {
                    const uint32_t v = lane < VM_PASS_PARTS ? ldc(flg + lane) : rounds + 1u;
                    done = __all(v >= rounds + 1u);
                    if (!done) {
                        __builtin_amdgcn_s_sleep(1);
                        if ((spin & 63u) == 63u && wall_clock64() - t0 > VM_PASS_TIMEOUT_TICKS) {
                            if (lane == 0) {
                                S.go = 0;
                                __hip_atomic_store((vm_g_u32 *)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            break;
                        }
                    }
                }
            } else if (lane == 0) {
                const unsigned long long add = 1ull | (first ? 1ull << (8 + 6 * xcc) : 0ull);
                __hip_atomic_fetch_add((vm_g_u64 *)bar, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t target = (uint32_t)VM_PASS_PARTS * (rounds + 1u);
                unsigned long long seen;
                while (((uint32_t)(seen = __hip_atomic_load((vm_g_cu64 *)bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xFFu) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > VM_PASS_TIMEOUT_TICKS) { // never hang the device: report and go on
                        S.go = 0;
                        __hip_atomic_store((vm_g_u32 *)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                if (first) { // on how many XCDs does the group sit?
                    int nx = 0;
                    for (int k = 0; k < 8; ++k)
                        nx += ((seen >> (8 + 6 * k)) & 63ull) != 0;
                    S.wt = nx > 1 ? 1u : 0u;
                }
            }
        }
        VM_PTS(prof_ph, 6);
        __syncthreads();
        ++rounds;
        if (S.go == 0)
            timed_out = true;
    };

    // phases 0..3, then the closing step (ph == 4): the last phase's records, second copy -> canonical arrays
    for (int ph = 0; ph < 5 && !timed_out; ++ph) {
        const bool closing = ph == 4;
        const int pi = closing ? 0 : ph >> 1, pj = closing ? 0 : ph & 1;
        const bool src_t = ph == 2 || ph == 4; // which copy holds the sums before the last phase's records
        const uint32_t epoch = epoch0 + (uint32_t)ph, pe = epoch - 1u, want = (pe << 2) | 1u;
        // records of the last phase (read) and of this one (written): both sets live in the workspace
        const uint32_t o_rtag = (pe & 1u) ? o_tag2 : 0u, o_ra = (pe & 1u) ? o_a2 : o_a, o_rb = (pe & 1u) ? o_b2 : o_b;
        const uint32_t o_wtag = (epoch & 1u) ? o_tag2 : 0u, o_wa = (epoch & 1u) ? o_a2 : o_a, o_wb = (epoch & 1u) ? o_b2 : o_b;
        // the copy of the sums and of the mask that is read (s) and the one that is written (d)
        const char *const sbase = src_t ? wb : sb, *const dbase = src_t ? sb : wb;
        const uint32_t os_mean = src_t ? o_mean2 : o_mean, os_var = src_t ? o_var2 : o_var, os_tpsb = src_t ? o_tpsb2 : o_tpsb,
                       os_cross = src_t ? o_cross2 : o_cross, os_value = src_t ? o_value2 : o_value, os_imp = src_t ? o_imp2 : o_imp;
        const uint32_t od_mean = src_t ? o_mean : o_mean2, od_var = src_t ? o_var : o_var2, od_tpsb = src_t ? o_tpsb : o_tpsb2,
                       od_cross = src_t ? o_cross : o_cross2, od_value = src_t ? o_value : o_value2, od_imp = src_t ? o_imp : o_imp2;
        const int ppi = ((ph - 1) & 3) >> 1, ppj = (ph - 1) & 1; // parity class of the last phase's pixels (ph > 0)

        // test hook (vm_dbg_pass_force_timeout): one workgroup of group 0 walks away before the second barrier;
        // the rest of its group waits in vain, times out and raises the error word
        if ((force_wt & 2) && grp == 0 && part == 3 && ph == 1)
            break;
        const int px = ox + tx * 2 + pj, py = oy + ty * 2 + pi;
        const bool in_img = px < L.w && py < L.h;
        const int spx = in_img ? px : ox, spy = in_img ? py : oy; // a safe pixel for the loads of an idle wave
        VM_PTS(ph, 0);
        // ================= everything from memory in one round trip (no load depends on another): first every
        // byte offset, then the loads back to back.  (Address arithmetic between the loads costs a round trip
        // of its own as soon as it touches a register a load is still to fill -- the compiler's 32-bit
        // multiply-adds are v_mad_u64_u32 with an undefined high half: measured, a vmcnt(0) after six loads.)
        const int oxb = spx % 5, oyb = spy % 5, pbx = spx / 5, pby = spy / 5;
        const int begi = oyb >= 2 ? 1 : 0, begj = oxb >= 2 ? 1 : 0;
        const int pidx = mad24(spy, L.rs, spx);
        const uint32_t p4 = (uint32_t)pidx << 2, p8 = (uint32_t)pidx << 3;
        const uint32_t w4 = (uint32_t)wword << 2;
        // first the loads whose addresses are cheap: (d) the pixel's own state, (f) the ring neighbours' v,
        // the group's mask word -- in flight while the other offsets are worked out
        uint32_t f_ring;
        {
            const int k = sub & 7;
            const int rx = ((0x06A4 >> (2 * k)) & 3) - 1, ry = ((0x6A40 >> (2 * k)) & 3) - 1;
            const int nx = spx + rx, ny = spy + ry;
            const bool in = nx >= 0 && nx < L.w && ny >= 0 && ny < L.h;
            f_ring = (uint32_t)(in ? mad24(ny, L.rs, nx) : pidx) << 3;
        }
        __builtin_amdgcn_sched_barrier(0);
        PixelCtx c;
        c.px = spx;
        c.py = spy;
        c.idx = pidx;
        c.v = ldo<float2>(sb, p8);
        c.old_luma = ldo<float2>(sb, o_luma + p8);
        c.ui_b = ldo<float2>(sb, o_uib + p8);
        c.ui_axy = *(vm_g_cf32 *)(sb + (o_uiaxy + p4));
        RingLanes ring;
        ring.mine = ldo<float2>(sb, f_ring);
        const uint32_t oword = ldo<uint32_t>(sbase, os_imp + w4);
        __builtin_amdgcn_sched_barrier(0);
        // (a) mask words of the 2 x 2 blocks the pixel's window reaches + the last phase's tags in them
        // (lanes 0-31: block column begj - 1, lanes 32-63: begj; round r: block row begi - 1 + r)
        uint32_t a_word[2], a_tag[2];
        bool a_owned[2], a_ok[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int bx = pbx + begj - 1 + (hi ? 1 : 0), by = pby + begi - 1 + r;
            // words the tile does not own hold bits of gap pixels only as far as this pixel is
            // concerned: always the canonical array (their owner may rewrite them, never those bits)
            a_owned[r] = bx >= own_bx0 && bx < own_bx0 + own_nx && by >= own_by0 && by < own_by0 + own_ny;
            a_word[r] = (uint32_t)mad24(by + 1, L.imp_rs, bx + 1) << 2;
            const int x = 5 * bx + sub % 5, y = 5 * by + sub / 5;
            a_ok[r] = ph > 0 && sub < 25 && x >= ox && x < tx1 && y >= oy && y < ty1;
            a_tag[r] = o_rtag + ((uint32_t)(a_ok[r] ? mad24(y, L.rs, x) : pidx) << 2);
        }
        // (b) the mask word this wave folds for the group (word `slot` of the owned ones): its pixels' tags
        bool b_ok;
        uint32_t b_tag;
        {
            const int x = 5 * wbx + sub % 5, y = 5 * wby + sub / 5;
            b_ok = ph > 0 && has_word && sub < 25 && x >= ox && x < tx1 && y >= oy && y < ty1;
            b_tag = o_rtag + ((uint32_t)(b_ok ? mad24(y, L.rs, x) : pidx) << 2);
        }
        // (c) the last phase's tags and records within +-4 of the pixel: position sub < 25 of the 5 x 5 grid
        // of that phase's parity class; lanes 0-31 fetch rec_a, lanes 32-63 rec_b (speculatively: a record is
        // used only where the tag says "committed in that phase")
        const int sx0 = -4 + ((ppj ^ pj) & 1), sy0 = -4 + ((ppi ^ pi) & 1);
        bool c_ok;
        uint32_t c_tag, c_rec;
        {
            // (relative to the slot's pixel position even where that lies outside the image: the cells such a
            // slot owns -- the halo beside a tile the border cuts down to a sliver -- still take records)
            const int dx = sx0 + 2 * (sub % 5), dy = sy0 + 2 * (sub / 5);
            const int x = px + dx, y = py + dy;
            c_ok = ph > 0 && sub < 25 && dx <= 4 && dy <= 4 && x >= ox && x < tx1 && y >= oy && y < ty1;
            const uint32_t ri = (uint32_t)(c_ok ? mad24(y, L.rs, x) : pidx);
            c_tag = o_rtag + (ri << 2);
            c_rec = (hi ? o_rb : o_ra) + (ri << 4);
        }
        // (e) this lane's cell.  Lanes sub < 25: the window cell (sub % 5 - 2, sub / 5 - 2) of the pixel.
        // Lanes sub >= 25 (7 per half): the cells this slot owns OUTSIDE its window -- an edge slot owns the
        // halo column / row beside it, 3 away from its pixel (the records that reach such a cell are those of
        // the tile's outermost pixels, within the +-4 this wave stages anyway).  Halo coordinates: (0, 0) =
        // cell (ox - 2, oy - 2).
        const int phx = 2 + 2 * tx + pj, phy = 2 + 2 * ty + pi;
        const int ohx0 = tx == 0 ? 0 : phx, ohx1 = tx == 31 ? VM_HALO_W - 1 : phx + 1;
        const int ohy0 = ty == 0 ? 0 : phy, ohy1 = ty == 7 ? VM_HALO_H - 1 : phy + 1;
        const int wi_ = (sub * 13) >> 6, wj_ = sub - wi_ * 5;
        int chx, chy;
        if (sub < 25) {
            chx = phx + wj_ - 2;
            chy = phy + wi_ - 2;
        } else {
            const int xcol = (tx == 0 && pj == 1) ? 0 : ((tx == 31 && pj == 0) ? VM_HALO_W - 1 : -1);
            const int xrow = (ty == 0 && pi == 1) ? 0 : ((ty == 7 && pi == 0) ? VM_HALO_H - 1 : -1);
            const int ncol = xcol >= 0 ? ohy1 - ohy0 + 1 : 0;
            const int e = (sub - 25) + (hi ? 7 : 0);
            chx = chy = -1;
            if (e < ncol) {
                chx = xcol;
                chy = ohy0 + e;
            } else if (xrow >= 0) {
                int k = ohx0 + (e - ncol);
                if (xcol >= 0 && xcol == ohx0)
                    ++k; // the corner cell went with the column
                if (k <= ohx1 && k != xcol) {
                    chx = k;
                    chy = xrow;
                }
            }
        }
        const int qx = ox - 2 + chx, qy = oy - 2 + chy;
        // (a cell can lie in the image while the slot's pixel does not -- the halo left of / above a
        // tile cut down to a sliver by the image border: it is still owned, and copied, by that slot)
        const bool cell_ok = chx >= 0 && qx >= 0 && qx < L.w && qy >= 0 && qy < L.h;
        const bool okc = cell_ok && sub < 25; // a window neighbour of the pixel
        const uint32_t gi = (uint32_t)(cell_ok ? mad24(qy, L.rs, qx) : pidx);
        const uint32_t g4 = gi << 2, g8 = gi << 3;
        const float tps_axy = S.tps[(border_class(spy, L.h) * 5 + border_class(spx, L.w)) * 25 + 12] / 2;
        __builtin_amdgcn_sched_barrier(0);
        // ---- the loads
        uint32_t mword[2], mcanon[2], mtag[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            mword[r] = ldo<uint32_t>(sbase, os_imp + a_word[r]);
            mcanon[r] = ldo<uint32_t>(sb, o_imp + a_word[r]);
            mtag[r] = ldo<uint32_t>(wb, a_tag[r]);
        }
        uint32_t otag = ldo<uint32_t>(wb, b_tag);
        uint32_t ctag = ldo<uint32_t>(wb, c_tag);
        const float4 crec = ldo<float4>(wb, c_rec);
        float2 m = ldo<float2>(sbase, os_mean + g8), q = ldo<float2>(sbase, os_var + g8), tb = ldo<float2>(sbase, os_tpsb + g8);
        float cr = ldo<float>(sbase, os_cross + g4), val = ldo<float>(sbase, os_value + g4);
        c.tref = make_float2(0, 0);
        c.tmask = 0.0f;
        if (L.temp_mask) { // uniform in the launch
            c.tref = L.temp_ref[pidx];
            c.tmask = L.temp_mask[pidx];
        }
        __builtin_amdgcn_sched_barrier(0);
        c.tps_axy = tps_axy;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (!a_owned[r])
                mword[r] = mcanon[r];
            if (!a_ok[r])
                mtag[r] = 0;
        }
        if (!b_ok)
            otag = 0;
        if (!c_ok)
            ctag = 0;

#ifdef VM_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // profiling build only: when have the loads landed?
#endif
        VM_PTS(ph, 1);
        // ================= the group's mask word: a committed pixel sets its bit, a hit that did not move clears it
        if (ph > 0) {
            const uint32_t setb = (uint32_t)__ballot(!hi && otag == want);
            const uint32_t clrb = (uint32_t)__ballot(!hi && otag == ((pe << 2) | 2u));
            if (has_word && lane == 0)
                sto<uint32_t>(dbase, od_imp + w4, (oword | setb) & ~clrb, wt);
        }
        // ================= the last phase's records through LDS, folded into this lane's cell (fold_cell's order)
        bool touched = false;
        if (ph > 0) {
            const uint32_t cset = (uint32_t)__ballot(!hi && ctag == want); // wave-uniform
            if (cset) {
                if (sub < 25 && ctag == want)
                    S.rec[wave][(sub / 5 + 1) * 7 + sub % 5 + 1][hi ? 1 : 0] = crec;
                __builtin_amdgcn_wave_barrier();
                // The records that reach this lane's cell: the (at most 3 x 3) positions (i0 + a, j0 + b) of the
                // staged grid within +-2 of it.  All LDS reads first, then the sums in row-major order of the
                // committed pixels (fold_cell's order).
                const int cdx = chx - phx, cdy = chy - phy;                             // my cell, relative to the pixel
                const int i0 = (cdx - 2 - sx0 + 1) >> 1, j0 = (cdy - 2 - sy0 + 1) >> 1; // ceil((cd - 2 - s0) / 2): -1 .. 3
                const int e0x = sx0 + 2 * i0 - cdx, e0y = sy0 + 2 * j0 - cdy;           // offset of position (i0, j0) from my cell: -2 or -1
                const float4 *rbase = &S.rec[wave][(j0 + 1) * 7 + i0 + 1][0];
                // which of the 3 x 3 hold a commit: row j of the census is bits 5 j .. 5 j + 4
                uint32_t win[3];
                int bcx[3], bcy[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int jj = j0 + t;
                    const uint32_t row = (jj >= 0 && jj <= 4) ? (cset >> (5 * jj)) & 31u : 0u;
                    win[t] = cell_ok ? ((row << 1) >> (i0 + 1)) & 7u : 0u; // bit a: position (i0 + a, jj); i0 + 1 >= 0
                    if (e0y + 2 * t > 2)
                        win[t] = 0;
                    bcx[t] = border_class(px + sx0 + 2 * (i0 + t), L.w);
                    bcy[t] = border_class(py + sy0 + 2 * jj, L.h);
                }
                if (e0x + 4 > 2) { // the third column is out of reach (e0x = -1)
                    win[0] &= 3u;
                    win[1] &= 3u;
                    win[2] &= 3u;
                }
                float4 ra[9], rb[9];
                float kk[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int a = k % 3, t = k / 3;
                    ra[k] = rbase[(t * 7 + a) * 2];
                    rb[k] = rbase[(t * 7 + a) * 2 + 1];
                    const bool u = (win[t] >> a) & 1u;
                    kk[k] = S.tps[u ? (bcy[t] * 5 + bcx[a]) * 25 + (2 - e0y - 2 * t) * 5 + (2 - e0x - 2 * a) : 0];
                }
#pragma unroll
                for (int k0 = 0; k0 < 9; ++k0) {
#if VM_EXACT
                    // vm_set_commit_order: bit 0 = reversed, bit 1 = column-major (k = row * 3 + column)
                    const int kr = (P.commit_order & 1) ? 8 - k0 : k0;
                    const int k = (P.commit_order & 2) ? (kr % 3) * 3 + kr / 3 : kr;
#else
                    const int k = k0;
#endif
                    if ((win[k / 3] >> (k % 3)) & 1u) {
                        touched = true;
                        m.x += ra[k].x;
                        m.y += ra[k].y;
                        q.x += ra[k].z;
                        q.y += ra[k].w;
                        cr += rb[k].x;
                        tb.x += rb[k].y * kk[k];
                        tb.y += rb[k].z * kk[k];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (touched) {
                const float counter = (float)(window_count(qy, L.h) * window_count(qx, L.w));
                val = ssim_value(m.x, m.y, q.x, q.y, cr, counter, P.ssim_clamp);
            }
            // the cells this slot owns go to the other copy of the sums
            const bool owned = cell_ok && chx >= ohx0 && chx <= ohx1 && chy >= ohy0 && chy <= ohy1 && (sub >= 25 || !hi);
            if (owned) {
                sto<float2>(dbase, od_mean + g8, m, wt);
                sto<float2>(dbase, od_var + g8, q, wt);
                sto<float>(dbase, od_cross + g4, cr, wt);
                sto<float2>(dbase, od_tpsb + g8, tb, wt);
                sto<float>(dbase, od_value + g4, val, wt);
            }
        }
        VM_PTS(ph, 2);
        if (closing)
            break;

        // ================= mask test (get_improve_mask_idx, morph.cu:621-646) on the words + the last phase's tags
        bool hit = false;
        {
            const uint32_t *ib = S.imp + (oyb * 5 + oxb) * 9;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned long long sb = __ballot(mtag[r] == want), cb = __ballot(mtag[r] == ((pe << 2) | 2u));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t w0 = (uint32_t)__shfl(mword[r], h * 32);
                    const uint32_t word = (w0 | ((uint32_t)(sb >> (32 * h)) & 0x1FFFFFFu)) & ~((uint32_t)(cb >> (32 * h)) & 0x1FFFFFFu);
                    if (word & ib[(begi + r) * 3 + begj + h])
                        hit = true;
                }
            }
            hit = hit && in_img;
        }
        VM_PTS(ph, 3);
        if (hit) { // wave-uniform
            uint32_t state = 2;
            float2 step = make_float2(0, 0), luma = make_float2(0, 0);
            uint32_t n_eval = 0;
            ++my_cand;
            if (!pixel_locked(L, P.bcond, px, py)) {
                // tps.b of the pixel itself is the folded value of its own cell (lane 12)
                c.tps_b.x = __shfl(tb.x, 12, 32);
                c.tps_b.y = __shfl(tb.y, 12, 32);
                bool ok;
#if VM_EXACT
                NbX nb;
                nb.ok = okc;
                nb.m = m;
                nb.q = q;
                nb.cr = cr;
                nb.val = val;
                nb.counter = okc ? (float)(window_count(qy, L.h) * window_count(qx, L.w)) : 25.0f;
                ok = decide_with64(
                    L, P, c, [&](float dx, float dy) { return energy_x32(L, P, nb, c, dx, dy); }, ring, hi, step, n_eval);
                if (ok) { // the lumas commit_pixel_motion samples (morph.cu:997-1003)
                    const float nvx = c.v.x + step.x, nvy = c.v.y + step.y;
                    luma.x = tap(L.img0, L.w, L.h, L.rs, px - nvx + 0.5f, py - nvy + 0.5f);
                    luma.y = tap(L.img1, L.w, L.h, L.rs, px + nvx + 0.5f, py + nvy + 0.5f);
                }
#else
                Nb1 nb;
                if (is_interior(L, px, py)) { // wave-uniform: one pixel per wave
                    nb1_make<true>(nb, L, okc, qx, qy, m, q, cr, val);
                    ok = decide64<true>(L, P, nb, ring, c, sub, hi, step, luma, n_eval);
                } else {
                    nb1_make<false>(nb, L, okc, qx, qy, m, q, cr, val);
                    ok = decide64<false>(L, P, nb, ring, c, sub, hi, step, luma, n_eval);
                }
#endif
                if (ok)
                    state = 1;
            }
            VM_PTS(ph, 4);
            my_eval += n_eval;
            if (lane == 0) {
                if (state == 1) {
                    const float2 ol = c.old_luma;
                    sto<float4>(wb, o_wa + (p4 << 2), make_float4(luma.x - ol.x, luma.y - ol.y, luma.x * luma.x - ol.x * ol.x,
                                                                luma.y * luma.y - ol.y * ol.y), wt);
                    sto<float4>(wb, o_wb + (p4 << 2), make_float4(luma.x * luma.y - ol.x * ol.y, step.x, step.y, 0.0f), wt);
                    sto<float2>(sb, o_luma + p8, luma, wt);
                    sto<float2>(sb, o_uib + p8, make_float2(c.ui_b.x + 2 * step.x * c.ui_axy, c.ui_b.y + 2 * step.y * c.ui_axy), wt);
                    sto<float2>(sb, p8, make_float2(c.v.x + step.x, c.v.y + step.y), wt);
                }
                sto<uint32_t>(wb, o_wtag + p4, (epoch << 2) | state, wt);
            }
            if (state == 1)
                ++my_commit;
        }
        prof_ph = ph;
        tile_barrier(ph == 0);
        VM_PTS(ph, 7);
        if (ph == 0 && !timed_out) {
            pure_known = true;
            if (!wt && S.wt) {
                // The group turned out to sit on more than one XCD: its phase-0 stores were plain and
                // may still be in another XCD's L2.  Write them back (agent-scope release), meet once
                // more, and store write-through from now on.
                if (tid == 0) // every wave's stores have left (the barrier above): one write-back for the workgroup
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                wt = true;
                tile_barrier(false);
            }
        }
    }
    // ---- counts of the workgroup (plain stores; the next launch adds the slots up)
    if (lane == 0) {
        atomicAdd(&S.n_cand, my_cand);
        atomicAdd(&S.n_commit, my_commit);
        atomicAdd(&S.n_eval, my_eval);
    }
    __syncthreads();
    if (tid == 0) {
        if (S.n_commit)
            flags[iter_idx] = 1u; // every writer stores the same 1 (see k_step)
        *my_slot = make_uint4(S.n_cand, S.n_commit, S.n_eval, part == 0 ? 1u : 0u);
        if (b == 0)
            clock_probe(stats0 + (size_t)iter_idx * VM_STAT_WORDS, true);
#ifdef VM_PROF
        if (b < 256)
            vm_prof_buf[8192 + b * 2 + 1] = wall_clock64();
#endif
    }
}

// closes a batch of PASS launches: the counts the last launch left in its slots
__global__ __launch_bounds__(64) void SUF(k_pass_tail)(uint32_t *__restrict__ stats, const uint32_t *__restrict__ slots_prev,
                                                        int prev_iter_idx, int nslot_prev, int ntiles, int ngroups, int cap, int spread)
{
    pass_sum_slots(stats, slots_prev, prev_iter_idx, nslot_prev, ntiles, ngroups, cap, (int)threadIdx.x, spread != 0);
}

__global__ void SUF(k_next_iter)(int *iter_dev, int set, int value)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *iter_dev = set ? value : *iter_dev + value;
}

} // namespace

// ---------------------------------------------------------------------------
// launchers
#if defined(VM_PROF) && !VM_EXACT
extern "C" int vm_dbg_prof_read(void *dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(vm_prof_buf), bytes);
}
#endif

// `views`: device array of nbatch level views (frame pairs of one batch, same w x h);
// flags/stats: nbatch rows of `cap` iterations
void SUF(vm_launch_optimize)(const VmLevelView *views, int nbatch, int cap, int w, int h, const VmKParams &P,
                             const uint32_t *tables, int offx, int offy, uint32_t *flags, uint32_t *stats,
                             int iter_idx, int fixed_work, int threads, const int *iter_dev, int dense,
                             uint32_t *tile_list, hipStream_t s)
{
    dim3 b(threads), g((w + VM_PITCH_X - 1) / VM_PITCH_X, (h + VM_PITCH_Y - 1) / VM_PITCH_Y, nbatch);
#if !VM_EXACT
    if (!dense && tile_list) { // the listed form of a pruned pass (k_tile_scan)
        const int nwords = ((w + 4) / 5 + 2) * ((h + 4) / 5 + 2), tiles = (int)(g.x * g.y);
        hipLaunchKernelGGL(SUF(k_tile_scan), dim3(std::min((nwords + 255) / 256, 32), 1, nbatch), dim3(256), 0, s, views, cap, offx,
                           offy, flags, iter_idx, fixed_work, iter_dev, tile_list, tiles);
        hipLaunchKernelGGL(SUF(k_optimize_listed), dim3(std::min(tiles * nbatch, VM_TILE_LIST_GRID)), b, 0, s, views, cap, P, tables,
                           offx, offy, flags, stats, iter_idx, iter_dev, tile_list, tiles, nbatch);
        return;
    }
    if (!dense) {
        hipLaunchKernelGGL(SUF(k_optimize)<false>, g, b, 0, s, views, cap, P, tables, offx, offy, flags, stats,
                           iter_idx, fixed_work, iter_dev);
        return;
    }
    if (dense == 2) { // the 128-VGPR form: two workgroups per CU
        hipLaunchKernelGGL((SUF(k_optimize)<true, 7, 4>), g, b, 0, s, views, cap, P, tables, offx, offy, flags, stats,
                           iter_idx, fixed_work, iter_dev);
        return;
    }
    // Levels of at most 32 tiles per pass (240x135 and below): most tiles touch the image border, so
    // most tiles run the border form of the dense line search in some of their waves and wait for
    // them; the interior form beside it only doubles the code the CU's waves execute at once.
    // Without it (the same bits: the border form computes the same window counts at run time), us per
    // dense pass (r03, tools/dev_dense.py): 30 x 120x68 187.5 -> 183.6, 30 x 240x135 623 -> 604, 3 x 240x135
    // 182 -> 175; on large levels the interior form is what most waves run (1080p x 8 pairs, before the
    // fixed fan-out: 36.6 ms per pass with it, 50.5 without).  A rule on the level, never on the batch.
    // (This form also has no lean bodies, see tile_sweep: a phase of <= 16 candidates takes the two-lane
    // search, so against the general kernel its results move by FAST rounding -- the same for a pair
    // alone and in a batch, since the form follows from the level.)
    static const char *noint = getenv("VM_DENSE_NOINT"); // dev switch: 0 / 1 = never / always
    if (noint ? atoi(noint) != 0 : g.x * g.y <= 32) {
        hipLaunchKernelGGL((SUF(k_optimize)<true, VM_SMAX, VM_MIN_FANOUT, false>), g, b, 0, s, views, cap, P, tables, offx, offy,
                           flags, stats, iter_idx, fixed_work, iter_dev);
        return;
    }
#endif
    (void)dense;
    hipLaunchKernelGGL(SUF(k_optimize)<true>, g, b, 0, s, views, cap, P, tables, offx, offy, flags, stats, iter_idx,
                       fixed_work, iter_dev);
}

// a batch of `nit` iterations of the SPARSE schedule: list scan + one workgroup per pair
void SUF(vm_launch_optimize_sparse)(const VmLevelView *views, int nbatch, int cap, int w, int h, const VmKParams &P,
                                    const uint32_t *tables, uint32_t *flags, uint32_t *stats, int it0, int nit,
                                    int fixed_work, int threads, int dense, int lds_cap, int res_mode, hipStream_t s)
{
    const int nwords = ((w + 4) / 5 + 2) * ((h + 4) / 5 + 2);
    hipLaunchKernelGGL(SUF(k_sparse_scan), dim3((nwords + 255) / 256, 1, nbatch), dim3(256), 0, s, views);
#if !VM_EXACT
    if (!dense) {
        hipLaunchKernelGGL(SUF(k_sparse)<false>, dim3(1, 1, nbatch), dim3(threads), 0, s, views, cap, P, tables, flags,
                           stats, it0, nit, fixed_work, lds_cap, res_mode);
        return;
    }
#endif
    (void)dense;
    hipLaunchKernelGGL(SUF(k_sparse)<true>, dim3(1, 1, nbatch), dim3(threads), 0, s, views, cap, P, tables, flags, stats,
                       it0, nit, fixed_work, lds_cap, res_mode);
}

// the device iteration counter of graph-replayed sweeps: set it to, or advance it by, `value`
void SUF(vm_launch_next_iter)(int *iter_dev, int set, int value, hipStream_t s)
{
    hipLaunchKernelGGL(SUF(k_next_iter), dim3(1), dim3(64), 0, s, iter_dev, set, value);
}

// one pass (tile offset) in the SPLIT schedule: 4 phases x (decide, commit)
void SUF(vm_launch_optimize_split)(const VmLevelView *views, int nbatch, int cap, int w, int h,
                                   const VmKParams &P, const uint32_t *tables, int offx, int offy, int pass,
                                   uint32_t *flags, uint32_t *stats, int iter_idx, int fixed_work, int threads,
                                   int parts, hipStream_t s)
{
    const int gx = (w + VM_PITCH_X - 1) / VM_PITCH_X, gy = (h + VM_PITCH_Y - 1) / VM_PITCH_Y;
    for (int pi = 0; pi < 2; ++pi)
        for (int pj = 0; pj < 2; ++pj) {
            const uint32_t epoch = 1u + (uint32_t)((iter_idx * 4 + pass) * 4 + pi * 2 + pj);
            hipLaunchKernelGGL(SUF(k_decide), dim3(gx * gy * parts, 1, nbatch), dim3(threads), 0, s, views, cap, P,
                               tables, offx, offy, pi, pj, parts, epoch, flags, stats, iter_idx, fixed_work);
            hipLaunchKernelGGL(SUF(k_commit), dim3(gx, gy, nbatch), dim3(1024), 0, s, views, cap, P, tables, offx,
                               offy, pi, pj, epoch, flags, stats, iter_idx, fixed_work);
        }
}

// one phase of the STEP schedule: fold of the previous phase's records + line
// searches of this one; decide == 0: the closing fold-only launch of a batch of steps
void SUF(vm_launch_optimize_step)(const VmLevelView *views, int nbatch, int cap, int w, int h, const VmKParams &P,
                                  const uint32_t *tables, int offx, int offy, int pi, int pj, uint32_t epoch,
                                  uint32_t prev_epoch, int src, int decide, uint32_t *flags, uint32_t *stats,
                                  int iter_idx, int fixed_work, int threads, int parts, uint32_t *slots_cur,
                                  const uint32_t *slots_prev, int prev_iter_idx, hipStream_t s)
{
    const int gx = (w + VM_PITCH_X - 1) / VM_PITCH_X, gy = (h + VM_PITCH_Y - 1) / VM_PITCH_Y;
    const int n_fold = ((w + 63) / 64) * ((h + 15) / 16);
    const int nslot = gx * gy * parts;
    const dim3 grid(n_fold + (decide ? nslot : 0), 1, nbatch);
    if (threads <= 256)
        hipLaunchKernelGGL(SUF(k_step)<256>, grid, dim3(256), 0, s, views, cap, P, tables, offx, offy, pi, pj, parts,
                           epoch, prev_epoch, src, n_fold, flags, stats, iter_idx, fixed_work, slots_cur, slots_prev,
                           prev_iter_idx, nslot);
    else
        hipLaunchKernelGGL(SUF(k_step)<512>, grid, dim3(512), 0, s, views, cap, P, tables, offx, offy, pi, pj, parts,
                           epoch, prev_epoch, src, n_fold, flags, stats, iter_idx, fixed_work, slots_cur, slots_prev,
                           prev_iter_idx, nslot);
}

// one pass of the PASS schedule: four phases of every tile behind tile-local barriers.  `bar`:
// ngroups zeroed counters of this launch.  decide == 0: only the closing slot fold of a batch.
void SUF(vm_launch_optimize_pass)(const VmLevelView *views, int nbatch, int cap, int w, int h, const VmKParams &P,
                                  const uint32_t *tables, int offx, int offy, uint32_t epoch0, uint32_t *bar,
                                  uint32_t *flags, uint32_t *stats, int iter_idx, int fixed_work, uint32_t *slots_cur,
                                  const uint32_t *slots_prev, int prev_iter_idx, uint32_t *err, uint32_t *dbg,
                                  int decide, int force_wt, hipStream_t s)
{
    const int gx = (w + VM_PITCH_X - 1) / VM_PITCH_X, gy = (h + VM_PITCH_Y - 1) / VM_PITCH_Y;
    const int ntiles = gx * gy, ngroups = ntiles * nbatch;
    const int nblocks = (ngroups + 7) / 8 * 256;
    if (!decide) {
        hipLaunchKernelGGL(SUF(k_pass_tail), dim3(1), dim3(64), 0, s, stats, slots_prev, prev_iter_idx, nblocks, ntiles, ngroups, cap,
                           force_wt & 4);
        return;
    }
    hipLaunchKernelGGL(SUF(k_pass), dim3(nblocks), dim3(VM_PASS_T), 0, s, views, cap, P, tables, offx, offy, epoch0, ngroups,
                       ntiles, bar, flags, stats, iter_idx, fixed_work, slots_cur, slots_prev, prev_iter_idx, nblocks, err,
                       dbg, force_wt);
}

// how many workgroups of k_pass the device holds at once: a tile group's 32 workgroups wait for each other
// inside a launch, so a 256-workgroup chunk (8 groups) must be co-resident -- an MI355X in its default
// mode holds 256 (one per CU); a partitioned or smaller device does not, and gets the STEP schedule
int SUF(vm_pass_resident_blocks)(int device)
{
    hipDeviceProp_t pr;
    int per_cu = 0;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, SUF(k_pass), VM_PASS_T, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return per_cu * pr.multiProcessorCount;
}
