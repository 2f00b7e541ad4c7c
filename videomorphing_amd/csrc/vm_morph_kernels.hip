// vm_morph_kernels.hip -- HIP kernels of the halfway optimizer for gfx950.
//
// Compiled twice:  -DVM_EXACT=1 -ffp-contract=off  -> *_exact launchers
//                  -DVM_EXACT=0                    -> *_fast  launchers
// EXACT uses only IEEE +,-,*,/,sqrt in the reference's expression order, so a
// solve is bit-identical to the CPU oracle.  FAST lets the compiler fuse
// multiply-adds and uses v_rcp_f32 / v_sqrt_f32 (the analogue of the
// reference's --use_fast_math build).
//
// What the kernels compute follows Algorithm/morph.cu and upsample.cu of the
// reference (file:line cited per kernel); how they compute it is CDNA4-first:
// no texture units (manual bilinear taps on linear f32 images), all per-tile
// state including the improving-mask words staged in LDS, commits applied by a
// per-cell gather in a fixed order instead of LDS/global float atomics
// (deterministic, and no global atomics at all in the sweep), tile-level and
// phase-level early outs driven by the improving mask.
#include "vm_internal.h"

#ifndef VM_EXACT
#error "define VM_EXACT to 0 or 1"
#endif

#if VM_EXACT
#define SUF(name) name##_exact
#else
#define SUF(name) name##_fast
#endif

namespace {

__device__ __forceinline__ float fdiv(float a, float b)
{
#if VM_EXACT
    return a / b;
#else
    return a * __builtin_amdgcn_rcpf(b);
#endif
}

__device__ __forceinline__ float fsqrt(float a)
{
#if VM_EXACT
    return sqrtf(a);
#else
    return __builtin_amdgcn_sqrtf(a);
#endif
}

// border class of calc_border (morph.cu:39-81)
__device__ __forceinline__ int border_class(int p, int dim)
{
    return p < 2 ? p : (p == dim - 2 ? 3 : (p == dim - 1 ? 4 : 2));
}

// number of in-image pixels of the 5-wide window centred at p
__device__ __forceinline__ int window_count(int p, int dim)
{
    return min(p, 2) + min(dim - 1 - p, 2) + 1;
}

#if VM_EXACT
// ssim(), morph.cu:85-118, literally
__device__ __forceinline__ float ssim_value(float mx, float my, float vx, float vy, float cross,
                                            float counter, float clamp)
{
    if (counter <= 1)
        return 0;
    const float c2 = 58.5225f; // pow2(255*0.03)
    const float c3 = 29.26125f;
    mx = fdiv(mx, counter);
    my = fdiv(my, counter);
    vx = fdiv(vx - counter * mx * mx, counter);
    vy = fdiv(vy - counter * my * my, counter);
    vx = fmaxf(0.0f, vx);
    vy = fmaxf(0.0f, vy);
    cross = fdiv(cross - counter * mx * my, counter);
    float sx = fsqrt(vx), sy = fsqrt(vy);
    float c = fdiv(2 * sx * sy + c2, vx + vy + c2);
    float s = fdiv(fabsf(cross) + c3, sx * sy + c3);
    float value = c * s;
    return fmaxf(fminf(1.0f, value), clamp);
}
#else
// FAST form of ssim(): a, b = window means (sum * 1/n); sx2, sy2, sxy = raw second
// moment sums; n the window count, in = 1/n.  The variances are formed as
// (sum - n a a) / n like the reference does -- measured: forming them from
// pre-divided sums (E[x^2] - a^2) quantises the line search's tiny energy
// differences enough to cost 6 % of SSIM energy after 86 sweeps -- but with one
// rcp and one sqrt per evaluation:
//   c*s = (2 sx sy + c2)(|cov| + c3) / ((sx^2 + sy^2 + c2)(sx sy + c3)),  sx sy = sqrt(vx vy)
// Every SSIM value of FAST mode (stored ones and the trial ones of the line
// search) comes from this one function.
__device__ __forceinline__ float ssim_core(float a, float b, float sx2, float sy2, float sxy,
                                           float n, float in, float clamp)
{
    const float c2 = 58.5225f, c3 = 29.26125f;
    const float na = n * a, nb = n * b;
    const float vx = fmaxf(fmaf(-na, a, sx2) * in, 0.0f);
    const float vy = fmaxf(fmaf(-nb, b, sy2) * in, 0.0f);
    const float cov = fmaf(-na, b, sxy) * in;
    const float ss = __builtin_amdgcn_sqrtf(vx * vy);
    const float num = fmaf(2.0f, ss, c2) * (fabsf(cov) + c3);
    const float den = (vx + vy + c2) * (ss + c3);
    const float val = num * __builtin_amdgcn_rcpf(den);
    return fmaxf(fminf(val, 1.0f), clamp);
}

__device__ __forceinline__ float ssim_value(float mx, float my, float vx, float vy, float cross,
                                            float counter, float clamp)
{
    if (counter <= 1)
        return 0;
    const float in = __builtin_amdgcn_rcpf(counter);
    return ssim_core(mx * in, my * in, vx, vy, cross, counter, in, clamp);
}
#endif

// tex2D(linear, clamp, unnormalised) on a pitched f32 image: texel centres at
// i+0.5 (morph.cu:316-322); exact float weights
__device__ __forceinline__ float tap(const float *__restrict__ img, int w, int h, int rs, float x,
                                     float y)
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
    float t00 = img[j0 * rs + i0], t10 = img[j0 * rs + i1];
    float t01 = img[j1 * rs + i0], t11 = img[j1 * rs + i1];
    return (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
}

__device__ __forceinline__ float2 tap2(const float2 *__restrict__ img, int w, int h, int rs,
                                       float x, float y)
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
    float2 t00 = img[j0 * rs + i0], t10 = img[j0 * rs + i1];
    float2 t01 = img[j1 * rs + i0], t11 = img[j1 * rs + i1];
    float2 r;
    r.x = (1 - a) * (1 - b) * t00.x + a * (1 - b) * t10.x + (1 - a) * b * t01.x + a * b * t11.x;
    r.y = (1 - a) * (1 - b) * t00.y + a * (1 - b) * t10.y + (1 - a) * b * t01.y + a * b * t11.y;
    return r;
}

// ---------------------------------------------------------------------------
// kernel_initialize_level (morph.cu:173-244) + init_improving_mask (:246-260)
// One thread per pixel; the 5x5 gather of warped lumas runs on L1/L2-cached
// image rows.  60 B/pixel algorithmic traffic: HBM-bound, launched once per
// level.
__global__ __launch_bounds__(256) void SUF(k_init_level)(VmLevelView L, float ssim_clamp,
                                                         const uint32_t *__restrict__ tables)
{
    __shared__ float s_tps[625];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    for (int k = tid; k < 625; k += 256)
        s_tps[k] = __uint_as_float(tables[VM_TAB_TPS + k]);
    __syncthreads();

    // improving mask: grid-stride over the block-mask words
    {
        int nwords = L.imp_rs * L.imp_rows;
        int gtid = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid;
        int gstride = gridDim.x * gridDim.y * 256;
        for (int k = gtid; k < nwords; k += gstride) {
            int bx = k % L.imp_rs, by = k / L.imp_rs;
            L.impmask[k] = (bx == 0 || by == 0 || bx == L.imp_rs - 1 || by == L.imp_rows - 1)
                               ? 0u : ((1u << 25) - 1);
        }
    }

    const int x = blockIdx.x * 32 + threadIdx.x, y = blockIdx.y * 8 + threadIdx.y;
    if (x >= L.w || y >= L.h)
        return;
    const int By = border_class(y, L.h), Bx = border_class(x, L.w);
    const float *tp = s_tps + (By * 5 + Bx) * 25;
    int counter = 0;
    float mx = 0, my = 0, vx = 0, vy = 0, cross = 0, bx = 0, by = 0;
    float lcx = 0, lcy = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            int qx = x + j - 2, qy = y + i - 2;
            if (qx < 0 || qx >= L.w || qy < 0 || qy >= L.h)
                continue;
            float2 nv = L.v[qy * L.rs + qx];
            float tx = (float)qx + 0.5f, ty = (float)qy + 0.5f;
            float lx = tap(L.img0, L.w, L.h, L.rs, tx - nv.x, ty - nv.y);
            float ly = tap(L.img1, L.w, L.h, L.rs, tx + nv.x, ty + nv.y);
            float c = tp[i * 5 + j];
            bx += nv.x * c;
            by += nv.y * c;
            counter += 1;
            mx += lx;
            my += ly;
            vx += lx * lx;
            vy += ly * ly;
            cross += lx * ly;
            if (i == 2 && j == 2) {
                lcx = lx;
                lcy = ly;
            }
        }
    }
    const int idx = y * L.rs + x;
    L.luma[idx] = make_float2(lcx, lcy);
    L.mean[idx] = make_float2(mx, my);
    L.var[idx] = make_float2(vx, vy);
    L.cross[idx] = cross;
    L.value[idx] = ssim_value(mx, my, vx, vy, cross, (float)counter, ssim_clamp);
    L.tps_b[idx] = make_float2(bx, by);
    L.ui_axy[idx] = 0.0f;
    L.ui_b[idx] = make_float2(0.0f, 0.0f);
}

// ---------------------------------------------------------------------------
// UI constraint linearisation, morph.cu:345-388.  The reference does this on
// the host after a D2H copy of v; here one thread walks the (few) constraints
// in order on the device, so v never leaves HBM and the += order is the
// reference's.
__global__ void SUF(k_splat)(VmLevelView L, int w0, int h0, const vm_constraint *__restrict__ c,
                             int n)
{
    if (blockIdx.x != 0 || threadIdx.x != 0)
        return;
    for (int k = 0; k < n; ++k) {
        float x0 = (float)(((double)c[k].lx + 0.5) / w0 * L.w - 0.5f);
        float y0 = (float)(((double)c[k].ly + 0.5) / h0 * L.h - 0.5f);
        float x1 = (float)(((double)c[k].rx + 0.5) / w0 * L.w - 0.5f);
        float y1 = (float)(((double)c[k].ry + 0.5) / h0 * L.h - 0.5f);
        float weight = c[k].weight;
        float con_x = (x0 + x1) / 2.0f, con_y = (y0 + y1) / 2.0f;
        float vx = (x1 - x0) / 2.0f, vy = (y1 - y0) / 2.0f;
        int ylo = (int)floorf(con_y), yhi = (int)ceilf(con_y);
        int xlo = (int)floorf(con_x), xhi = (int)ceilf(con_x);
        for (int y = ylo; y <= yhi; ++y)
            for (int x = xlo; x <= xhi; ++x)
                if (x >= 0 && x < L.w && y >= 0 && y < L.h) {
                    int idx = y * L.rs + x;
                    float bw = (1 - fabsf(y - con_y)) * (1 - fabsf(x - con_x)) * weight;
                    float2 v = L.v[idx];
                    float2 b = L.ui_b[idx];
                    L.ui_axy[idx] += bw;
                    b.x += 2 * bw * (v.x - vx);
                    b.y += 2 * bw * (v.y - vy);
                    L.ui_b[idx] = b;
                }
    }
}

// ---------------------------------------------------------------------------
// upsample(), spatial half: upsample.cu:260-286 (internal_vector_to_image,
// rod::kernel_upsample<box_sampler>, conv_to_block_of_arrays fused into one
// pass: 10 B per destination pixel)
__global__ __launch_bounds__(256) void SUF(k_upsample)(float2 *__restrict__ dst, int dw, int dh,
                                                       int drs, const float2 *__restrict__ src,
                                                       int sw, int sh, int srs)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh)
        return;
    const float tw = (float)sw / dw, th = (float)sh / dh;
    const float mx = (float)dw / sw, my = (float)dh / sh;
    float2 s = tap2(src, sw, sh, srs, (x + 0.5f) * tw, (y + 0.5f) * th);
    dst[y * drs + x] = make_float2(s.x * mx, s.y * my);
}

// ---------------------------------------------------------------------------
// The sweep kernel: kernel_optimize_level and its device helpers,
// morph.cu:592-1345.
//
// One workgroup of T threads (T = 256..1024, chosen per level by the host)
// relaxes one 64x16 tile through the reference's four Jacobi phases.  Per phase:
//   1. the (at most 256) pixels of the phase are tested against the improving
//      mask and the candidates are compacted into an LDS list;
//   2. every candidate is handed to a group of L consecutive lanes, L the
//      largest power of two <= T / #candidates (4..32; 1 in EXACT mode).  The L
//      lanes split the 25 window neighbours of the pixel between them, keep
//      their share of the window sums in registers for the whole line search,
//      and combine the per-neighbour SSIM terms with DPP butterflies, so the
//      ~21 dependent energy evaluations of a pixel cost 25/L SSIM evaluations
//      each instead of 25.  Dense phases fill the machine with pixels, sparse
//      phases (the common case once the improving mask has pruned the level)
//      fill it with neighbours: the latency of a nearly idle tile, which bounds
//      the launch, drops by up to 25x;
//   3. accepted moves are published as per-pixel records and every tile+halo
//      cell gathers the records of the pixels whose window covers it, in a
//      fixed order (no float atomics anywhere, LDS or global).

struct TileLds {
    float2 mean[VM_NCELL], var[VM_NCELL], tpsb[VM_NCELL];
    float cross[VM_NCELL], value[VM_NCELL];
    // per phase pixel (slot = (y>>1)*32 + (x>>1) inside the tile)
    float2 d_mean[256], d_var[256], d_step[256];
    float d_cross[256];
    int d_ok[256];           // 0: untouched, 1: commit, 2: candidate that failed
    int list[256];           // compacted candidate slots
    int n_act;
    float tps[625];
    uint32_t imp[225];
    uint32_t mask[6][16];    // improving-mask words covering the tile +-1 block
};

struct PixelCtx {
    int px, py;      // image coordinates
    int hc;          // LDS cell of (px-2, py-2): top-left of the 5x5 window
    int idx;         // global element index
    float2 v, old_luma;
    float tps_axy, ui_axy;
    float2 tps_b, ui_b;
};

#if VM_EXACT
// ---- EXACT: literal ssim_change (morph.cu:671-728) + energy_change (:730-761),
// flag == false; one lane per pixel, neighbours read from LDS in row-major order
struct NbCache {};
__device__ __forceinline__ void nb_load(NbCache &, const VmLevelView &, const VmKParams &,
                                        const TileLds &, const PixelCtx &, int, int) {}

__device__ __forceinline__ float energy_change(const VmLevelView &L, const VmKParams &P,
                                               const TileLds &S, const NbCache &,
                                               const PixelCtx &c, float dx, float dy, int)
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
            const int cell = c.hc + i * VM_HALO_W + j;
            const float counter = (float)(ny * window_count(qx, L.w));
            const float2 m = S.mean[cell], q = S.var[cell];
            const float ns = ssim_value(m.x + dmx, m.y + dmy, q.x + dvx, q.y + dvy,
                                        S.cross[cell] + dcross, counter, P.ssim_clamp);
            change += S.value[cell] - ns;
        }
    }
    float v_tps = c.tps_axy * (dx * dx + dy * dy);
    v_tps += c.tps_b.x * dx;
    v_tps += c.tps_b.y * dy;
    float v_ui = c.ui_axy * (dx * dx + dy * dy);
    v_ui += c.ui_b.x * dx;
    v_ui += c.ui_b.y * dy;
    return (P.w_ui * v_ui + P.w_ssim * change + 0.0f) * L.inv_wh + P.w_tps * v_tps;
}
#define VM_MIN_FANOUT 1
#define VM_MAX_FANOUT 1
#else
// ---- FAST: the same energy, evaluated by L lanes per pixel.  Lane `sub` owns the
// window neighbours sub, sub+L, sub+2L, ... (at most VM_SMAX of them) and keeps
// their sums, pre-divided by the window count, in registers.
#define VM_SMAX 7
#define VM_MIN_FANOUT 4
#ifndef VM_MAX_FANOUT
#define VM_MAX_FANOUT 32
#endif
struct NbCache {
    float A[VM_SMAX], B[VM_SMAX];                   // window means (sum / n)
    float VX[VM_SMAX], VY[VM_SMAX], X[VM_SMAX];       // raw second-moment sums
    float N[VM_SMAX], IN[VM_SMAX], M[VM_SMAX];         // count, 1/count, validity
    float VAL[VM_SMAX]; // current SSIM value of the neighbour (the differences value - new
                        // are summed, as the reference does: they are 1e-3..1e-6 of the values)
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
// sum over the aligned group of Lf lanes (Lf = 4, 8, 16 or 32, uniform in the
// workgroup); every lane of the group ends with the same bits
__device__ __forceinline__ float group_sum(float x, int Lf)
{
    x += dpp_xor1(x);
    x += dpp_xor2(x);
    if (Lf >= 8) x += dpp_half_mirror(x);
    if (Lf >= 16) x += dpp_mirror(x);
    if (Lf >= 32) x += swz_xor16(x);
    return x;
}

__device__ __forceinline__ void nb_load(NbCache &nb, const VmLevelView &L, const VmKParams &P,
                                        const TileLds &S, const PixelCtx &c, int sub, int Lf)
{
#pragma unroll
    for (int j = 0; j < VM_SMAX; ++j) {
        const int k = sub + j * Lf;
        const int i = k / 5, jj = k - i * 5;
        const int qx = c.px + jj - 2, qy = c.py + i - 2;
        const bool ok = k < 25 && qx >= 0 && qx < L.w && qy >= 0 && qy < L.h;
        const int cell = ok ? c.hc + i * VM_HALO_W + jj : 0;
        const float n = ok ? (float)(window_count(qy, L.h) * window_count(qx, L.w)) : 1.0f;
        const float in = ok ? __builtin_amdgcn_rcpf(n) : 0.0f;
        const float2 m = S.mean[cell], q = S.var[cell];
        nb.A[j] = m.x * in;
        nb.B[j] = m.y * in;
        nb.VX[j] = q.x;
        nb.VY[j] = q.y;
        nb.X[j] = S.cross[cell];
        nb.N[j] = n;
        nb.IN[j] = in;
        nb.M[j] = ok ? 1.0f : 0.0f;
        nb.VAL[j] = S.value[cell];
    }
}

__device__ __forceinline__ float energy_change(const VmLevelView &L, const VmKParams &P,
                                               const TileLds &, const NbCache &nb,
                                               const PixelCtx &c, float dx, float dy, int Lf)
{
    const float vx = c.v.x + dx, vy = c.v.y + dy;
    const float lx = tap(L.img0, L.w, L.h, L.rs, c.px - vx + 0.5f, c.py - vy + 0.5f);
    const float ly = tap(L.img1, L.w, L.h, L.rs, c.px + vx + 0.5f, c.py + vy + 0.5f);
    const float dmx = lx - c.old_luma.x, dmy = ly - c.old_luma.y;
    const float dvx = lx * lx - c.old_luma.x * c.old_luma.x;
    const float dvy = ly * ly - c.old_luma.y * c.old_luma.y;
    const float dcross = lx * ly - c.old_luma.x * c.old_luma.y;
    float acc = 0;
#pragma unroll
    for (int j = 0; j < VM_SMAX; ++j) {
        if (j * Lf < 25) { // uniform in the workgroup
            const float in = nb.IN[j];
            const float val = ssim_core(fmaf(dmx, in, nb.A[j]), fmaf(dmy, in, nb.B[j]),
                                        nb.VX[j] + dvx, nb.VY[j] + dvy, nb.X[j] + dcross,
                                        nb.N[j], in, P.ssim_clamp);
            acc = fmaf(nb.M[j], nb.VAL[j] - val, acc);
        }
    }
    const float change = group_sum(acc, Lf);
    const float dd = dx * dx + dy * dy;
    const float v_tps = fmaf(c.tps_axy, dd, fmaf(c.tps_b.x, dx, c.tps_b.y * dy));
    const float v_ui = fmaf(c.ui_axy, dd, fmaf(c.ui_b.x, dx, c.ui_b.y * dy));
    return (P.w_ui * v_ui + P.w_ssim * change) * L.inv_wh + P.w_tps * v_tps;
}
#endif

// fover_update_isec_min, morph.cu:794-831
__device__ __forceinline__ void fover_isec(float cx, float cy, float gx, float gy, float e0x,
                                           float e0y, float e1x, float e1y, float &t_min)
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

// fover_calc_isec_min (morph.cu:833-870) with fover_calc_vtx (:782-792, note
// the `p - off` of the original) for one sign
__device__ __forceinline__ void fover_ring(const VmLevelView &L, int px, int py, float sgn,
                                           float vx, float vy, float gx, float gy, float &t_min)
{
    const int rx[8] = {-1, 0, 1, 1, 1, 0, -1, -1};
    const int ry[8] = {-1, -1, -1, 0, 1, 1, 1, 0};
    float ex[8], ey[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float ux = vx, uy = vy;
        int qx = px + rx[k], qy = py + ry[k];
        if (qx >= 0 && qx < L.w && qy >= 0 && qy < L.h) {
            float2 nv = L.v[qy * L.rs + qx];
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

__global__ __launch_bounds__(1024) void SUF(k_optimize)(VmLevelView L, VmKParams P,
                                                        const uint32_t *__restrict__ tables,
                                                        int offx, int offy,
                                                        uint32_t *__restrict__ flags,
                                                        uint32_t *__restrict__ stats, int iter_idx,
                                                        int fixed_work)
{
    __shared__ TileLds S;
    const int tid = threadIdx.x, T = blockDim.x;

    // converged in the previous iteration: nothing left to do (sticky)
    if (!fixed_work && iter_idx > 0 && flags[iter_idx - 1] == 0)
        return;

    const int ox = blockIdx.x * VM_PITCH_X + offx, oy = blockIdx.y * VM_PITCH_Y + offy;
    if (ox >= L.w || oy >= L.h)
        return;

    // --- improving-mask words of the tile and its ring of neighbour blocks ---
    const int bx0 = ox / 5 - 1, by0 = oy / 5 - 1;
    const int bx1 = min(ox + VM_TILE_W - 1, L.w - 1) / 5 + 1;
    const int by1 = min(oy + VM_TILE_H - 1, L.h - 1) / 5 + 1;
    const int nbx = bx1 - bx0 + 1, nby = by1 - by0 + 1; // <= 16, <= 6
    uint32_t mymask = 0;
    if (tid < nbx * nby) {
        int mx = tid % nbx, my = tid / nbx;
        mymask = L.impmask[(by0 + my + 1) * L.imp_rs + (bx0 + mx + 1)];
        S.mask[my][mx] = mymask;
    }
    // tile-level early out: no set bit anywhere near the tile means no pixel of
    // it is a candidate in any phase, and re-deriving the SSIM values from
    // unchanged sums reproduces them bit for bit
    if (!__syncthreads_or(mymask != 0))
        return;

    for (int k = tid; k < 625; k += T)
        S.tps[k] = __uint_as_float(tables[VM_TAB_TPS + k]);
    for (int k = tid; k < 225; k += T)
        S.imp[k] = tables[VM_TAB_IMP + k];

    // --- LoadSSIM (morph.cu:1214-1234) + the tile's tps.b ---
    for (int c = tid; c < VM_NCELL; c += T) {
        int gx = ox - 2 + c % VM_HALO_W, gy = oy - 2 + c / VM_HALO_W;
        bool in = gx >= 0 && gx < L.w && gy >= 0 && gy < L.h;
        int g = gy * L.rs + gx;
        S.mean[c] = in ? L.mean[g] : make_float2(0, 0);
        S.var[c] = in ? L.var[g] : make_float2(0, 0);
        S.tpsb[c] = in ? L.tps_b[g] : make_float2(0, 0);
        S.cross[c] = in ? L.cross[g] : 0.0f;
        S.value[c] = in ? L.value[g] : 0.0f;
    }
    if (tid == 0)
        S.n_act = 0;
    __syncthreads();

    bool improving = false;
    uint32_t st_cand = 0, st_commit = 0;

    for (int pi = 0; pi < 2; ++pi) {
        for (int pj = 0; pj < 2; ++pj) {
            // ---- 1. candidates of this phase (get_improve_mask_idx, morph.cu:621-646;
            // pixel_on_border :648-667) ----
            if (tid < 256) {
                const int tx = tid & 31, ty = tid >> 5;
                const int px = ox + tx * 2 + pj, py = oy + ty * 2 + pi;
                int state = 0;
                if (px < L.w && py < L.h) {
                    const int oxb = px % 5, oyb = py % 5;
                    const int mcx = px / 5 - bx0, mcy = py / 5 - by0;
                    const int begi = oyb >= 2 ? 1 : 0, begj = oxb >= 2 ? 1 : 0;
                    const uint32_t *ib = S.imp + (oyb * 5 + oxb) * 9;
                    bool hit = false;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            if (S.mask[mcy + begi + i - 1][mcx + begj + j - 1] &
                                ib[(begi + i) * 3 + begj + j])
                                hit = true;
                    if (hit) {
                        state = 2; // in the mask: its bit is cleared unless it commits
                        if (!pixel_locked(L, P.bcond, px, py))
                            S.list[atomicAdd(&S.n_act, 1)] = tid;
                    }
                }
                S.d_ok[tid] = state;
            }
            __syncthreads();
            const int n_act = S.n_act;

            if (n_act > 0) {
                st_cand += n_act;
                // ---- 2. optimize_pixel (morph.cu:1030-1083) on the pre-phase state,
                // L lanes per candidate ----
                int Lf = VM_MIN_FANOUT;
                while (Lf * 2 <= VM_MAX_FANOUT && Lf * 2 * n_act <= T)
                    Lf *= 2;
                const int slots = T / Lf;
                const int sub = tid & (Lf - 1), grp = tid / Lf;
                for (int base = 0; base < n_act; base += slots) {
                    const int li = base + grp;
                    if (li < n_act) {
                        const int slot = S.list[li];
                        const int tx = slot & 31, ty = slot >> 5;
                        PixelCtx c;
                        c.px = ox + tx * 2 + pj;
                        c.py = oy + ty * 2 + pi;
                        c.hc = (ty * 2 + pi) * VM_HALO_W + (tx * 2 + pj);
                        c.idx = c.py * L.rs + c.px;
                        c.v = L.v[c.idx];
                        c.old_luma = L.luma[c.idx];
                        c.ui_axy = L.ui_axy[c.idx];
                        c.ui_b = L.ui_b[c.idx];
                        c.tps_b = S.tpsb[c.hc + 2 * VM_HALO_W + 2];
                        c.tps_axy = S.tps[(border_class(c.py, L.h) * 5 + border_class(c.px, L.w)) * 25 + 12] / 2;
                        NbCache nb;
                        nb_load(nb, L, P, S, c, sub, Lf);
                        // compute_gradient, morph.cu:763-778
                        float gx = energy_change(L, P, S, nb, c, P.eps, 0, Lf) - energy_change(L, P, S, nb, c, -P.eps, 0, Lf);
                        float gy = energy_change(L, P, S, nb, c, 0, P.eps, Lf) - energy_change(L, P, S, nb, c, 0, -P.eps, Lf);
                        gx = -gx;
                        gy = -gy;
                        const float ng = fsqrt(gx * gx + gy * gy);
                        if (ng != 0) {
                            gx = fdiv(gx, ng);
                            gy = fdiv(gy, ng);
                            // prevent_foldover, morph.cu:872-883
                            float t_min = 10;
                            fover_ring(L, c.px, c.py, -1.0f, -c.v.x, -c.v.y, -gx, -gy, t_min);
                            fover_ring(L, c.px, c.py, 1.0f, c.v.x, c.v.y, gx, gy, t_min);
                            float cc = fmaxf(t_min - P.eps, 0.0f);
                            // golden_section_search, morph.cu:885-947
                            const float R = 0.618033989f, C = 1.0f - R;
                            float a = 0;
                            float b = a * R + cc * C, x = b * R + cc * C;
                            float fb = energy_change(L, P, S, nb, c, gx * b, gy * b, Lf);
                            float fx = energy_change(L, P, S, nb, c, gx * x, gy * x, Lf);
                            while (cc - a > P.eps) {
                                const bool lt = fx < fb;
                                if (lt) {
                                    a = b;
                                    b = x;
                                    x = b * R + cc * C;
                                } else {
                                    cc = x;
                                    x = b * R + a * C;
                                }
                                const float f = energy_change(L, P, S, nb, c, gx * x, gy * x, Lf);
                                if (lt) {
                                    fb = fx;
                                    fx = f;
                                } else {
                                    float t = b;
                                    b = x;
                                    x = t;
                                    fx = fb;
                                    fb = f;
                                }
                            }
                            const float tmin = fx < fb ? x : b, fmin = fx < fb ? fx : fb;
                            if (fmin < 0 && sub == 0) {
                                S.d_step[slot] = make_float2(gx * tmin, gy * tmin);
                                S.d_ok[slot] = 1;
                            }
                        }
                    }
                }
            }
            __syncthreads();

            // ---- 3. commit_pixel_motion (morph.cu:990-1026): own-pixel state, mask
            // bit, and the record the per-cell gather below reads ----
            bool ok = false;
            if (tid < 256) {
                const int state = S.d_ok[tid];
                if (state != 0) {
                    const int tx = tid & 31, ty = tid >> 5;
                    const int px = ox + tx * 2 + pj, py = oy + ty * 2 + pi;
                    const int mcx = px / 5 - bx0, mcy = py / 5 - by0;
                    const uint32_t bit = 1u << ((px % 5) + (py % 5) * 5);
                    if (state == 1) {
                        ok = true;
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
                    } else {
                        atomicAnd(&S.mask[mcy][mcx], ~bit);
                    }
                }
            }
            if (tid == 0)
                S.n_act = 0;
            const int ncommit = __syncthreads_count(ok);
            if (ncommit) {
                improving = true;
                st_commit += ncommit;
                // ssim_update (morph.cu:951-988) + the tps.b scatter (:1006-1015) as
                // a gather: every tile+halo cell adds the records of the committed
                // pixels whose 5x5 window contains it, in row-major order of those
                // pixels; then UpdateSSIM (:1258-1279)
                for (int cell = tid; cell < VM_NCELL; cell += T) {
                    const int ry = cell / VM_HALO_W - 2, rx = cell % VM_HALO_W - 2; // tile-relative
                    const int qx = ox + rx, qy = oy + ry;
                    if (qx < 0 || qx >= L.w || qy < 0 || qy >= L.h)
                        continue;
                    float2 m = S.mean[cell], q = S.var[cell], tb = S.tpsb[cell];
                    float cr = S.cross[cell];
                    int ylo = max(ry - 2, 0), yhi = min(ry + 2, VM_TILE_H - 1);
                    int xlo = max(rx - 2, 0), xhi = min(rx + 2, VM_TILE_W - 1);
                    ylo += (ylo & 1) ^ pi;
                    xlo += (xlo & 1) ^ pj;
                    bool touched = false;
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
                    if (touched) {
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
        }
    }

    // ---- SaveSSIM (morph.cu:1236-1256), tps.b and the owned mask words ----
    if (improving) {
        for (int c = tid; c < VM_NCELL; c += T) {
            int gx = ox - 2 + c % VM_HALO_W, gy = oy - 2 + c / VM_HALO_W;
            if (gx < 0 || gx >= L.w || gy < 0 || gy >= L.h)
                continue;
            int g = gy * L.rs + gx;
            L.mean[g] = S.mean[c];
            L.var[g] = S.var[c];
            L.tps_b[g] = S.tpsb[c];
            L.cross[g] = S.cross[c];
            L.value[g] = S.value[c];
        }
    }
    if (tid < nbx * nby) {
        int mx = tid % nbx, my = tid / nbx;
        // words owned by this tile: blocks that contain one of its pixels
        if (mx >= 1 && mx <= nbx - 2 && my >= 1 && my <= nby - 2)
            L.impmask[(by0 + my + 1) * L.imp_rs + (bx0 + mx + 1)] = S.mask[my][mx];
    }
    if (tid == 0) {
        if (improving)
            atomicOr(&flags[iter_idx], 1u);
        // per-iteration activity counters: active tiles, candidate visits, commits
        atomicAdd(&stats[iter_idx * 4 + 0], 1u);
        atomicAdd(&stats[iter_idx * 4 + 1], st_cand);
        atomicAdd(&stats[iter_idx * 4 + 2], st_commit);
    }
}

} // namespace

// ---------------------------------------------------------------------------
// launchers

void SUF(vm_launch_init_level)(const VmLevelView &L, float ssim_clamp, const uint32_t *tables,
                               hipStream_t s)
{
    dim3 b(32, 8), g((L.w + 31) / 32, (L.h + 7) / 8);
    hipLaunchKernelGGL(SUF(k_init_level), g, b, 0, s, L, ssim_clamp, tables);
}

void SUF(vm_launch_optimize)(const VmLevelView &L, const VmKParams &P, const uint32_t *tables,
                             int offx, int offy, uint32_t *flags, uint32_t *stats, int iter_idx,
                             int fixed_work, int threads, hipStream_t s)
{
    dim3 b(threads), g((L.w + VM_PITCH_X - 1) / VM_PITCH_X, (L.h + VM_PITCH_Y - 1) / VM_PITCH_Y);
    hipLaunchKernelGGL(SUF(k_optimize), g, b, 0, s, L, P, tables, offx, offy, flags, stats,
                       iter_idx, fixed_work);
}

void SUF(vm_launch_upsample)(float2 *dst, int dw, int dh, int drs, const float2 *src, int sw,
                             int sh, int srs, hipStream_t s)
{
    dim3 b(64, 4), g((dw + 63) / 64, (dh + 3) / 4);
    hipLaunchKernelGGL(SUF(k_upsample), g, b, 0, s, dst, dw, dh, drs, src, sw, sh, srs);
}

void SUF(vm_launch_splat)(const VmLevelView &L, int w0, int h0, const vm_constraint *dev_c, int n,
                          hipStream_t s)
{
    hipLaunchKernelGGL(SUF(k_splat), dim3(1), dim3(64), 0, s, L, w0, h0, dev_c, n);
}
