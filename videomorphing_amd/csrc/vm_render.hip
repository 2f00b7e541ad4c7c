// vm_render.hip -- compositor and result-delivery kernels for gfx950.
//
// k_render:  kernel_render_halfway_image, Algorithm/render.cu:16-60.  The
//   reference samples float4 textures that RenderStage2 re-uploads every frame
//   (UI/RenderWidget.cpp:229-266); here the Poisson-extended canvases stay
//   resident as RGBA8 (uchar -> float is exact, so the taps see the same
//   values) and v/u as pitched float2: 8 B + 8 B + 3 B of compulsory traffic
//   per pixel plus gathered canvas taps served by L2.  k_render is the plain form
//   (VM_RENDER=plain, and fields of 4 GiB and more); the product path is
//   k_render_win: the same arithmetic with the taps of v / u served from an LDS
//   window (below).
// k_upscale: CMatchingThread::update_result + Resize,
//   Algorithm/MatchingThread.cpp:22-136.
#include "vm_internal.h"
#include <cstdlib>
#include <cstring>

namespace {

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

__device__ __forceinline__ float3 tap_rgb(const uchar4 *__restrict__ img, int w, int h, float x,
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
    uchar4 t00 = img[(size_t)j0 * w + i0], t10 = img[(size_t)j0 * w + i1];
    uchar4 t01 = img[(size_t)j1 * w + i0], t11 = img[(size_t)j1 * w + i1];
    float w00 = (1 - a) * (1 - b), w10 = a * (1 - b), w01 = (1 - a) * b, w11 = a * b;
    float3 r;
    r.x = w00 * (float)t00.x + w10 * (float)t10.x + w01 * (float)t01.x + w11 * (float)t11.x;
    r.y = w00 * (float)t00.y + w10 * (float)t10.y + w01 * (float)t01.y + w11 * (float)t11.y;
    r.z = w00 * (float)t00.z + w10 * (float)t10.z + w01 * (float)t01.z + w11 * (float)t11.z;
    return r;
}

__global__ __launch_bounds__(256) void k_render(uint8_t *__restrict__ out, int out_pitch, int w,
                                                int h, int rs, int ex, float color_fa,
                                                float geo_fa, int color_from,
                                                const uchar4 *__restrict__ ext0,
                                                const uchar4 *__restrict__ ext1,
                                                const float2 *__restrict__ vf,
                                                const float2 *__restrict__ uf)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h)
        return;
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    const float alpha = 0.8f;
    const float s1 = 2 * geo_fa - 1;
    const float s2 = 4 * geo_fa - 4 * geo_fa * geo_fa;
    const float qx = (float)x, qy = (float)y;
    float px = qx, py = qy;
    float2 v = tap2(vf, w, h, rs, px + 0.5f, py + 0.5f);
    float2 u = uf ? tap2(uf, w, h, rs, px + 0.5f, py + 0.5f) : make_float2(0.0f, 0.0f);
    for (int i = 0; i < 20; ++i) {
        px = qx - s1 * v.x - s2 * u.x;
        py = qy - s1 * v.y - s2 * u.y;
        float2 t = tap2(vf, w, h, rs, px + 0.5f, py + 0.5f);
        v.x = alpha * t.x + (1 - alpha) * v.x;
        v.y = alpha * t.y + (1 - alpha) * v.y;
        if (uf) {
            t = tap2(uf, w, h, rs, px + 0.5f, py + 0.5f);
            u.x = alpha * t.x + (1 - alpha) * u.x;
            u.y = alpha * t.y + (1 - alpha) * u.y;
        } else {
            // a zero path stays zero: alpha*0 + (1-alpha)*0
        }
    }
    float3 c0 = tap_rgb(ext0, cw, ch, px - v.x + ex + 0.5f, py - v.y + ex + 0.5f);
    float3 c1 = tap_rgb(ext1, cw, ch, px + v.x + ex + 0.5f, py + v.y + ex + 0.5f);
    double r, g, b;
    if (color_from == 0) {
        r = c0.x + 0.5; g = c0.y + 0.5; b = c0.z + 0.5;
    } else if (color_from == 1) {
        r = c0.x * (1 - color_fa) + c1.x * color_fa + 0.5;
        g = c0.y * (1 - color_fa) + c1.y * color_fa + 0.5;
        b = c0.z * (1 - color_fa) + c1.z * color_fa + 0.5;
    } else {
        r = c1.x + 0.5; g = c1.y + 0.5; b = c1.z + 0.5;
    }
    uint8_t *o = out + (size_t)y * out_pitch + 3 * x;
    o[0] = (uint8_t)r; // make_uchar3: truncation (render.cu:49-56)
    o[1] = (uint8_t)g;
    o[2] = (uint8_t)b;
}

// ---------------------------------------------------------------------------
// Lean index arithmetic of a tap (the float results are tap2's expressions in tap2's order):
//   * the clamp of floor() to [-1, w] is one v_med3_f32, the clamps of i0 / i0 + 1 to [0, w - 1] one v_med3_i32 each
//     (i0 is already within [-1, w]: the max(-1, .) the compiler keeps is redundant);
//   * row offsets by 24-bit multiplies (full rate), texel addresses as 32-bit byte offsets from a scalar base
//     (no sign extension, no 64-bit address arithmetic per texel): the launcher takes the kernel below only when
//     the field is smaller than 4 GiB.
__device__ __forceinline__ int med3_i32(int a, int b, int c)     // median = clamp of a to [b, c] when b <= c
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#ifndef VM_RENDER_ITERS
#define VM_RENDER_ITERS 20      // render.cu:29 (anything else: a timing experiment)
#endif

struct TapIdx {
    uint32_t o00, o10, o01, o11;    // byte offsets of the four texels
    float a, b;
};

__device__ __forceinline__ TapIdx tap_index(float x, float y, float fw, float fh, int wm1, int hm1, uint32_t rs)
{
    TapIdx t;
    const float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    t.a = xb - fi;
    t.b = yb - fj;
    fi = __builtin_amdgcn_fmed3f(fi, -1.0f, fw);    // = fminf(fmaxf(fi, -1), w), NaN -> -1 like there
    fj = __builtin_amdgcn_fmed3f(fj, -1.0f, fh);
    const int i = (int)fi, j = (int)fj;
    const uint32_t i0 = (uint32_t)med3_i32(i, 0, wm1), i1 = (uint32_t)med3_i32(i + 1, 0, wm1);
    const uint32_t r0 = __umul24((uint32_t)med3_i32(j, 0, hm1), rs), r1 = __umul24((uint32_t)med3_i32(j + 1, 0, hm1), rs);
    t.o00 = (r0 + i0) << 3; t.o10 = (r0 + i1) << 3;
    t.o01 = (r1 + i0) << 3; t.o11 = (r1 + i1) << 3;
    return t;
}

// ---------------------------------------------------------------------------
// k_render_win: k_render with the 21 dependent taps of v (and u) served from LDS and lean index arithmetic.  Measured
// (round 5, profiles/r05_notes.md): k_render's loop costs 4.0 us per iteration and 1080p frame at TWO limits at once --
// the texture path takes ~19 cycles per 64-lane 8-byte gather and CU (4 gathers per tap), and its ~56 VALU
// instructions per tap (two of them quarter-rate multiplies, 64-bit address arithmetic per texel) cost the same in
// issue slots: an LDS window alone (112 us) or lean indices alone (97 us) leave the 98 us where they were, both
// together give 64 us.  A workgroup of RW x RH = 32 x 16 pixels stages one window of the field around where its pixels land --
// the block displaced by the warp of its centre, RR cells of margin for the variation of the warp across the block
// and the path of the fixed-point iteration -- with CLAMPED source coordinates, so that window cell (i - ox, j - oy)
// holds exactly the texel tap2 fetches for the raw floor index i (tap2's clamps of i0, i0 + 1 to the image commute
// with the staging), and a tap inside the window needs no clamp at all: floor, convert, two unsigned compares, one
// multiply-add, four ds_read_b64.  A tap outside takes global gathers with the lean indices above.  Same float
// expressions in the same order: byte-identical output (tests/test_gpu_parity.py::test_render_*).  Tiles are dealt to
// the XCDs in contiguous bands (a workgroup's id modulo 8 is its XCD).
#ifndef VM_RENDER_RR
#define VM_RENDER_RR 10
#endif
// tap_rgb with the lean indices (the canvas is far below 4 GiB whenever the field is)
__device__ __forceinline__ float3 tap_rgb_lean(const uchar4 *__restrict__ img, float fw, float fh, int wm1, int hm1, uint32_t w, float x,
                                               float y)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    const float a = xb - fi, b = yb - fj;
    fi = __builtin_amdgcn_fmed3f(fi, -1.0f, fw);
    fj = __builtin_amdgcn_fmed3f(fj, -1.0f, fh);
    const int i = (int)fi, j = (int)fj;
    const uint32_t i0 = (uint32_t)med3_i32(i, 0, wm1), i1 = (uint32_t)med3_i32(i + 1, 0, wm1);
    const uint32_t r0 = __umul24((uint32_t)med3_i32(j, 0, hm1), w), r1 = __umul24((uint32_t)med3_i32(j + 1, 0, hm1), w);
    const char *base = (const char *)img;
    const uchar4 t00 = *(const uchar4 *)(base + ((r0 + i0) << 2)), t10 = *(const uchar4 *)(base + ((r0 + i1) << 2));
    const uchar4 t01 = *(const uchar4 *)(base + ((r1 + i0) << 2)), t11 = *(const uchar4 *)(base + ((r1 + i1) << 2));
    const float w00 = (1 - a) * (1 - b), w10 = a * (1 - b), w01 = (1 - a) * b, w11 = a * b;
    float3 r;
    r.x = w00 * (float)t00.x + w10 * (float)t10.x + w01 * (float)t01.x + w11 * (float)t11.x;
    r.y = w00 * (float)t00.y + w10 * (float)t10.y + w01 * (float)t01.y + w11 * (float)t11.y;
    r.z = w00 * (float)t00.z + w10 * (float)t10.z + w01 * (float)t01.z + w11 * (float)t11.z;
    return r;
}

#ifndef VM_RENDER_RH
#define VM_RENDER_RH 16     // 32 x 16 pixels per workgroup: 3.8 staged cells per pixel (8 rows: 6.0; 63.0 -> 60.7 us per frame)
#endif
constexpr int RW = 32, RH = VM_RENDER_RH, RR = VM_RENDER_RR, WW = RW + 2 * RR + 1, WH = RH + 2 * RR + 1;

typedef const volatile __attribute__((address_space(3))) unsigned long long *LdsWords;

__device__ __forceinline__ float2 lds8(LdsWords win, uint32_t c)
{
    const unsigned long long q = win[c];
    return make_float2(__uint_as_float((uint32_t)q), __uint_as_float((uint32_t)(q >> 32)));
}

template <bool HAS_U>
__global__ __launch_bounds__(RW * RH) void k_render_win(uint8_t *__restrict__ out, int out_pitch, int w, int h, int rs, int ex,
                                                    float color_fa, float geo_fa, int color_from,
                                                    const uchar4 *__restrict__ ext0, const uchar4 *__restrict__ ext1,
                                                    const float2 *__restrict__ vf, const float2 *__restrict__ uf, int tiles_x,
                                                    int ntiles, bool lean_canvas)
{
    __shared__ float2 win_v[WH * WW];
    __shared__ float2 win_u[HAS_U ? WH * WW : 1];
    const int blk = blockIdx.x, per = (ntiles + 7) / 8;
    const int tile = (blk % 8) * per + blk / 8;
    if (tile >= ntiles)
        return;                                 // the whole workgroup
    const int bx = (tile % tiles_x) * RW, by = (tile / tiles_x) * RH;
    const int tid = threadIdx.y * RW + threadIdx.x;
    const float fw = (float)w, fh = (float)h;
    const int wm1 = w - 1, hm1 = h - 1;
    const float alpha = 0.8f;
    const float s1 = 2 * geo_fa - 1;
    const float s2 = 4 * geo_fa - 4 * geo_fa * geo_fa;
    int ox, oy;
    {
        const int cx = min(bx + RW / 2, wm1), cy = min(by + RH / 2, hm1);
        const float2 vc = vf[cy * rs + cx];
        const float2 uc = HAS_U ? uf[cy * rs + cx] : make_float2(0.0f, 0.0f);
        // (a non-finite or absurd centre puts the window nowhere useful: every tap then takes the global path)
        const float dx = __builtin_amdgcn_fmed3f(s1 * vc.x + s2 * uc.x, -1e6f, 1e6f), dy = __builtin_amdgcn_fmed3f(s1 * vc.y + s2 * uc.y, -1e6f, 1e6f);
        ox = bx - (int)rintf(dx) - RR;
        oy = by - (int)rintf(dy) - RR;
    }
    for (int i = tid; i < WH * WW; i += RW * RH) {
        const int wy = i / WW, wx = i - wy * WW;
        const int src = min(max(oy + wy, 0), hm1) * rs + min(max(ox + wx, 0), wm1);
        win_v[i] = vf[src];
        if (HAS_U)
            win_u[i] = uf[src];
    }
    __syncthreads();
    const int x = bx + threadIdx.x, y = by + threadIdx.y;
    if (x >= w || y >= h)
        return;
    const int cw = w + 2 * ex, ch = h + 2 * ex;
    const float qx = (float)x, qy = (float)y;
    float px = qx, py = qy;
    float2 v, u = make_float2(0.0f, 0.0f);
    const LdsWords wv = (LdsWords)win_v, wu = (LdsWords)win_u;
    // one tap of v (and u) at (px + 0.5, py + 0.5): tap2's expression
    auto tap = [&](float2 &tv, float2 &tu) {
        const float xb = (px + 0.5f) - 0.5f, yb = (py + 0.5f) - 0.5f;
        const float fi = floorf(xb), fj = floorf(yb);
        const float a = xb - fi, b = yb - fj;
        const uint32_t a0 = (uint32_t)(int)fi - (uint32_t)ox, b0 = (uint32_t)(int)fj - (uint32_t)oy;
        // the window is read unconditionally (cell 0 for a tap outside), the global gathers only by the lanes outside:
        // written as "if (inside) LDS else global" the compiler merges the two into flat loads through selected pointers
        const bool inside = a0 < (uint32_t)(WW - 1) && b0 < (uint32_t)(WH - 1);
        const uint32_t c = inside ? __umul24(b0, (uint32_t)WW) + a0 : 0u;     // (24-bit multiply-add: full rate)
        // (and as volatile 8-byte words: plain loads are sunk below the branch and merged all the same)
        float2 t00 = lds8(wv, c), t10 = lds8(wv, c + 1), t01 = lds8(wv, c + WW), t11 = lds8(wv, c + WW + 1);
        float2 u00, u10, u01, u11;
        if (HAS_U) { u00 = lds8(wu, c); u10 = lds8(wu, c + 1); u01 = lds8(wu, c + WW); u11 = lds8(wu, c + WW + 1); }
        if (!inside) {
            const TapIdx t = tap_index(px + 0.5f, py + 0.5f, fw, fh, wm1, hm1, (uint32_t)rs);
            const char *bv = (const char *)vf, *bu = (const char *)uf;
            t00 = *(const float2 *)(bv + t.o00); t10 = *(const float2 *)(bv + t.o10);
            t01 = *(const float2 *)(bv + t.o01); t11 = *(const float2 *)(bv + t.o11);
            if (HAS_U) {
                u00 = *(const float2 *)(bu + t.o00); u10 = *(const float2 *)(bu + t.o10);
                u01 = *(const float2 *)(bu + t.o01); u11 = *(const float2 *)(bu + t.o11);
            }
        }
        tv.x = (1 - a) * (1 - b) * t00.x + a * (1 - b) * t10.x + (1 - a) * b * t01.x + a * b * t11.x;
        tv.y = (1 - a) * (1 - b) * t00.y + a * (1 - b) * t10.y + (1 - a) * b * t01.y + a * b * t11.y;
        if (HAS_U) {
            tu.x = (1 - a) * (1 - b) * u00.x + a * (1 - b) * u10.x + (1 - a) * b * u01.x + a * b * u11.x;
            tu.y = (1 - a) * (1 - b) * u00.y + a * (1 - b) * u10.y + (1 - a) * b * u01.y + a * b * u11.y;
        }
    };
    {
        float2 tv, tu;
        tap(tv, tu);
        v = tv;
        if (HAS_U) u = tu;
    }
    for (int i = 0; i < VM_RENDER_ITERS; ++i) {
        // (without a path u stays +0 and s2 * u is a loop invariant -- +-0, or NaN for a non-finite geo_fa: still
        // subtracted, so that the result is k_render's in every case)
        px = qx - s1 * v.x - s2 * u.x;
        py = qy - s1 * v.y - s2 * u.y;
        float2 tv, tu;
        tap(tv, tu);
        v.x = alpha * tv.x + (1 - alpha) * v.x;
        v.y = alpha * tv.y + (1 - alpha) * v.y;
        if (HAS_U) {
            u.x = alpha * tu.x + (1 - alpha) * u.x;
            u.y = alpha * tu.y + (1 - alpha) * u.y;
        }
    }
    float3 c0, c1;
    if (lean_canvas) {
        c0 = tap_rgb_lean(ext0, (float)cw, (float)ch, cw - 1, ch - 1, (uint32_t)cw, px - v.x + ex + 0.5f, py - v.y + ex + 0.5f);
        c1 = tap_rgb_lean(ext1, (float)cw, (float)ch, cw - 1, ch - 1, (uint32_t)cw, px + v.x + ex + 0.5f, py + v.y + ex + 0.5f);
    } else {
        c0 = tap_rgb(ext0, cw, ch, px - v.x + ex + 0.5f, py - v.y + ex + 0.5f);
        c1 = tap_rgb(ext1, cw, ch, px + v.x + ex + 0.5f, py + v.y + ex + 0.5f);
    }
    double r, g, b;
    if (color_from == 0) {
        r = c0.x + 0.5; g = c0.y + 0.5; b = c0.z + 0.5;
    } else if (color_from == 1) {
        r = c0.x * (1 - color_fa) + c1.x * color_fa + 0.5;
        g = c0.y * (1 - color_fa) + c1.y * color_fa + 0.5;
        b = c0.z * (1 - color_fa) + c1.z * color_fa + 0.5;
    } else {
        r = c1.x + 0.5; g = c1.y + 0.5; b = c1.z + 0.5;
    }
    uint8_t *o = out + (size_t)y * out_pitch + 3 * x;
    o[0] = (uint8_t)r; // make_uchar3: truncation (render.cu:49-56)
    o[1] = (uint8_t)g;
    o[2] = (uint8_t)b;
}

// BiLinear of MatchingThread.cpp:103-136 on the level's v scaled by (rx, ry)
__global__ __launch_bounds__(256) void k_upscale(float2 *__restrict__ dst, int w0, int h0,
                                                 int dpitch, const float2 *__restrict__ v, int w,
                                                 int h, int rs)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w0 || y >= h0)
        return;
    const float rx = (float)w0 / (float)w, ry = (float)h0 / (float)h;
    if (w == w0 && h == h0) {
        float2 s = v[y * rs + x];
        dst[(size_t)y * dpitch + x] = s; // ratio 1: copied unscaled (MatchingThread.cpp:42,55-58)
        return;
    }
    const float fy = (float)((y + 0.5) / h0 * h - 0.5);
    const float fx = (float)((x + 0.5) / w0 * w - 0.5);
    int xi[2] = {(int)floorf(fx), (int)ceilf(fx)};
    int yi[2] = {(int)floorf(fy), (int)ceilf(fy)};
    const float uu = fx - xi[0], vv = fy - yi[0];
    float2 val[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int tx = min(max(xi[i], 0), w - 1), ty = min(max(yi[j], 0), h - 1);
            float2 s = v[ty * rs + tx];
            val[i][j] = make_float2(s.x * rx, s.y * ry);
        }
    float2 r;
    r.x = val[0][0].x * (1 - uu) * (1 - vv) + val[0][1].x * (1 - uu) * vv +
          val[1][0].x * uu * (1 - vv) + val[1][1].x * uu * vv;
    r.y = val[0][0].y * (1 - uu) * (1 - vv) + val[0][1].y * (1 - uu) * vv +
          val[1][0].y * uu * (1 - vv) + val[1][1].y * uu * vv;
    dst[(size_t)y * dpitch + x] = r;
}

// the frames the temporal pyramid skipped: _vector[beg] * (1 - fa) + _vector[end] * fa
// (MatchingThread.cpp:62-78; a cv::Mat expression = addWeighted in float: two products, one sum)
__global__ __launch_bounds__(256) void k_blend_v(float2 *__restrict__ dst, int dpitch, const float2 *__restrict__ a,
                                                 const float2 *__restrict__ b, int spitch, int w0, int h0, float alpha,
                                                 float beta)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w0 || y >= h0)
        return;
    const float2 p = a[(size_t)y * spitch + x], q = b[(size_t)y * spitch + x];
    float2 r;
    r.x = p.x * alpha + q.x * beta;
    r.y = p.y * alpha + q.y * beta;
    dst[(size_t)y * dpitch + x] = r;
}

} // namespace

void vm_launch_blend_v(float2 *dst, int dpitch, const float2 *a, const float2 *b, int spitch, int w0, int h0,
                       float alpha, float beta, hipStream_t s)
{
    dim3 blk(64, 4), g((w0 + 63) / 64, (h0 + 3) / 4);
    hipLaunchKernelGGL(k_blend_v, g, blk, 0, s, dst, dpitch, a, b, spitch, w0, h0, alpha, beta);
}

void vm_launch_upscale(float2 *dst, int w0, int h0, int dpitch, const float2 *v, int w, int h,
                       int rs, hipStream_t s)
{
    dim3 b(64, 4), g((w0 + 63) / 64, (h0 + 3) / 4);
    hipLaunchKernelGGL(k_upscale, g, b, 0, s, dst, w0, h0, dpitch, v, w, h, rs);
}

// a kernel that only takes time: one wave reading the constant 100 MHz counter until `ticks` have passed
__global__ __launch_bounds__(64) void k_spin(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks)
        __builtin_amdgcn_s_sleep(8);
}

void vm_launch_spin(unsigned long long ticks, int workgroups, hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(workgroups), dim3(64), 0, s, ticks); }

void vm_launch_render(uint8_t *out, int out_pitch, int w, int h, int rs, int ex, float color_fa,
                      float geo_fa, int color_from, const uchar4 *ext0, const uchar4 *ext1,
                      const float2 *v, const float2 *u, hipStream_t s)
{
    static const char *mode = getenv("VM_RENDER");
    static const bool plain = mode && !strcmp(mode, "plain");      // the first kernel, for A/B runs
    // the window kernel addresses texels by 32-bit byte offsets and multiplies rows in 24 bits
    const bool small = (uint64_t)rs * (uint64_t)h * 8ull < (1ull << 32) && rs < (1 << 24) && h < (1 << 24);
    if (plain || !small) {
        dim3 b(64, 4), g((w + 63) / 64, (h + 3) / 4);
        hipLaunchKernelGGL(k_render, g, b, 0, s, out, out_pitch, w, h, rs, ex, color_fa, geo_fa, color_from, ext0, ext1, v, u);
        return;
    }
    const int tiles_x = (w + RW - 1) / RW, ntiles = tiles_x * ((h + RH - 1) / RH);
    const uint64_t cw = (uint64_t)w + 2 * (uint64_t)ex, ch = (uint64_t)h + 2 * (uint64_t)ex;
    const bool lean_canvas = cw * ch * 4ull < (1ull << 32) && cw < (1u << 24) && ch < (1u << 24);   // 32-bit texel offsets, 24-bit rows
    dim3 b(RW, RH), g(((ntiles + 7) / 8) * 8);
    if (u)
        hipLaunchKernelGGL(k_render_win<true>, g, b, 0, s, out, out_pitch, w, h, rs, ex, color_fa, geo_fa, color_from, ext0,
                           ext1, v, u, tiles_x, ntiles, lean_canvas);
    else
        hipLaunchKernelGGL(k_render_win<false>, g, b, 0, s, out, out_pitch, w, h, rs, ex, color_fa, geo_fa, color_from, ext0,
                           ext1, v, u, tiles_x, ntiles, lean_canvas);
}
