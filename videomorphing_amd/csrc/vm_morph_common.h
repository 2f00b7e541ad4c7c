// vm_morph_common.h -- device helpers shared by the optimizer kernels
// (vm_morph_kernels.hip, vm_sweep_kernels.hip).  Included once per arithmetic
// mode: VM_EXACT selects IEEE division / square root and the literal SSIM
// formula, otherwise v_rcp_f32 / v_sqrt_f32 and the one-rcp-one-sqrt form.
#ifndef VM_MORPH_COMMON_H
#define VM_MORPH_COMMON_H

#include "vm_internal.h"

#ifndef VM_EXACT
#error "define VM_EXACT to 0 (FAST), 1 (EXACT) or 2-5 (diagnostic builds of the EXACT source)"
#endif

#if VM_EXACT == 4
// VM_MATH_REF_TEX8: the EXACT source, IEEE arithmetic, with the texture fetches quantised the way the
// reference's tex2D(linear) fetches are on CUDA hardware (tex8_weight below)
#define SUF(name) name##_tex8
#define VM_TEX8 1
#elif VM_EXACT == 5
// VM_MATH_REF_TEX8_TRUNC: the same with the weights truncated instead of rounded (sensitivity to the rule)
#define SUF(name) name##_tex8t
#define VM_TEX8 2
#elif VM_EXACT == 3
// VM_MATH_REF_FASTMATH (sweeps only): the EXACT source -- the reference's own expressions in its own order --
// with what its project file compiles it with, --use_fast_math (MdiEditor.vcxproj:208-213): fused
// multiply-adds, approximate division (x * rcp(y), CUDA: __fdividef) and approximate square root
#define SUF(name) name##_reffm
#elif VM_EXACT == 2
// the EXACT source compiled with -ffp-contract=fast (VM_MATH_EXACT_FMA, sweeps only): fused multiply-adds
// wherever the compiler contracts, IEEE division and square root
#define SUF(name) name##_exactf
#elif VM_EXACT
#define SUF(name) name##_exact
#else
#define SUF(name) name##_fast
#endif

namespace {


__device__ __forceinline__ float fdiv(float a, float b)
{
#if VM_EXACT && VM_EXACT != 3
    return a / b;
#else
    return a * __builtin_amdgcn_rcpf(b);
#endif
}

__device__ __forceinline__ float fsqrt(float a)
{
#if VM_EXACT && VM_EXACT != 3
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
// FAST form of ssim(): a, b = window means (sum / n); sx2, sy2, sxy = raw second
// moment sums; n = window count.  With V = sum - n a a (= n var) and C = n cov,
//   c*s = (2 sqrt(Vx Vy) + n c2)(|C| + n c3) / ((Vx + Vy + n c2)(sqrt(Vx Vy) + n c3))
// -- the 1/n factors of the four terms cancel, one v_rcp_f32 and one v_sqrt_f32 per
// evaluation.  The variances are formed as (sum - n a a) like the reference does:
// forming them from pre-divided sums (E[x^2] - a^2) was measured to quantise the
// line search's tiny energy differences enough to cost 6 % of SSIM energy after 86
// sweeps (profiles/r01_notes.md).  Every SSIM value of FAST mode -- the stored ones
// and the trial ones of the line search -- comes from this one function, so that a
// zero step changes the energy by exactly zero.  For interior pixels n is the
// compile-time constant 25 and n c2, n c3 fold.
__device__ __forceinline__ float ssim_core(float a, float b, float sx2, float sy2, float sxy,
                                           float n, float clamp)
{
    const float nc2 = n * 58.5225f, nc3 = n * 29.26125f;
    const float na = n * a, nb = n * b;
    const float vx = fmaxf(fmaf(-na, a, sx2), 0.0f);
    const float vy = fmaxf(fmaf(-nb, b, sy2), 0.0f);
    const float cov = fmaf(-na, b, sxy);
    const float ss = __builtin_amdgcn_sqrtf(vx * vy);
    const float num = fmaf(2.0f, ss, nc2) * (fabsf(cov) + nc3);
    const float den = ((vx + vy) + nc2) * (ss + nc3);
    const float val = num * __builtin_amdgcn_rcpf(den);
    return fmaxf(fminf(val, 1.0f), clamp);
}

__device__ __forceinline__ float ssim_value(float mx, float my, float vx, float vy, float cross,
                                            float counter, float clamp)
{
    if (counter <= 1)
        return 0;
    const float in = counter == 25.0f ? 0.04f : __builtin_amdgcn_rcpf(counter);
    return ssim_core(mx * in, my * in, vx, vy, cross, counter, clamp);
}
#endif

#ifdef VM_TEX8
// CUDA's linear filter (the reference samples both images and the coarser level's field through it:
// morph.cu:29-30, 316-322; taps at :212-213, 680-681, 960-961; imgop_upsample.cu:17-31) evaluates
//   (1-a)(1-b) T[i,j] + a(1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1],  a = frac(x - 0.5), b = frac(y - 0.5)
// with a and b "stored in 9-bit fixed point format with 8 bits of fractional value (so 1.0 is exactly
// represented)" (CUDA C Programming Guide, appendix Texture Fetching, Linear Filtering).  The guide does not
// say how the fraction is brought to 8 bits.  Rule taken here (VM_TEX8 == 1): round to nearest,
// a_q = floor(256 a + 0.5) / 256 -- the only reading under which a_q can reach the 1.0 the format is said
// to represent exactly (a truncated fraction never exceeds 255/256); a_q = 1 gives T[i+1] exactly, i.e. what
// rounding the COORDINATE to 1/256 would give.  VM_TEX8 == 2 truncates instead (sensitivity check).  The
// four products and three sums stay f32 in the order tap() has them: the hardware's internal order and
// width are not documented either, and with 8-bit weights the weight products are exact.
__device__ __forceinline__ float tex8_weight(float a)
{
#if VM_TEX8 == 1
    return floorf(a * 256.0f + 0.5f) * 0.00390625f;
#else
    return floorf(a * 256.0f) * 0.00390625f;
#endif
}
#else
__device__ __forceinline__ float tex8_weight(float a) { return a; }
#endif

// tex2D(linear, clamp, unnormalised) on a pitched f32 image: texel centres at
// i+0.5 (morph.cu:316-322); exact float weights (the TEX8 diagnostic builds: 8-bit ones, above)
__device__ __forceinline__ float tap(const float *__restrict__ img, int w, int h, int rs, float x,
                                     float y)
{
    float xb = x - 0.5f, yb = y - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = tex8_weight(xb - fi), b = tex8_weight(yb - fj);
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
    float a = tex8_weight(xb - fi), b = tex8_weight(yb - fj);
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

} // namespace

#endif
