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
#include "vm_morph_common.h"

namespace {

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

} // namespace

// ---------------------------------------------------------------------------
// launchers

void SUF(vm_launch_init_level)(const VmLevelView &L, float ssim_clamp, const uint32_t *tables,
                               hipStream_t s)
{
    dim3 b(32, 8), g((L.w + 31) / 32, (L.h + 7) / 8);
    hipLaunchKernelGGL(SUF(k_init_level), g, b, 0, s, L, ssim_clamp, tables);
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
