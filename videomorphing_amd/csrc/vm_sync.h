// vm_sync.h -- device-side structures and launchers of vm_sync.hip: the synchronisation stage
// (CSyncThread, Algorithm/SyncThread.cpp; render_resample_image, Algorithm/render.cu:99-246)
#ifndef VM_SYNC_H
#define VM_SYNC_H
#include <hip/hip_runtime.h>
#include <stdint.h>

// summation bricks of the dot products (the fixed order)
#define VM_SB_X 32
#define VM_SB_Y 8
#define VM_SB_Z 8
#define VM_SYNC_SC_WORDS 16 // r0[3], r1[2][3], dot[3]
#define VM_SYNC_TICKET_WORDS(blocks) (3 * (1 + ((blocks) + 31) / 32) * 32)

struct VmSyncGrid {
    int w, h, d;
    int nbx, nby, nbz, nb; // bricks
    int per_xcd;           // ceil(nb / 8): the launch has 8 * per_xcd workgroups
};

// CG state of one level, all tight (d, h, w) float arrays
struct VmSyncSys {
    float *x[3];       // the solution d_x, d_y, d_z (SyncThread.h:33)
    float *r[3];       // residuals, start as the right-hand sides
    float *p[2][3];    // search directions, ping-pong (a brick reads its neighbours' OLD p)
    float *om[3];      // A p
    float *diag;       // the diagonal of A: UI term first, then the stencil's increments in order
    const float *tab;  // [5][5][5][25] off-diagonal entries per border-state triple
    double *part;      // [9][nb] brick partial sums: p.omega [3], r.r [2 parities][3]
    float *sc;         // VM_SYNC_SC_WORDS scalars
    unsigned *ticket;  // arrival counters: per component 1 + ceil(workgroups / 32), each on its own line
};

void vm_sync_launch_diag(const VmSyncGrid &g, float *diag, float w_tps, hipStream_t s);
void vm_sync_launch_scatter(float *dst, const int *idx, const float *val, int n, hipStream_t s);
void vm_sync_launch_rr(const VmSyncGrid &g, const VmSyncSys &S, hipStream_t s);
void vm_sync_launch_iteration(const VmSyncGrid &g, const VmSyncSys &S, int k, hipStream_t s);
void vm_sync_launch_finish(const VmSyncGrid &g, const VmSyncSys &S, int k, hipStream_t s);
void vm_sync_launch_upsample(float *dst, int dw, int dh, const float *src, int sw, int sh, float ratio, int pages,
                             hipStream_t s);
void vm_sync_launch_result(const float *X, const float *Y, const float *Z, int w, int h, int w0, int h0, float4 *out,
                           hipStream_t s);
void vm_sync_launch_render(uint8_t *out, int out_pitch, int w, int h, int d, float fa, int frame, const float4 *vec,
                           const uchar4 *video0, const uchar4 *video1, const float2 *forw0, const float2 *forw1,
                           hipStream_t s);
#endif
