// vm_host.h -- host-side objects behind the opaque handles of include/vmorph.h.
#ifndef VM_HOST_H
#define VM_HOST_H

#include "vm_internal.h"
#include <mutex>
#include <vector>

int vm_fail(int code, const char *fmt, ...);
struct vm_ctx;
bool vm_ctx_alive(const vm_ctx *c); // vm_api.cpp: is this context still alive?

#define VM_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return vm_fail(VM_E_DEVICE, "%s:%d %s: %s", __FILE__, __LINE__, #call,         \
                           hipGetErrorString(e_));                                         \
    } while (0)

// HIP's current device belongs to the calling host thread, and the API is driven from worker
// threads (MatchingThread, thread pools): every entry point that allocates or launches makes
// the context's device current first and restores the caller's device on return.
struct VmDeviceGuard {
    int prev = -1;
    bool switched = false;
    bool ok = true; // false: the device could not be made current -- nothing may be launched
    explicit VmDeviceGuard(int dev)
    {
        // HIP's "last error" is per thread and sticky across libraries: an error another
        // library left behind (RCCL probing peers, a framework's failed query) must not be
        // reported by the hipGetLastError() checks that follow this entry point's launches
        (void)hipGetLastError();
        if (hipGetDevice(&prev) != hipSuccess || prev != dev) {
            ok = hipSetDevice(dev) == hipSuccess;
            switched = ok && prev >= 0;
            (void)hipGetLastError();
        }
    }
    ~VmDeviceGuard()
    {
        if (switched)
            (void)hipSetDevice(prev);
    }
    VmDeviceGuard(const VmDeviceGuard &) = delete;
    VmDeviceGuard &operator=(const VmDeviceGuard &) = delete;
};
#define VM_CAT2(a, b) a##b
#define VM_CAT(a, b) VM_CAT2(a, b)
// In functions that return a status: a device that cannot be made current is an error, not a
// launch on whatever device the calling thread happened to have.
#define VM_ON_DEVICE(ctx)                                                                          \
    VmDeviceGuard VM_CAT(vm_device_guard_, __LINE__)((ctx)->device);                                \
    if (!VM_CAT(vm_device_guard_, __LINE__).ok)                                                    \
        return vm_fail(VM_E_DEVICE, "%s: device %d cannot be made current", __func__, (ctx)->device)
// ... and in destructors / void functions (best effort)
#define VM_ON_DEVICE_VOID(ctx) VmDeviceGuard VM_CAT(vm_device_guard_, __LINE__)((ctx)->device)

struct vm_ctx {
    std::recursive_mutex mu;         // a context is single-threaded by contract; this makes misuse safe
    int device = 0;
    int math_mode = VM_MATH_EXACT;
    vm_kern_params kp{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // done_ev: recorded on `stream` (under `mu`, never inside a graph capture) whenever a solver call has enqueued its
    // last write of a level; xfer_ev: scratch for the same purpose in a consumer that holds `mu` itself.  Another
    // context's stream waits on one of them instead of the host draining this stream (vm_frame_set_v_from_level)
    hipEvent_t done_ev = nullptr, xfer_ev = nullptr;
    uint32_t *tables = nullptr;      // VM_TAB_WORDS words
    uint32_t *flags = nullptr;       // per-iteration "improving" flags (device)
    uint32_t *flags_host = nullptr;  // pinned mirror
    uint32_t *stats = nullptr;       // per-iteration activity counters, 4 words each (device)
    uint32_t *stats_host = nullptr;  // pinned mirror
    uint32_t *step_slots = nullptr;  // STEP schedule: per-workgroup activity counts of the last two launches
    size_t step_slots_words = 0;     // capacity of ONE of the two halves, in words
    // PASS schedule: tile-barrier counters (one per tile group and launch of a batch), the
    // error word a timed-out barrier raises (+ pinned mirror), optional XCD-placement record
    uint32_t *pass_bar = nullptr;
    size_t pass_bar_words = 0;
    uint32_t *pass_err = nullptr, *pass_err_host = nullptr;
    int pass_resident[8] = {-1, -1, -1, -1, -1, -1, -1, -1}; // co-resident k_pass workgroups on this device, per arithmetic build (math_mode); -1: not asked yet
    uint32_t *pass_dbg = nullptr;    // vm_dbg_pass_xcd: 256 words, XCC id per workgroup of the last launch
    void *pass_snap = nullptr;       // AUTO: the levels' slabs as they stood before the current PASS batch
    size_t pass_snap_bytes = 0;
    bool pass_latched_off = false;   // AUTO: a tile barrier timed out once on this context: STEP from then on
    bool pass_latched_by_test = false; // ... and it was vm_dbg_pass_force_timeout's doing (only then the hook may lift it)
    int pass_fallbacks = 0;          // how often that happened (vm_dbg_pass_fallbacks)
    int pass_test_timeout = 0;       // vm_dbg_pass_force_timeout: the next PASS launches behave as if a barrier timed out
    int sweep_threads = 0;           // 0 = automatic
    int sweep_mode = 0;              // VM_SWEEP_AUTO / TILE / SPLIT
    int sweep_parts = 0;             // workgroups per tile in the SPLIT schedule, 0 = automatic
    int flags_cap = 0;
    VmLevelView *views = nullptr;    // device copies of the level views of the current batch
    int views_cap = 0;
    vm_constraint *cons_dev = nullptr;
    int cons_cap = 0;
    // hipGraph replay of launch-bound TILE sweeps (vm_api.cpp): 8 iterations per graph
    int *iter_dev = nullptr;         // device iteration counter read by the replayed kernels
    struct SweepGraph {
        int math_mode;
        int n, w, h, cap, fixed_work, threads, dense, order;
        const void *views, *flags, *stats, *tile_list;
        vm_kern_params kp;
        hipGraphExec_t exec;
    };
    std::vector<SweepGraph> graphs;
    int commit_order = 0;         // vm_set_commit_order (EXACT, diagnostic): order 0..3
    int sparse_resident = 0;      // vm_dbg_sparse_resident: 0 = automatic, 1 = never, 2 / 3 = tests (k_sparse, sv_phases)
    unsigned long long sparse_resident_visits = 0; // vm_dbg_sparse_resident_visits: tile visits served from the resident LDS copy
    uint32_t *tile_list = nullptr;   // the listed form of pruned TILE passes over big batches (k_tile_scan): counters, stamps, entries
    size_t tile_list_words = 0;
    int use_graphs = -1;             // -1: not decided yet, 0: off (VM_NO_GRAPH or a failed capture), 1: on
    void *mgb_sys = nullptr;         // device descriptors of the systems of the current Poisson batch (vm_poisson_api.cpp)
    void *mgb_shared = nullptr;      // ... and their PCG scalars + block / tile counts, contiguous: ONE clear and ONE read-back per check for the whole batch
    // vm_dbg_poisson_profile: HIP-event time of the launch that carries the PCG update (k_mgb_update, or the level-0
    // restriction with the update fused in), summed over the launches of the solves since the probe was switched on,
    // and what those launches processed
    bool mgb_prof = false;
    double mgb_prof_us = 0, mgb_prof_unknown_launches = 0;
    int mgb_prof_launches = 0, mgb_prof_fused = 0;
};

struct vm_level {
    int w = 0, h = 0, rs = 0, imp_rs = 0, imp_rows = 0;
    void *slab = nullptr;
    size_t slab_bytes = 0;
    void *ws = nullptr;              // SPLIT / STEP workspace, allocated on first use (vm_api.cpp)
    void *sp_ws = nullptr;           // SPARSE workspace (word lists, stamps), allocated on first use
    bool has_state = false;
    VmLevelView view{};
    // pages of a video level: where lvl.temp.ref / lvl.temp.mask of the page live (the view
    // points at them only while the page is swept with flag == true)
    float2 *temp_ref_store = nullptr;
    float *temp_mask_store = nullptr;
};

struct vm_pyr {
    vm_ctx *ctx = nullptr;
    int device = 0;                  // of ctx: the buffers can be freed after the context is gone
    std::vector<vm_level> lv;
};

// One page of a video level: the level state of one frame pair plus what couples it to its
// neighbours in time -- the four flow fields of the page (PyramidLevel::f0/f1/b0/b1,
// Pyramid.h:85-90, pitched float2 instead of cudaArray) and lvl.temp.ref / lvl.temp.mask.
struct vm_video_page {
    vm_level lv;
    void *tslab = nullptr;                                   // flows + temporal arrays
    float2 *flow[4] = {nullptr, nullptr, nullptr, nullptr};  // f0, f1, b0, b1
    float2 *temp_ref = nullptr;
    float *temp_mask = nullptr;
};

// The stage-2 pyramid of a video pair (class Pyramid with depth > 1, Pyramid.h:14-48):
// level l holds depth[l] pages; level 0 finest, the last level holds only v.
// A lane of the level pipeline of vm_video_solve: its own stream and sweep scratch (a context of
// its own on the same device) and its own splat accumulators, so that independent (level, chain
// step) tasks can run side by side (vm_video.cpp).
struct vm_video_lane {
    vm_ctx *c = nullptr;
    long long *acc = nullptr;
};

struct vm_video {
    vm_ctx *ctx = nullptr;
    int device = 0;
    int depth0 = 1;                       // frames of the video (the placeholder level's depth)
    std::vector<int> depth;               // pages per level
    std::vector<int> factor_t;            // temporal stride the level was built with (pyramid.cu:468)
    std::vector<float> factor_d;          // per level (pyramid.cu:470-477); factor_d0 for the placeholder
    float factor_d0 = 1.0f;
    std::vector<std::vector<vm_video_page>> pages;
    // scratch of the splat (sized for the finest level): fixed-point accumulators, v_cur, weight
    long long *acc = nullptr;
    float2 *vcur = nullptr;
    float *weight = nullptr;
    std::vector<vm_video_lane> lanes;     // created by the first pipelined solve
    float2 *result_tmp = nullptr;         // vm_frame_set_v_from_video: two full-resolution planes (blended frames)
    size_t result_tmp_elems = 0;
};

struct vm_frame {
    vm_ctx *ctx = nullptr;
    int device = 0;
    int w = 0, h = 0, ex = 0, cw = 0, ch = 0, rs = 0;
    uchar4 *ext[2] = {nullptr, nullptr};  // (w+2ex) x (h+2ex) RGBA8 canvases
    uchar4 *crop[2] = {nullptr, nullptr}; // w x h originals (CPoissonExt::_image1/_image2, PoissonExt.cpp:26-27)
    float2 *v = nullptr, *u = nullptr;    // h x rs
    bool u_zero = true;                   // the quadratic path is all zeros (never uploaded / computed: the reference app's
                                          // state, UI/MdiEditor.cpp:1898-1903): vm_render_halfway then skips its 21 taps of u --
                                          // a zero path stays zero through the fixed-point steps, the bytes are the same
    uint8_t *out = nullptr;               // h x w x 3
    uint8_t *rgb_stage = nullptr;         // vm_frame_upload_rgb: the two RGB8 frames as they arrive (2 x h x w x 3), allocated on first use
    // solver workspace (allocated on first use), pws2[side - 1]: one per side (both sides of a frame are in flight
    // together); the quadratic path uses side 1's
    void *pws2[2] = {nullptr, nullptr};
    size_t pws2_bytes[2] = {0, 0};
};

// level-wise pieces of the solver shared by the frame-pair API (vm_api.cpp) and the video
// API (vm_video.cpp); a "level" here is one page of one pyramid level
int vm_level_alloc(vm_ctx *c, vm_level &l, bool with_images);
void vm_level_free(vm_level &l);
int vm_level_upsample(vm_ctx *c, vm_level &dst, const vm_level &src);
int vm_level_init(vm_ctx *c, vm_level &l, int w0, int h0, const vm_constraint *cons, int n);
int vm_level_read_field(vm_ctx *c, vm_level &l, int field, void *host);
int vm_iteration_cap(float max_iter, int *cap);
int vm_optimize_levels(vm_ctx *c, vm_level **lv, int n, float max_iter, volatile const int *run_flag,
                       int fixed_work, vm_progress *out);

// Morph::cpu_optimize_level (morph.cu:419-590) on the host: v_out is a tight
// (h, w, 2) array
int vm_host_coarse_solve(int w, int h, int w0, int h0, const vm_kern_params &kp,
                         const vm_constraint *cons, int n, float *v_out, int depth = 1);

#endif
