// Internal (not installed) declarations shared by the translation units of libr3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/r3d.h"

struct r3d_buf {
    void *p = nullptr;
    size_t cap = 0;
};

#define R3D_MAX_PROF 16
#define R3D_PIN_BYTES 65536
#define R3D_MAX_DEVICES 64   /* per-device caches of launch geometry */
#define R3D_PROF_SETS 4

struct r3d_prof_set {
    hipEvent_t ev[R3D_MAX_PROF + 1] = {};
    const char *name[R3D_MAX_PROF] = {};
    int n = 0;
    bool pending = false;
};

#define R3D_SGM_MAX_LANES 6   /* workspaces that can exist; r3d_sgbm_compute_batch* uses R3D_SGM_LANES of them (env, default 3) */
#define R3D_SGM_LANES 3
#define R3D_SGM_SLABS 8   // most column slabs of the cost / forward-scan overlap (sgm.hip)

// one SGM pipeline lane: its own grow-only workspace, stream and profiling event ring.  Single-map calls use lane 0 on
// the context stream; r3d_sgbm_compute_batch_dev spreads maps over the lanes so that kernels with complementary
// bottlenecks (cost: VALU + writes, hscan: HBM, vscan: mixed) of consecutive maps overlap.
struct r3d_sgm_ws {
    r3d_buf rec_l, rec_r, cost, cspec, hsum, ltop, ckpt, raw, mins, lrd, lrd2, flags, spk_l, spk_c;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipStream_t aux = nullptr;                       // second stream of the lane: cost slabs ahead of the forward scan
    hipEvent_t vs_fork = nullptr, vs_join = nullptr;   // the balanced split of the vertical scan (tail launch on `aux`)
    hipEvent_t slab_ev[R3D_SGM_SLABS + 1] = {};      // [j]: cost of slab j written; [R3D_SGM_SLABS]: fork point
    r3d_prof_set prof[R3D_PROF_SETS];
    int prof_cur = 0;
    bool ev_created = false;
};

struct r3d_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    bool profiling = false;
    // set when a call gave up on work that is still queued on the stream (the registration loop's deadline): the arena and the
    // workspaces that work uses must not be handed to another call, so every later compute call on this ctx fails (R3D_E_HIP)
    // until it is destroyed
    bool poisoned = false;
    // grow-only workspace
    r3d_buf img_l, img_r, out;
    r3d_sgm_ws ws[R3D_SGM_MAX_LANES];
    hipEvent_t fork_ev = nullptr;
    // geometry of the last sgbm call (for debug fetch)
    int last_w = 0, last_h = 0, last_w1 = 0, last_dp = 0, last_impl = 0;
    // profiling sums accumulate per kernel name over all lanes
    const char *acc_name[R3D_MAX_PROF] = {};
    double acc_ms[R3D_MAX_PROF] = {};
    int acc_cnt[R3D_MAX_PROF] = {};
    int n_acc = 0;
    // cloud workspace (cloud.hip)
    std::vector<r3d_buf> cloud_bufs;
    hipEvent_t icp_ev = nullptr;   // polled once per registration iteration
    double *icp_host = nullptr;    // pinned landing buffer of the per-iteration sums
    unsigned pin_seq = 0;          // sequence number of the latest PinRead
    void *pin = nullptr;           // pinned landing buffer of the small device -> host reads between kernels (R3D_PIN_BYTES)
    // pre/post-processing workspace (prepost.hip)
    std::vector<r3d_buf> pp_bufs;
    r3d_buf pp_minmax, pp_lut;
    double pp_lut_sigma = -1.0;
    int pp_lut_n = 0;
};

int r3d_fail(r3d_ctx *ctx, int code, const char *fmt, ...);
int r3d_reserve(r3d_ctx *ctx, r3d_buf &b, size_t bytes);

#define R3D_HIP(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return r3d_fail((ctx), R3D_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                                    \
    } while (0)

// profiling helpers: record an event before each named kernel when ctx->profiling
void r3d_prof_harvest(r3d_ctx *ctx, r3d_prof_set &ps);
static inline void r3d_prof_begin(r3d_ctx *ctx, r3d_sgm_ws &ws) {
    if (!ctx->profiling) return;
    if (!ws.ev_created) {
        for (auto &ps : ws.prof)
            for (int i = 0; i <= R3D_MAX_PROF; i++) (void)hipEventCreate(&ps.ev[i]);
        ws.ev_created = true;
    }
    ws.prof_cur = (ws.prof_cur + 1) % R3D_PROF_SETS;
    r3d_prof_set &ps = ws.prof[ws.prof_cur];
    if (ps.pending) r3d_prof_harvest(ctx, ps);
    ps.n = 0;
}
static inline void r3d_prof_mark(r3d_ctx *ctx, r3d_sgm_ws &ws, hipStream_t st, const char *name) {
    if (!ctx->profiling) return;
    r3d_prof_set &ps = ws.prof[ws.prof_cur];
    if (ps.n >= R3D_MAX_PROF) return;
    (void)hipEventRecord(ps.ev[ps.n], st);
    ps.name[ps.n] = name;
    ps.n++;
}
static inline void r3d_prof_end(r3d_ctx *ctx, r3d_sgm_ws &ws, hipStream_t st) {
    if (!ctx->profiling) return;
    r3d_prof_set &ps = ws.prof[ws.prof_cur];
    (void)hipEventRecord(ps.ev[ps.n], st);
    ps.pending = true;
}


// roctx ranges around every C-ABI entry point ("rocprofv3 --marker-trace attributes kernels to entry
// points").  The marker library is looked up at run time (librocprofiler-sdk-roctx.so, the one rocprofv3 listens to, else the
// legacy libroctx64.so); without it, or with R3D_ROCTX=0, a range is two predictable branches.
bool r3d_roctx_push(const char *name);   // true if a range was opened (the scope pops only then: contexts live on several threads)
void r3d_roctx_pop();
struct r3d_roctx_scope {
    bool pushed;
    explicit r3d_roctx_scope(const char *name) : pushed(r3d_roctx_push(name)) {}
    ~r3d_roctx_scope() { if (pushed) r3d_roctx_pop(); }
    r3d_roctx_scope(const r3d_roctx_scope &) = delete;
    r3d_roctx_scope &operator=(const r3d_roctx_scope &) = delete;
};
#define R3D_ROCTX_RANGE(name) r3d_roctx_scope r3d_roctx_scope_(name)

// sgm.hip
int r3d_sgm_run(r3d_ctx *ctx, int lane, hipStream_t st, const r3d_sgbm_params *p, const uint8_t *d_left, const uint8_t *d_right,
                int w, int h, int stride, int16_t *d_disp);
int r3d_selftest_run(r3d_ctx *ctx);
int r3d_speckle_run(r3d_ctx *ctx, r3d_sgm_ws &ws, hipStream_t st, int16_t *d_img, int w, int h, int newVal, int maxSize, int maxDiff);
int r3d_streambench_run(r3d_ctx *ctx, int mode, int rows, size_t row_bytes, int write, int delay, int reps, float *ms);
