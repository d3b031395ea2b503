// Internal (not installed) declarations shared by the translation units of libr3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/r3d.h"

struct r3d_buf {
    void *p = nullptr;
    size_t cap = 0;
};

#define R3D_MAX_PROF 16

struct r3d_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    bool profiling = false;
    // grow-only workspace
    r3d_buf img_l, img_r, rec_l, rec_r, cost, cspec, hsum, raw, mins, lrd, out, flags;
    // geometry of the last sgbm call (for debug fetch)
    int last_w = 0, last_h = 0, last_w1 = 0, last_dp = 0;
    // profiling slots
    hipEvent_t ev[R3D_MAX_PROF + 1] = {};
    const char *ev_name[R3D_MAX_PROF] = {};
    int n_ev = 0;
    bool ev_created = false;
    // cloud workspace (cloud.hip)
    std::vector<r3d_buf> cloud_bufs;
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
static inline void r3d_prof_begin(r3d_ctx *ctx) {
    ctx->n_ev = 0;
    if (!ctx->profiling) return;
    if (!ctx->ev_created) {
        for (int i = 0; i <= R3D_MAX_PROF; i++) (void)hipEventCreate(&ctx->ev[i]);
        ctx->ev_created = true;
    }
}
static inline void r3d_prof_mark(r3d_ctx *ctx, const char *name) {
    if (!ctx->profiling || ctx->n_ev >= R3D_MAX_PROF) return;
    (void)hipEventRecord(ctx->ev[ctx->n_ev], ctx->stream);
    ctx->ev_name[ctx->n_ev] = name;
    ctx->n_ev++;
}
static inline void r3d_prof_end(r3d_ctx *ctx) {
    if (!ctx->profiling) return;
    (void)hipEventRecord(ctx->ev[ctx->n_ev], ctx->stream);
}

// sgm.hip
int r3d_sgm_run(r3d_ctx *ctx, const r3d_sgbm_params *p, const uint8_t *d_left, const uint8_t *d_right, int w, int h,
                int stride, int16_t *d_disp);
int r3d_selftest_run(r3d_ctx *ctx);
