// api.hip -- context, memory/timing helpers and the  extern "C" entry points declared in include/r3d.h.
#include <dlfcn.h>
#include <mutex>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "r3d_internal.h"

static std::string g_init_err;

// ---- roctx markers, resolved lazily with dlopen (no link-time dependency on a profiler library) -------------------------------
namespace {
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)();
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
std::once_flag g_roctx_once;   // one ctx per calling thread is the documented model: the lookup must be safe under concurrent first calls
void roctx_lookup() {
    const char *e = getenv("R3D_ROCTX");
    if (e && !strcmp(e, "0")) return;
    for (const char *lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        roctx_push_fn pu = (roctx_push_fn)dlsym(h, "roctxRangePushA");
        roctx_pop_fn po = (roctx_pop_fn)dlsym(h, "roctxRangePop");
        if (pu && po) { g_roctx_pop = po; g_roctx_push = pu; return; }
    }
}
}  // namespace
bool r3d_roctx_push(const char *name) {
    std::call_once(g_roctx_once, roctx_lookup);   // both pointers are published before call_once returns on any thread
    if (!g_roctx_push) return false;
    (void)g_roctx_push(name);
    return true;
}
void r3d_roctx_pop() {
    if (g_roctx_pop) (void)g_roctx_pop();
}


int r3d_fail(r3d_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_init_err = buf;
    return code;
}

int r3d_reserve(r3d_ctx *ctx, r3d_buf &b, size_t bytes) {
    if (bytes <= b.cap) return R3D_OK;
    if (b.p) {
        (void)hipDeviceSynchronize();   // lanes run on their own streams
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return r3d_fail(ctx, R3D_E_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return R3D_OK;
}

void r3d_prof_harvest(r3d_ctx *ctx, r3d_prof_set &ps) {
    ps.pending = false;
    if (ps.n == 0 || hipEventSynchronize(ps.ev[ps.n]) != hipSuccess) return;
    for (int i = 0; i < ps.n; i++) {
        float t = 0;
        if (hipEventElapsedTime(&t, ps.ev[i], ps.ev[i + 1]) != hipSuccess) continue;
        int k = 0;
        while (k < ctx->n_acc && strcmp(ctx->acc_name[k], ps.name[i]) != 0) k++;
        if (k == ctx->n_acc) {
            if (ctx->n_acc >= R3D_MAX_PROF) continue;
            ctx->acc_name[k] = ps.name[i];
            ctx->acc_ms[k] = 0;
            ctx->acc_cnt[k] = 0;
            ctx->n_acc++;
        }
        ctx->acc_ms[k] += t;
        ctx->acc_cnt[k]++;
    }
}

extern "C" {

int r3d_init(int device, r3d_ctx **out) {
    if (!out) return r3d_fail(nullptr, R3D_E_BADARG, "r3d_init: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return r3d_fail(nullptr, R3D_E_NODEVICE, "r3d_init: no HIP device (%s)", e != hipSuccess ? hipGetErrorString(e) : "count=0");
    if (device < 0 || device >= n) return r3d_fail(nullptr, R3D_E_BADARG, "r3d_init: device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return r3d_fail(nullptr, R3D_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return r3d_fail(nullptr, R3D_E_NODEVICE, "r3d_init: device %d is %s; this library contains gfx950 code only", device, prop.gcnArchName);
    r3d_ctx *ctx = new (std::nothrow) r3d_ctx();
    if (!ctx) return r3d_fail(nullptr, R3D_E_OOM, "r3d_init: out of host memory");
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        int rc = r3d_fail(nullptr, R3D_E_HIP, "r3d_init: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return R3D_OK;
}

void r3d_destroy(r3d_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipDeviceSynchronize();
    r3d_buf *bufs[] = {&ctx->img_l, &ctx->img_r, &ctx->out};
    for (r3d_buf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (r3d_sgm_ws &ws : ctx->ws) {
        r3d_buf *wb[] = {&ws.rec_l, &ws.rec_r, &ws.cost, &ws.cspec, &ws.hsum, &ws.ltop, &ws.ckpt, &ws.raw, &ws.mins, &ws.lrd, &ws.lrd2, &ws.flags, &ws.spk_l, &ws.spk_c};
        for (r3d_buf *b : wb)
            if (b->p) (void)hipFree(b->p);
        if (ws.ev_created)
            for (auto &ps : ws.prof)
                for (int i = 0; i <= R3D_MAX_PROF; i++) (void)hipEventDestroy(ps.ev[i]);
        if (ws.stream) (void)hipStreamDestroy(ws.stream);
        if (ws.done) (void)hipEventDestroy(ws.done);
        if (ws.aux) (void)hipStreamDestroy(ws.aux);
        if (ws.vs_fork) (void)hipEventDestroy(ws.vs_fork);
        if (ws.vs_join) (void)hipEventDestroy(ws.vs_join);
        for (hipEvent_t e : ws.slab_ev)
            if (e) (void)hipEventDestroy(e);
    }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->icp_ev) (void)hipEventDestroy(ctx->icp_ev);
    if (ctx->icp_host) (void)hipHostFree(ctx->icp_host);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    for (r3d_buf &b : ctx->cloud_bufs)
        if (b.p) (void)hipFree(b.p);
    for (r3d_buf &b : ctx->pp_bufs)
        if (b.p) (void)hipFree(b.p);
    if (ctx->pp_minmax.p) (void)hipFree(ctx->pp_minmax.p);
    if (ctx->pp_lut.p) (void)hipFree(ctx->pp_lut.p);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *r3d_last_error(const r3d_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_err.c_str(); }

int r3d_sync(r3d_ctx *ctx) {
    R3D_ROCTX_RANGE("r3d_sync");
    if (!ctx) return R3D_E_BADARG;
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

int r3d_set_stream(r3d_ctx *ctx, void *s) {
    if (!ctx) return R3D_E_BADARG;
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return R3D_OK;
}
void *r3d_get_stream(r3d_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int r3d_stream_wait_event(r3d_ctx *ctx, void *ev) {
    if (!ctx || !ev) return R3D_E_BADARG;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    R3D_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)ev, 0));
    return R3D_OK;
}

int r3d_debug_streambench(r3d_ctx *ctx, int32_t mode, int32_t rows, uint64_t row_bytes, int32_t write, int32_t delay, int32_t reps, float *ms) {
    R3D_ROCTX_RANGE("r3d_debug_streambench");
    if (!ctx || !ms || rows <= 0 || reps <= 0) return R3D_E_BADARG;
    return r3d_streambench_run(ctx, mode, rows, row_bytes, write, delay, reps, ms);
}

int r3d_selftest(r3d_ctx *ctx) {
    R3D_ROCTX_RANGE("r3d_selftest");
    if (!ctx) return R3D_E_BADARG;
    return r3d_selftest_run(ctx);
}

int r3d_dev_alloc(r3d_ctx *ctx, uint64_t bytes, void **out) {
    R3D_ROCTX_RANGE("r3d_dev_alloc");
    if (!ctx || !out) return R3D_E_BADARG;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) return r3d_fail(ctx, R3D_E_OOM, "hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    return R3D_OK;
}
int r3d_dev_free(r3d_ctx *ctx, void *p) {
    R3D_ROCTX_RANGE("r3d_dev_free");
    if (!ctx) return R3D_E_BADARG;
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    R3D_HIP(ctx, hipFree(p));
    return R3D_OK;
}
int r3d_copy_h2d(r3d_ctx *ctx, void *d, const void *h, uint64_t bytes) {
    R3D_ROCTX_RANGE("r3d_copy_h2d");
    if (!ctx || (!d && bytes) || (!h && bytes)) return R3D_E_BADARG;
    R3D_HIP(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}
int r3d_copy_d2h(r3d_ctx *ctx, void *h, const void *d, uint64_t bytes) {
    R3D_ROCTX_RANGE("r3d_copy_d2h");
    if (!ctx || (!d && bytes) || (!h && bytes)) return R3D_E_BADARG;
    R3D_HIP(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}
int r3d_event_create(r3d_ctx *ctx, void **out) {
    if (!ctx || !out) return R3D_E_BADARG;
    hipEvent_t e;
    R3D_HIP(ctx, hipEventCreate(&e));
    *out = (void *)e;
    return R3D_OK;
}
int r3d_event_destroy(r3d_ctx *ctx, void *e) {
    if (!ctx) return R3D_E_BADARG;
    R3D_HIP(ctx, hipEventDestroy((hipEvent_t)e));
    return R3D_OK;
}
int r3d_event_record(r3d_ctx *ctx, void *e) {
    if (!ctx || !e) return R3D_E_BADARG;
    R3D_HIP(ctx, hipEventRecord((hipEvent_t)e, ctx->stream));
    return R3D_OK;
}
int r3d_event_elapsed_ms(r3d_ctx *ctx, void *a, void *b, float *ms) {
    if (!ctx || !a || !b || !ms) return R3D_E_BADARG;
    R3D_HIP(ctx, hipEventSynchronize((hipEvent_t)b));
    R3D_HIP(ctx, hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return R3D_OK;
}

int r3d_set_profiling(r3d_ctx *ctx, int enabled) {
    if (!ctx) return R3D_E_BADARG;
    ctx->profiling = enabled != 0;
    return R3D_OK;
}

int r3d_sgbm_profile(r3d_ctx *ctx, float *ms, int32_t max_slots, char *names, int32_t names_bytes) {
    if (!ctx) return R3D_E_BADARG;
    for (r3d_sgm_ws &ws : ctx->ws)
        for (auto &ps : ws.prof)
            if (ps.pending) r3d_prof_harvest(ctx, ps);
    int n = ctx->n_acc < max_slots ? ctx->n_acc : max_slots;
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        if (ms) ms[i] = ctx->acc_cnt[i] ? (float)(ctx->acc_ms[i] / ctx->acc_cnt[i]) : 0.f;
        if (names) {
            size_t len = strlen(ctx->acc_name[i]) + 1;
            if (off + len <= (size_t)names_bytes) { memcpy(names + off, ctx->acc_name[i], len); off += len; }
        }
    }
    ctx->n_acc = 0;
    return n;
}

int r3d_sgbm_compute_dev(r3d_ctx *ctx, const r3d_sgbm_params *p, const uint8_t *d_left, const uint8_t *d_right, int32_t w,
                         int32_t h, int32_t stride, int16_t *d_disp) {
    R3D_ROCTX_RANGE("r3d_sgbm_compute_dev");
    if (!ctx) return R3D_E_BADARG;
    return r3d_sgm_run(ctx, 0, ctx->stream, p, d_left, d_right, w, h, stride, d_disp);
}

int r3d_sgbm_compute_batch_dev(r3d_ctx *ctx, const r3d_sgbm_params *p, int32_t n, const uint8_t *const *d_left,
                               const uint8_t *const *d_right, int32_t w, int32_t h, int32_t stride, int16_t *const *d_disp) {
    return r3d_sgbm_compute_batch_events_dev(ctx, p, n, d_left, d_right, w, h, stride, d_disp, nullptr);
}

int r3d_sgbm_compute_batch_events_dev(r3d_ctx *ctx, const r3d_sgbm_params *p, int32_t n, const uint8_t *const *d_left,
                                      const uint8_t *const *d_right, int32_t w, int32_t h, int32_t stride, int16_t *const *d_disp,
                                      void *const *done_events) {
    R3D_ROCTX_RANGE("r3d_sgbm_compute_batch_dev");
    if (!ctx) return R3D_E_BADARG;
    if (n < 0 || (n > 0 && (!d_left || !d_right || !d_disp))) return r3d_fail(ctx, R3D_E_BADARG, "sgbm batch: bad argument");
    if (n == 0) return R3D_OK;
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    // maps in flight: R3D_SGM_LANES (default 3; measured A/B values 2 .. 6)
    static const int max_lanes = [] { const char *e = getenv("R3D_SGM_LANES"); const int v = e ? atoi(e) : R3D_SGM_LANES; return v < 1 ? 1 : (v > R3D_SGM_MAX_LANES ? R3D_SGM_MAX_LANES : v); }();
    const int lanes = n < max_lanes ? n : max_lanes;
    if (!ctx->fork_ev) R3D_HIP(ctx, hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
    for (int l = 0; l < lanes; l++) {
        r3d_sgm_ws &ws = ctx->ws[l];
        if (!ws.stream) R3D_HIP(ctx, hipStreamCreateWithFlags(&ws.stream, hipStreamNonBlocking));
        if (!ws.done) R3D_HIP(ctx, hipEventCreateWithFlags(&ws.done, hipEventDisableTiming));
    }
    // fork: every lane starts after whatever is already queued on the context stream
    R3D_HIP(ctx, hipEventRecord(ctx->fork_ev, ctx->stream));
    for (int l = 0; l < lanes; l++) R3D_HIP(ctx, hipStreamWaitEvent(ctx->ws[l].stream, ctx->fork_ev, 0));
    int rc = R3D_OK;
    for (int i = 0; i < n && rc == R3D_OK; i++) {
        const int l = i % lanes;
        rc = r3d_sgm_run(ctx, l, ctx->ws[l].stream, p, d_left[i], d_right[i], w, h, stride, d_disp[i]);
        if (rc == R3D_OK && done_events && done_events[i]) {
            // a stale or foreign-device event handle fails here: remember it, stop launching, and STILL run the join below
            const hipError_t e = hipEventRecord((hipEvent_t)done_events[i], ctx->ws[l].stream);
            if (e != hipSuccess) rc = r3d_fail(ctx, R3D_E_HIP, "sgbm batch: recording done_events[%d] failed: %s", i, hipGetErrorString(e));
        }
    }
    // join: the context stream continues after every lane has drained -- on EVERY exit path after the fork, so that maps already
    // enqueued on the lanes stay ordered before whatever the caller queues next on the context stream
    for (int l = 0; l < lanes; l++) {
        hipError_t e = hipEventRecord(ctx->ws[l].done, ctx->ws[l].stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ws[l].done, 0);
        if (e != hipSuccess) {                 // cannot order the lane by an event: drain it on the host instead
            (void)hipStreamSynchronize(ctx->ws[l].stream);
            if (rc == R3D_OK) rc = r3d_fail(ctx, R3D_E_HIP, "sgbm batch: joining lane %d failed: %s", l, hipGetErrorString(e));
        }
    }
    return rc;
}

int r3d_sgbm_compute(r3d_ctx *ctx, const r3d_sgbm_params *p, const uint8_t *left, const uint8_t *right, int32_t w, int32_t h,
                     int32_t stride, int16_t *disp) {
    R3D_ROCTX_RANGE("r3d_sgbm_compute");
    if (!ctx) return R3D_E_BADARG;
    if (!left || !right || !disp) return r3d_fail(ctx, R3D_E_BADARG, "sgbm: null host pointer");
    if (w <= 0 || h <= 0 || stride < w) return r3d_fail(ctx, R3D_E_BADARG, "sgbm: bad size %dx%d stride %d", w, h, stride);
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ib = (size_t)stride * h, ob = (size_t)w * h * 2;
    int rc;
    if ((rc = r3d_reserve(ctx, ctx->img_l, ib)) || (rc = r3d_reserve(ctx, ctx->img_r, ib)) || (rc = r3d_reserve(ctx, ctx->out, ob))) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(ctx->img_l.p, left, ib, hipMemcpyHostToDevice, ctx->stream));
    R3D_HIP(ctx, hipMemcpyAsync(ctx->img_r.p, right, ib, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = r3d_sgm_run(ctx, 0, ctx->stream, p, (const uint8_t *)ctx->img_l.p, (const uint8_t *)ctx->img_r.p, w, h, stride, (int16_t *)ctx->out.p))) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(disp, ctx->out.p, ob, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

int r3d_filter_speckles(r3d_ctx *ctx, int16_t *img, int32_t w, int32_t h, int32_t new_val, int32_t max_speckle_size, int32_t max_diff) {
    R3D_ROCTX_RANGE("r3d_filter_speckles");
    if (!ctx) return R3D_E_BADARG;
    if (!img || w <= 0 || h <= 0) return r3d_fail(ctx, R3D_E_BADARG, "filter_speckles: bad argument");
    R3D_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)w * h * 2;
    int rc;
    if ((rc = r3d_reserve(ctx, ctx->out, bytes))) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(ctx->out.p, img, bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = r3d_speckle_run(ctx, ctx->ws[0], ctx->stream, (int16_t *)ctx->out.p, w, h, new_val, max_speckle_size, max_diff))) return rc;
    R3D_HIP(ctx, hipMemcpyAsync(img, ctx->out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return R3D_OK;
}

int r3d_sgbm_debug_fetch(r3d_ctx *ctx, int16_t *cost, int16_t *hsum, int16_t *raw) {
    R3D_ROCTX_RANGE("r3d_sgbm_debug_fetch");
    if (!ctx) return R3D_E_BADARG;
    if (ctx->last_w == 0) return r3d_fail(ctx, R3D_E_BADARG, "debug_fetch: no sgbm call yet");
    if (ctx->last_w1 <= 0) return r3d_fail(ctx, R3D_E_BADARG, "debug_fetch: the last call had an empty matching range (all-invalid map), no volumes exist");
    R3D_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t vol = (size_t)ctx->last_h * ctx->last_w1 * ctx->last_dp * 2;
    if (cost) R3D_HIP(ctx, hipMemcpy(cost, ctx->ws[0].cost.p, vol, hipMemcpyDeviceToHost));
    if (hsum) {
        if (ctx->last_impl == 3) return r3d_fail(ctx, R3D_E_UNSUPPORTED, "debug_fetch: the v3 pipeline never materialises L_left + L_right (set R3D_SGM_IMPL=v2)");
        R3D_HIP(ctx, hipMemcpy(hsum, ctx->ws[0].hsum.p, vol, hipMemcpyDeviceToHost));
    }
    if (raw) R3D_HIP(ctx, hipMemcpy(raw, ctx->ws[0].lrd.p, (size_t)ctx->last_w * ctx->last_h * 2, hipMemcpyDeviceToHost));
    return R3D_OK;
}

}  // extern "C"
