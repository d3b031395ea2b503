/*
 * r3d.h -- C ABI of lib3d_reconstruction_project_amd (libr3d_hip.so): the MI355X (gfx950) hot path of the
 * stereo-depth + point-cloud fusion pipeline.
 *
 * The reference (aagsi/3D_Reconstruction_Project) has NO FFI of its own: it is Python glue that calls
 * third-party C++ (OpenCV, Open3D) through their Python bindings.  Each entry point below therefore cites the
 * reference call site (file:line under /root/reference) whose third-party call it replaces; INTEGRATION.md
 * shows the ctypes stub a maintainer would add at that call site.
 *
 * Conventions: plain pointers and sizes only (no torch / numpy types); every function returns 0 (R3D_OK) or a
 * negative R3D_E_* code and leaves a message retrievable by r3d_last_error().  "not converged" is NOT an error.
 * One r3d_ctx = one HIP device + one HIP stream + a grow-only device workspace; calls on one ctx must be
 * serialised by the caller (the reference issues its hot calls from one worker thread, main.py:56-61);
 * different ctxs may be used from different threads.  `_dev` variants take DEVICE pointers, enqueue on the
 * ctx stream and do not synchronise; the others take HOST pointers, copy in/out and return after completion.
 */
#ifndef R3D_H
#define R3D_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R3D_OK 0
#define R3D_E_BADARG (-1)      /* null pointer, bad size, parameter outside the documented domain */
#define R3D_E_HIP (-2)         /* a HIP runtime call failed; message holds hipGetErrorString */
#define R3D_E_OOM (-3)         /* device or host allocation failed */
#define R3D_E_UNSUPPORTED (-4) /* valid for the reference, outside this library's exact-arithmetic envelope */
#define R3D_E_NODEVICE (-5)    /* no gfx950 device visible */

typedef struct r3d_ctx r3d_ctx;

/* ---- context ------------------------------------------------------------------------------------------- */
/* replaces: o3d.core.Device("CUDA:0") literals (normal_estimation.py:10, pointcloud_processing.py:12) */
int r3d_init(int device, r3d_ctx **out);
void r3d_destroy(r3d_ctx *ctx);
const char *r3d_last_error(const r3d_ctx *ctx); /* ctx may be NULL: returns the last r3d_init error */
int r3d_sync(r3d_ctx *ctx);                     /* hipStreamSynchronize on the ctx stream */
int r3d_set_stream(r3d_ctx *ctx, void *hip_stream /* hipStream_t, NULL = ctx-owned stream */);
void *r3d_get_stream(r3d_ctx *ctx);
/* checks the cross-lane primitives (DPP shifts, permlane swaps, wave reductions) the kernels rely on */
int r3d_selftest(r3d_ctx *ctx);

/* device memory + timing helpers so that a host language without a HIP binding (Python/ctypes) can keep
 * inputs resident in HBM and time kernels with HIP events on the ctx stream */
int r3d_dev_alloc(r3d_ctx *ctx, uint64_t bytes, void **out_dev_ptr);
int r3d_dev_free(r3d_ctx *ctx, void *dev_ptr);
int r3d_copy_h2d(r3d_ctx *ctx, void *dev_dst, const void *host_src, uint64_t bytes);
int r3d_copy_d2h(r3d_ctx *ctx, void *host_dst, const void *dev_src, uint64_t bytes);
int r3d_event_create(r3d_ctx *ctx, void **out_event);
int r3d_event_destroy(r3d_ctx *ctx, void *event);
int r3d_event_record(r3d_ctx *ctx, void *event);                        /* on the ctx stream */
int r3d_event_elapsed_ms(r3d_ctx *ctx, void *start, void *stop, float *out_ms); /* synchronises `stop` */

/* ---- stereo: semi-global block matching, 3-way ------------------------------------------------------------
 * replaces: cv2.StereoSGBM_create(**kwargs) + matcher.compute(gray_left, gray_right)
 *   Calib_depth/depth1.py:202-214,331  depth2.py:146-158,251  depth3.py:234-246,339
 *   Calib_depth/depth4.py:156-168,254  depth_test.py:162-174,260
 * Field names and meaning equal the StereoSGBM_create keyword arguments the reference passes.
 * Only mode == R3D_SGBM_MODE_3WAY (cv2.STEREO_SGBM_MODE_SGBM_3WAY == 2) is implemented. */
#define R3D_SGBM_MODE_3WAY 2
typedef struct {
    int32_t minDisparity;
    int32_t numDisparities; /* multiple of 16, <= 256 */
    int32_t blockSize;      /* odd, 1..11 */
    int32_t P1, P2;
    int32_t disp12MaxDiff;
    int32_t preFilterCap;
    int32_t uniquenessRatio;
    int32_t speckleWindowSize;
    int32_t speckleRange;
    int32_t mode;
} r3d_sgbm_params;

/* left/right: uint8 single-channel rectified images, `stride` bytes per row; disp: int16 w*h, disparity x16,
 * invalid = (minDisparity-1)*16, columns outside [max(minD+D,0), w+min(minD,0)) invalid -- as matcher.compute */
int r3d_sgbm_compute(r3d_ctx *ctx, const r3d_sgbm_params *p, const uint8_t *left, const uint8_t *right, int32_t w,
                     int32_t h, int32_t stride, int16_t *disp);
int r3d_sgbm_compute_dev(r3d_ctx *ctx, const r3d_sgbm_params *p, const uint8_t *d_left, const uint8_t *d_right,
                         int32_t w, int32_t h, int32_t stride, int16_t *d_disp);

/* per-kernel HIP-event timing (events on the ctx stream around every kernel launch while profiling is enabled).
 * r3d_sgbm_profile returns, and then resets, the AVERAGE launch duration per kernel over all r3d_sgbm_compute*
 * calls since the previous r3d_sgbm_profile.  names: NUL-separated list, one entry per slot; ms: one float per
 * slot.  Returns the number of slots.  (Events live in a 4-deep ring, so profiling never stalls the stream.) */
int r3d_set_profiling(r3d_ctx *ctx, int enabled);
int r3d_sgbm_profile(r3d_ctx *ctx, float *ms, int32_t max_slots, char *names, int32_t names_bytes);

/* debug / stage parity: copies intermediate results of the LAST sgbm call to HOST buffers (NULL = skip).
 *   cost   int16 [h][w1][dp]  aggregated block cost C (dp = 128 if D<=128 else 256; entries d>=D undefined)
 *   hsum   int16 [h][w1][dp]  L_left + L_right
 *   raw    int16 [h][w]       disparity after the row LR check, before the 3x3 median */
int r3d_sgbm_debug_fetch(r3d_ctx *ctx, int16_t *cost, int16_t *hsum, int16_t *raw);

#ifdef __cplusplus
}
#endif
#endif /* R3D_H */
