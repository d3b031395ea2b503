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
/* hip_stream: a hipStream_t.  NULL = back to the ctx-owned stream (created hipStreamNonBlocking: NOT ordered against the null
 * stream).  The legacy null stream is therefore not selectable as 0: pass hipStreamLegacy ((hipStream_t)1) for it, or -- what the
 * Python layer does (distributed.shared_stream) -- make a named stream current on the caller's side and pass that. */
int r3d_set_stream(r3d_ctx *ctx, void *hip_stream);
void *r3d_get_stream(r3d_ctx *ctx);
/* the ctx stream waits for a hipEvent_t (created by r3d_event_create on any ctx of this device, or by the caller): the join
 * between two contexts / streams without a host synchronisation */
int r3d_stream_wait_event(r3d_ctx *ctx, void *hip_event);
/* diagnostic: prices an access shape (rows independent streams of row_bytes; mode 0 = 256 B, 1 = 1 KB per wave request) */
int r3d_debug_streambench(r3d_ctx *ctx, int32_t mode, int32_t rows, uint64_t row_bytes, int32_t write, int32_t delay, int32_t reps, float *ms);
/* diagnostic: the stable sort of point indices by (cell key, index) that every grid / voxel build starts with.  key_order 0: x
 * fastest (search grid), 1: legacy voxel index (z fastest), 2: Morton code of the cell, 3: tensor voxel index; impl 0: the
 * hand-written run-based counting sort (default path), 1: the library radix sort (fallback path).  keys_out may be NULL. */
int r3d_debug_sort_by_cell(r3d_ctx *ctx, const double *xyz, int64_t n, const double *org3, double cell, const int32_t *dims3, int32_t key_order,
                           int32_t impl, int32_t *idx_out, uint64_t *keys_out);
/* diagnostic: the hand-written exclusive scan (k_scan_sums + k_scan_apply) on n int32 host values; op 0 = sum, 1 = running maximum
 * (values >= 0).  out[i] = op over in[0 .. i-1], out[0] = 0. */
int r3d_debug_exclusive_scan(r3d_ctx *ctx, const int32_t *in, int64_t n, int32_t op, int32_t *out);
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

/* n independent pairs of equal size (a multi-view batch, BASELINE config C5, or consecutive video frames): the maps
 * are spread over up to 3 internal lanes (own stream + workspace each) so that kernels of different maps overlap;
 * forks from / joins into the ctx stream, i.e. to the caller it behaves like n r3d_sgbm_compute_dev calls. */
int r3d_sgbm_compute_batch_dev(r3d_ctx *ctx, const r3d_sgbm_params *p, int32_t n, const uint8_t *const *d_left,
                               const uint8_t *const *d_right, int32_t w, int32_t h, int32_t stride, int16_t *const *d_disp);
/* the same, and map i's completion is recorded into done_events[i] (hipEvent_t, may be NULL per entry) on the lane that ran it:
 * a consumer on ANOTHER stream / context (r3d_stream_wait_event) can start on map i while later maps are still in flight
 * (the C5 view chain on one GPU: the cloud stages of view i run underneath the SGM kernels of views i+1 ...) */
int r3d_sgbm_compute_batch_events_dev(r3d_ctx *ctx, const r3d_sgbm_params *p, int32_t n, const uint8_t *const *d_left,
                                      const uint8_t *const *d_right, int32_t w, int32_t h, int32_t stride, int16_t *const *d_disp,
                                      void *const *done_events);

/* cv2.filterSpeckles(img, newVal, maxSpeckleSize, maxDiff) on an int16 image, in place (host buffer): the last stage of
 * StereoSGBM.compute when speckleWindowSize > 0 (Calib_depth/depth4.py:164-165, depth_test.py:170-171) */
int r3d_filter_speckles(r3d_ctx *ctx, int16_t *img, int32_t w, int32_t h, int32_t new_val, int32_t max_speckle_size, int32_t max_diff);

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

/* ---- point clouds (float64 xyz triplets, like the legacy open3d.geometry.PointCloud the reference passes) ----- */

/* replaces: pcd.voxel_down_sample(voxel_size)   pointcloud_alignment.py:22-23, test/check84.py:180
 * origin = min_bound - voxel/2, key = floor((p-origin)/voxel); out = per-voxel mean of points (and colors /
 * normals when given; pass NULL otherwise).  Output arrays need room for n triplets; *out_n = voxels written, in
 * lexicographic key order (the original's order is hash-map order, i.e. unspecified). */
int r3d_voxel_downsample(r3d_ctx *ctx, const double *xyz, const double *colors, const double *normals, int64_t n, double voxel,
                         double *out_xyz, double *out_colors, double *out_normals, int64_t *out_n);

/* replaces: o3d.t.geometry.PointCloud.voxel_down_sample(voxel) on a Float32 tensor cloud   pointcloud_processing.py:26-27,
 * test/GICP1.py:71-72   [recalled, Open3D 0.18 reduction "mean"].  Unlike the legacy grid above: coordinates and attributes are
 * rounded to float32, key = floor(p / voxel) evaluated in float32 with the grid origin at 0, sums and the division in float32
 * (members added in their original order; the original's order is that of device atomics).  Outputs hold float32 values. */
int r3d_voxel_downsample_tensor(r3d_ctx *ctx, const double *xyz, const double *colors, const double *normals, int64_t n, double voxel,
                                double *out_xyz, double *out_colors, double *out_normals, int64_t *out_n);

/* replaces: pcd.estimate_normals(KDTreeSearchParamHybrid(radius, max_nn))   pointcloud_alignment.py:27-28,
 * test/GICP1.py:77,95,97,148; tensor estimate_normals(max_nn, radius)   normal_estimation.py:20.
 * radius <= 0 selects KDTreeSearchParamKNN(max_nn).  <= max_nn nearest neighbours with distance < radius (query
 * included, nearest first); covariance and normal as legacy Open3D computes them -- one pass of nine cumulants over the raw
 * coordinates, then FastEigen3x3 (closed form), with the fused multiply-adds of the build that recorded the reference's frames:
 * the normals of those frames are reproduced SIGN INCLUDED to <= 5e-12 (oracle/normals.c has the derivation); the sign is the
 * one the closed form produces.  Fewer than 3 neighbours (or an all-zero covariance) -> (0,0,1).
 * prev_normals (may be NULL): when given, each normal is flipped to agree with it and a degenerate one keeps it (legacy
 * behaviour of a cloud that already carries normals). */
int r3d_estimate_normals(r3d_ctx *ctx, const double *xyz, int64_t n, double radius, int32_t max_nn, const double *prev_normals,
                         double *normals);

/* per-point neighbourhood score used by remove_statistical_outlier / remove_radius_outlier
 * (pointcloud_processing.py:35,39; test/check_lama1.py:175): count_radius <= 0: mean distance to the k nearest
 * points, the point itself included; count_radius > 0: number of points within that radius, itself included. */
int r3d_neighbor_score(r3d_ctx *ctx, const double *xyz, int64_t n, int32_t k, double count_radius, double *score);

/* disparity map -> point cloud, cv2.reprojectImageTo3D semantics ([X Y Z W]^T = Q [x y disp/16 1]^T, point = XYZ/W).
 * The reference computes / loads Q (Calib_depth/depth1.py:169,183, depth2.py:49,66) but never calls reprojectImageTo3D;
 * this is the join between its two halves.  Pixels with disp < min_valid_x16 are dropped;
 * out_xyz needs room for w*h triplets, out_pixel (may be NULL) receives the row-major pixel index of each point. */
int r3d_reproject_disparity(r3d_ctx *ctx, const int16_t *disp, int32_t w, int32_t h, const double *Q4x4, int32_t min_valid_x16,
                            double *out_xyz, int32_t *out_pixel, int64_t *out_n);

/* Device-resident join of the two halves ("SGM -> disparity -> cloud -> voxel -> normals, all
 * device-resident"): takes the DEVICE disparity map r3d_sgbm_compute_dev wrote and chains reprojectImageTo3D semantics
 * -> |z| <= max_depth filter (max_depth <= 0: only points at infinity, W = 0, are dropped) -> rigid pose (pose4x4 may be NULL) -> voxel_down_sample(voxel)
 * (voxel <= 0: off) -> estimate_normals(KDTreeSearchParamHybrid(normal_radius, max_nn)) (max_nn <= 0: off; radius <= 0: kNN)
 * without leaving HBM; only the final cloud is copied to the host arrays (room for `capacity` triplets each; out_normals
 * may be NULL when max_nn <= 0).  *out_n = points produced; R3D_E_BADARG if they exceed capacity.  Same kernels and
 * arithmetic as the separate host-buffer entry points, so the results are identical to chaining those. */
int r3d_disparity_to_cloud_dev(r3d_ctx *ctx, const int16_t *d_disp, int32_t w, int32_t h, const double *Q4x4, int32_t min_valid_x16,
                               double max_depth, const double *pose4x4, double voxel, double normal_radius, int32_t max_nn,
                               int64_t capacity, double *out_xyz, double *out_normals, int64_t *out_n);

/* k-nearest-neighbour graph (indices in the caller's numbering, nearest first, the point itself first; missing
 * entries -1 / 1e300).  radius <= 0: unbounded.  Feeds orient_normals_consistent_tangent_plane(k)
 * (normal_estimation.py:21), whose spanning-tree propagation is sequential host work.  d2 may be NULL. */
int r3d_knn_graph(r3d_ctx *ctx, const double *xyz, int64_t n, int32_t k, double radius, int32_t *nbr, double *d2);

/* replaces: pcd.orient_normals_consistent_tangent_plane(k)   normal_estimation.py:21   (Open3D legacy
 * OrientNormalsConsistentTangentPlane [recalled]; the tensor method converts to legacy and calls it).
 * r3d_orient_normals_graph is the exact form: `delaunay_edges` = n_edges (a, b) index pairs, the edges of the Delaunay
 * tetrahedralisation of the cloud (Open3D obtains it from Qhull on the host; the Python layer does the same).  The library
 * builds the Euclidean MST of those edges, adds the k-nearest-neighbour edges from the device graph (the point itself is one
 * of the k) that are not Delaunay edges, weights 1 - |n_a . n_b|, takes the spanning tree and propagates the sign from the
 * highest point (turned towards +z).  Equal weights are visited in (a, b) order (the original's order is unspecified).
 * r3d_orient_normals is the k-NN-graph-only variant for callers without a tetrahedralisation: no Euclidean-MST edges, every
 * connected component is rooted at its own highest point -- a DEVIATION from Open3D, not used by the drop-in classes.
 * Spanning trees and propagation are sequential host work inside the library; normals are flipped in place. */
int r3d_orient_normals(r3d_ctx *ctx, const double *xyz, int64_t n, int32_t k, double *normals);
int r3d_orient_normals_graph(r3d_ctx *ctx, const double *xyz, int64_t n, int32_t k, const int32_t *delaunay_edges, int64_t n_edges,
                             double *normals);

/* replaces: pcd.transform(T)   pointcloud_alignment.py:42  (rotate_only != 0 for normals) ; T row-major 4x4 */
int r3d_transform_points(r3d_ctx *ctx, const double *xyz, int64_t n, const double *T4x4, int32_t rotate_only, double *out);

/* replaces: o3d.pipelines.registration.registration_icp(source, target, max_dist, init, PointToPoint / PointToPlane,
 *           ICPConvergenceCriteria(relative_fitness, relative_rmse, max_iteration))   pointcloud_alignment.py:35-39,
 *           test/check2.py:151-154;  registration_generalized_icp(...)   test/GICP1.py:99-102 */
#define R3D_ICP_POINT_TO_POINT 0
#define R3D_ICP_POINT_TO_PLANE 1
#define R3D_ICP_GENERALIZED 2
typedef struct {
    int32_t mode;
    int32_t max_iteration;              /* Open3D default 30; pointcloud_alignment.py passes 100 */
    double max_correspondence_distance; /* 1-NN accepted iff distance < this */
    double relative_fitness;            /* 1e-6 */
    double relative_rmse;               /* 1e-6 */
    double gicp_epsilon;                /* 1e-3; <= 0 selects the default */
} r3d_icp_params;
typedef struct {
    int32_t iterations; /* ComputeTransformation calls performed */
    int32_t converged;  /* stopped by the relative criteria (not an error when 0) */
    int64_t correspondences;
    double fitness;     /* correspondences / source points */
    double inlier_rmse; /* Euclidean, over correspondences */
    double setup_ms;    /* host wall time: uploads, target grid build, source sort */
    double loop_ms;     /* host wall time of the evaluate / solve loop (iterations + 1 evaluations) */
} r3d_icp_stats;
/* src_normals: GICP only (covariances C = I - (1-eps) n n^T, as Open3D derives them from normals);
 * tgt_normals: point-to-plane and GICP.  init4x4 may be NULL (identity).  T4x4: row-major result.
 * The loop (correspondences, statistics, convergence test, update) runs on the device; the host enqueues evaluations in batches
 * and reads the loop state once per batch, so `iterations` / `converged` are exactly those of the sequential loop. */
int r3d_icp(r3d_ctx *ctx, const r3d_icp_params *p, const double *src, int64_t ns, const double *src_normals, const double *tgt,
            int64_t nt, const double *tgt_normals, const double *init4x4, double *T4x4, r3d_icp_stats *stats);

/* replaces: the whole body of PointCloudAlignment.align_point_clouds (pointcloud_alignment.py:6-43; caller main.py:48) in ONE
 * call, device-resident between the stages: voxel_down_sample(voxel_size) of both clouds (:22-23) ->
 * estimate_normals(KDTreeSearchParamHybrid(normal_radius, normal_max_nn)) on both (:27-28) -> registration (:35-39) ->
 * source.transform(T) (:42).  Returns what the reference returns: the DOWN-SAMPLED, transformed source (points, colours
 * averaged per voxel, rotated normals).  out arrays need room for ns triplets; out_colors only with src_colors; out_normals
 * only when normal_max_nn > 0.  voxel_size <= 0 skips the down-sampling; stats->setup_ms then covers everything before the
 * ICP loop.  Same kernels as the separate entry points: identical results to chaining those. */
typedef struct r3d_align_params {
    r3d_icp_params icp;
    double voxel_size;
    double normal_radius;
    int32_t normal_max_nn;
    int32_t reserved;
} r3d_align_params;
int r3d_align_point_clouds(r3d_ctx *ctx, const r3d_align_params *p, const double *src, const double *src_colors, int64_t ns,
                           const double *tgt, int64_t nt, const double *init4x4, double *out_xyz, double *out_colors,
                           double *out_normals, int64_t *out_n, double *T4x4, r3d_icp_stats *stats);

/* ---- resident scan-loop model: the growing `combined_pcd` of main.py:28,34-54 and test/GICP1.py:134-155 kept in HBM -----------
 * The reference passes the whole accumulated cloud to align_point_clouds for every frame (main.py:48), which down-samples it
 * again (pointcloud_alignment.py:23), and appends the aligned frame (main.py:49).  An r3d_model owns device buffers for the
 * model's points / colours / normals: per frame only the frame goes up and the 4x4 + statistics come down.  Each call runs the
 * kernels of the one-shot entry points on the same values in the same order (the model is re-voxelised from its resident
 * points), so the results equal those of r3d_align_point_clouds / r3d_icp + `+=` on host clouds.  Attributes follow the legacy
 * operator+=: colours / normals survive an append only if the model is empty or carries them AND the appended cloud does. */
typedef struct r3d_model r3d_model;
int r3d_model_create(r3d_ctx *ctx, r3d_model **out);
void r3d_model_destroy(r3d_model *m);
int r3d_model_clear(r3d_model *m);
int r3d_model_size(r3d_model *m, int64_t *n, int32_t *has_colors, int32_t *has_normals);
/* The model keeps its legacy voxel grid (pointcloud_alignment.py:23 re-down-samples the target every frame, main.py:48) as a resident
 * table that a new frame is merged into while the model's minimum corner -- the grid's origin -- stays where it is; it is rebuilt
 * from all points otherwise (same means bit for bit either way).  Diagnostics: voxels in the table, full rebuilds and incremental
 * updates so far.  Any pointer may be NULL. */
int r3d_model_voxel_table_stats(r3d_model *m, int64_t *voxels, int32_t *rebuilds, int32_t *updates);
/* combined.points = frame.points ... (main.py:42-45) and `combined += cloud` for host clouds; colors / normals may be NULL */
int r3d_model_append(r3d_model *m, const double *xyz, const double *colors, const double *normals, int64_t n);
/* main.py:48-49: aligned = align_point_clouds(frame, combined, threshold, voxel_size, max_iter); combined += aligned.
 * Normals are estimated only where they can reach a result (target normals for the plane / GICP estimators, source normals
 * for GICP or when the model carries normals): the point-to-point estimator never reads them and += drops them otherwise.
 * *appended (may be NULL) = points added (the down-sampled frame). */
int r3d_model_align_append(r3d_model *m, const r3d_align_params *p, const double *src, const double *src_colors, int64_t ns,
                           double *T4x4, r3d_icp_stats *stats, int64_t *appended);
/* replaces: o3d.geometry.PointCloud.create_from_rgbd_image(o3d.geometry.RGBDImage.create_from_color_and_depth(color, depth,
 *           depth_scale, depth_trunc, convert_rgb_to_intensity=False), intrinsic) + the flip transform   test/check84.py:155-159,
 * 172-178 -- how every recorded frame of test/output84 became a cloud (verified on all 163 frames, so this
 * entry point is PINNED by the reference's own PLY files).  z = float32(raw) / float32(depth_scale); z > depth_trunc or z == 0
 * dropped; x = (u - ppx) z / fx, y = (v - ppy) z / fy in float64; flip_yz != 0: (x, -y, -z).  Row-major pixel order.  depth:
 * uint16 [h][stride]; color (may be NULL): uint8 [h][color_stride bytes] with 3 channels per pixel, written as channel / 255 in
 * the order given.  Output arrays need room for w*h entries; out_pixel (may be NULL) = linear pixel index of every point. */
typedef struct r3d_depth_camera {
    double fx, fy, ppx, ppy;   /* camera_intrinsic.json */
    double depth_scale;        /* raw units per metre as the script passes it: 1.0 / sensor depth scale (cast to float32 inside) */
    double depth_trunc;        /* metres; 3.0 in the reference */
    int32_t flip_yz;           /* 1: the (x, -y, -z) flip of check84.py:172-178 */
    int32_t reserved;
} r3d_depth_camera;
int r3d_backproject_depth(r3d_ctx *ctx, const uint16_t *depth, int32_t w, int32_t h, int32_t stride, const r3d_depth_camera *cam,
                          const uint8_t *color, int32_t color_stride, double *out_xyz, double *out_colors, int32_t *out_pixel, int64_t *out_n);
/* the first frame / a later frame of the scanning loop given as a DEPTH IMAGE: back-projection on the device, then exactly
 * r3d_model_append / r3d_model_align_append (0.6 MB goes up instead of 6.8 MB of float64 points).  *frame_points = valid pixels;
 * a frame without any (a failed capture, main.py:53-54) changes nothing: identity, zeroed statistics, *appended = 0. */
int r3d_model_append_depth(r3d_model *m, const uint16_t *depth, int32_t w, int32_t h, int32_t stride, const r3d_depth_camera *cam,
                           const uint8_t *color, int32_t color_stride, int64_t *appended);
int r3d_model_align_append_depth(r3d_model *m, const r3d_align_params *p, const uint16_t *depth, int32_t w, int32_t h, int32_t stride,
                                 const r3d_depth_camera *cam, const uint8_t *color, int32_t color_stride, double *T4x4, r3d_icp_stats *stats,
                                 int64_t *frame_points, int64_t *appended);
/* test/GICP1.py:145-146: aligned = align_point_clouds(frame, combined) (registration of the frame against the WHOLE model with
 * its normals, :99-103, from identity); combined += aligned.  src_colors / src_normals may be NULL where the mode allows. */
int r3d_model_register_append(r3d_model *m, const r3d_icp_params *p, const double *src, const double *src_colors, const double *src_normals,
                              int64_t ns, double *T4x4, r3d_icp_stats *stats);
/* test/GICP1.py:148: combined.estimate_normals(KDTreeSearchParamHybrid(radius, max_nn)); existing normals keep their orientation */
int r3d_model_estimate_normals(r3d_model *m, double radius, int32_t max_nn);
/* host arrays with room for r3d_model_size() triplets; colors / normals may be NULL (and are left alone when the model has none) */
int r3d_model_download(r3d_model *m, double *xyz, double *colors, double *normals);

/* ---- device-pointer forms (clouds that stay in HBM: the multi-view exchange of BASELINE config C5) -----------------------------
 * r3d_disparity_to_cloud_resident = r3d_disparity_to_cloud_dev with DEVICE output arrays (capacity triplets each): the cloud is
 * left in the caller's device buffers, ordered on the context stream (r3d_set_stream lets that be the caller's / torch's
 * stream, so an RCCL collective enqueued there afterwards sees it); only *out_n comes back to the host.
 * r3d_icp_dev = r3d_icp on device clouds (returns when the loop has finished).  r3d_transform_points_dev = r3d_transform_points
 * on device arrays, asynchronous on the context stream; d_out may equal d_xyz. */
int r3d_disparity_to_cloud_resident(r3d_ctx *ctx, const int16_t *d_disp, int32_t w, int32_t h, const double *Q4x4, int32_t min_valid_x16,
                                    double max_depth, const double *pose4x4, double voxel, double normal_radius, int32_t max_nn,
                                    int64_t capacity, double *d_out_xyz, double *d_out_normals, int64_t *out_n);
int r3d_icp_dev(r3d_ctx *ctx, const r3d_icp_params *p, const double *d_src, int64_t ns, const double *d_src_normals, const double *d_tgt,
                int64_t nt, const double *d_tgt_normals, const double *init4x4, double *T4x4, r3d_icp_stats *stats);
int r3d_transform_points_dev(r3d_ctx *ctx, const double *d_xyz, int64_t n, const double *T4x4, int32_t rotate_only, double *d_out);
/* n_blocks device arrays of triplets, block b moved by T4x4s[16 b .. 16 b + 15] (rotation only where rotate_only[b] != 0;
 * rotate_only may be NULL), in one launch per 16 blocks: the fuse step of the multi-view exchange (every gathered view's points
 * and normals into view 0's frame before mesh_reconstruction.py:22-37).  Pointer tables and counts are HOST arrays; same
 * arithmetic as r3d_transform_points_dev; asynchronous on the context stream; d_out[b] may equal d_xyz[b]. */
int r3d_transform_blocks_dev(r3d_ctx *ctx, int32_t n_blocks, const double *const *d_xyz, const int64_t *counts, const double *T4x4s,
                             const int32_t *rotate_only, double *const *d_out);


/* ---- per-frame stages either side of the matcher in Calib_depth/depth*.py --------------
 * OpenCV is a dependency of the reference that is absent here, and the reference records no output of these calls:
 * parity of this group is UNPINNED (restated from OpenCV 4.x's published algorithms; see oracle/prepost_oracle.py). */

/* replaces: cv2.initUndistortRectifyMap(mtx, dist, R, P, image_size, cv2.CV_16SC2)   depth2.py:125-128 (depth1.py:208-211)
 * camera3x3 row-major; dist: n_dist in {0,4,5,8,12,14} coefficients k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 tx ty (tilt must be
 * 0); R3x3 may be NULL (identity); new_camera: 3 x new_camera_cols (3 or 4) row-major, the projection matrix P1/P2 as
 * stored in the calibration file.  Host outputs: map1 int16 [h][w][2] (integer source x, y), map2 uint16 [h][w]
 * (5+5 fraction bits, INTER_TAB_SIZE = 32).  fp64, running sums along each row like the original's scalar loop. */
int r3d_init_undistort_rectify_map(r3d_ctx *ctx, const double *camera3x3, const double *dist, int32_t n_dist, const double *R3x3,
                                   const double *new_camera, int32_t new_camera_cols, int32_t w, int32_t h, int16_t *map1,
                                   uint16_t *map2);

/* replaces: cv2.remap(frame, map_x, map_y, cv2.INTER_LINEAR)   depth2.py:243-244
 * uint8 source with cn = 1, 3 or 4 interleaved channels, fixed-point maps as above, BORDER_CONSTANT(border_value):
 * dst = (sum of 4 taps x 15-bit table weights + 2^14) >> 15.  gray (may be NULL, needs cn >= 3): the BGR2GRAY image of the
 * remapped frame written by the same kernel (depth2.py:247-248 converts the rectified frame right away). */
int r3d_remap_u8(r3d_ctx *ctx, const uint8_t *src, int32_t sw, int32_t sh, int32_t sstride, int32_t cn, const int16_t *map1,
                 const uint16_t *map2, int32_t dw, int32_t dh, int32_t border_value, uint8_t *dst, uint8_t *gray);
int r3d_remap_u8_dev(r3d_ctx *ctx, const uint8_t *d_src, int32_t sw, int32_t sh, int32_t sstride, int32_t cn,
                     const int16_t *d_map1, const uint16_t *d_map2, int32_t dw, int32_t dh, int32_t border_value, uint8_t *d_dst,
                     uint8_t *d_gray);

/* replaces: cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)   depth2.py:247-248:  (B*1868 + G*9617 + R*4899 + 2^13) >> 14 */
int r3d_bgr2gray(r3d_ctx *ctx, const uint8_t *bgr, int32_t w, int32_t h, int32_t stride, int32_t cn, uint8_t *gray);
int r3d_bgr2gray_dev(r3d_ctx *ctx, const uint8_t *d_bgr, int32_t w, int32_t h, int32_t stride, int32_t cn, uint8_t *d_gray);

/* replaces: wls_filter = cv2.ximgproc.createDisparityWLSFilter(matcher_left=stereo_matcher); setLambda(8000);
 * setSigmaColor(1.5)   depth2.py:164-166;   wls_filter.filter(disparity_left, gray_left, None, disparity_right)   :255 */
typedef struct r3d_wls_params {
    double lambda;                 /* setLambda */
    double sigma_color;            /* setSigmaColor */
    double lambda_attenuation;     /* 0.25 */
    double discontinuity_roll_off; /* 0.001 */
    int32_t min_disparity;         /* of the LEFT matcher: ROI = columns [max(0,minD+D), w - max(0,-minD)) */
    int32_t num_disparities;
    int32_t discontinuity_radius;  /* ceil(0.5*blockSize) for an SGBM matcher */
    int32_t lrc_thresh;            /* 24 (1.5 px in x16 units) */
    int32_t num_iter;              /* 3 */
    int32_t solver;                /* R3D_WLS_SOLVER_*: how the smoother's tridiagonal systems are solved */
} r3d_wls_params;
/* CONTRACT of the two solvers (tests/test_prepost_gpu.py asserts it up to 3264x2448, D = 128):
 *   R3D_WLS_SOLVER_SEQUENTIAL  one Thomas sweep per line in the float32 operation order of the CPU original as restated in
 *       oracle/prepost_oracle.py: confidence map and filtered map BIT-EXACT against that restatement.  39 waves on a
 *       1024-SIMD chip: 7.8 ms per 8 MP frame.
 *   R3D_WLS_SOLVER_PARTITIONED (default)  solves the SAME tridiagonal systems block-parallel (31-unknown blocks + Schur
 *       complement over the separators): exact in exact arithmetic, a different float32 rounding order.  TOLERANCE: the
 *       confidence map is bit-exact; the filtered int16 map (x16 fixed point) differs from the sequential solver's by AT MOST
 *       1 LSB (1/16 px) on FEWER THAN 0.1 % of the pixels (a value that rounds half-to-even differently), and not at all
 *       outside the ROI.  0.69 ms per 8 MP frame.  Callers that need the bit-exact map select the sequential solver. */
#define R3D_WLS_SOLVER_PARTITIONED 0
#define R3D_WLS_SOLVER_SEQUENTIAL 1
/* disp_left / disp_right: int16 x16 maps of the left and the right matcher (the right one holds negative values);
 * guide: uint8 left view with guide_cn = 1 or 3 channels.  out: int16 x16, 16*(minD-1) outside the ROI.
 * confidence (may be NULL): float [h][w] map in [0,255] (getConfidenceMap()). */
int r3d_wls_filter(r3d_ctx *ctx, const r3d_wls_params *p, const int16_t *disp_left, const int16_t *disp_right, const uint8_t *guide,
                   int32_t guide_cn, int32_t guide_stride, int32_t w, int32_t h, int16_t *out, float *confidence);
int r3d_wls_filter_dev(r3d_ctx *ctx, const r3d_wls_params *p, const int16_t *d_disp_left, const int16_t *d_disp_right,
                       const uint8_t *d_guide, int32_t guide_cn, int32_t guide_stride, int32_t w, int32_t h, int16_t *d_out,
                       float *d_confidence);

/* replaces: cv2.normalize(filtered_disparity, None, 0, 255, cv2.NORM_MINMAX)   depth2.py:256 (int16 in, int16 out) */
int r3d_normalize_minmax_s16(r3d_ctx *ctx, const int16_t *src, int64_t n, double alpha, double beta, int16_t *dst);
int r3d_normalize_minmax_s16_dev(r3d_ctx *ctx, const int16_t *d_src, int64_t n, double alpha, double beta, int16_t *d_dst);

#ifdef __cplusplus
}
#endif
#endif /* R3D_H */
