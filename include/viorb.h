/* include/viorb.h — C ABI of libviorb_hip.so: the MI355X (gfx950) drop-in for VIORB's per-frame
 * visual-inertial front-end. Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns an int status: VIORB_OK (0) or a negative VIORB_ERR_* code; counts come
 *     back through out-parameters; nothing throws across the ABI and nothing is printed on the hot
 *     path (the reference has no error codes: empty image -> silent return, src/ORBextractor.cc:1046).
 *   - caller owns every output buffer and passes its capacity; the library owns device memory inside
 *     handles. One handle = one HIP stream of work; different handles may be used concurrently from
 *     different threads, one handle must not be (same contract as one ORBextractor instance per camera,
 *     reference src/Frame.cc:258-261).
 *   - "*_device" entry points take device pointers + a hipStream_t (passed as void*) and only enqueue
 *     work; the host-buffer entry points are the literal drop-ins and include the PCIe copies.
 *   - there is NO CPU fallback: without a HIP device every compute entry point returns
 *     VIORB_ERR_NO_DEVICE.
 */
#ifndef VIORB_H
#define VIORB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIORB_OK                 0
#define VIORB_ERR_INVALID_ARG   -1
#define VIORB_ERR_NO_DEVICE     -2
#define VIORB_ERR_HIP           -3   /* a HIP runtime call failed; see viorb_last_error() */
#define VIORB_ERR_CAPACITY      -4   /* an internal or caller capacity was exceeded; outputs truncated */
#define VIORB_ERR_UNSUPPORTED   -5

/* ABI/version probe. */
int viorb_abi_version(void);
/* Human-readable description of the last error on the calling thread (never NULL). */
const char* viorb_last_error(void);
/* Number of visible HIP devices (0 when there is none); does not create a context. */
int viorb_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * ORB extractor — replaces ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-111,
 * src/ORBextractor.cc:410-470 ctor, :1043-1105 operator()).
 * ---------------------------------------------------------------------------------------------- */

/* Layout-identical to cv::KeyPoint (28 bytes): what operator() fills (src/ORBextractor.cc:837-847,
 * :1095-1101). class_id is always -1. */
typedef struct viorb_keypoint {
    float x, y;        /* pt, in level-0 pixel units (level coords * scale factor) */
    float size;        /* 31 * scale[octave], truncated to int */
    float angle;       /* degrees [0,360), intensity-centroid orientation */
    float response;    /* FAST score */
    int32_t octave;
    int32_t class_id;
} viorb_keypoint;

/* Mirrors the five ctor arguments (reference include/ORBextractor.h:51-52; YAML keys
 * ORBextractor.{nFeatures,scaleFactor,nLevels,iniThFAST,minThFAST}). */
typedef struct viorb_extractor_params {
    int32_t nfeatures;
    float   scale_factor;
    int32_t nlevels;       /* 1..16 */
    int32_t ini_th_fast;
    int32_t min_th_fast;
} viorb_extractor_params;

typedef struct viorb_extractor viorb_extractor;   /* opaque */

/* Create an extractor able to process up to max_batch same-sized images per call on HIP device
 * `device`. Device buffers are sized lazily for the first image size seen and re-sized when it
 * changes. max_batch = 1 gives the literal per-camera object of the reference.
 * Limits (each refused with an error, never served with a wrong result): a per-level quota (mnFeaturesPerLevel) above 13 000; a FAST
 * cell wider or taller than 255 px; see also viorb_extractor_max_keypoints. There is no limit on the candidates of a level. A
 * per-level quota above ~1100 (e.g. 1700 features on ONE level; any nfeatures up to ~5000 with the reference's 8 levels of 1.2 stays
 * below) moves the quadtree's node lists from LDS to global memory: same output, a slower quadtree. */
int viorb_extractor_create(const viorb_extractor_params* params, int max_batch, int device,
                           viorb_extractor** out);
int viorb_extractor_destroy(viorb_extractor* h);

/* Scale tables (GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares, reference include/ORBextractor.h:66-84) and per-level feature quotas
 * (mnFeaturesPerLevel). Any pointer may be NULL. Arrays hold nlevels entries. */
int viorb_extractor_tables(const viorb_extractor* h, float* scale, float* inv_scale, float* sigma2,
                           float* inv_sigma2, int32_t* features_per_level);

/* Upper bound on the keypoints of one image: size output buffers with it. The quadtree keeps at most quota + 2 keypoints on a level —
 * except on a level whose roots (round(width / height) of the bordered level) outnumber a quarter of its quota, where the unchecked first
 * round keeps up to 4 x roots (src/ORBextractor.cc:514-552): a panorama-shaped image with a very small nfeatures (no camera configuration
 * of the reference's settings files comes near it: roots <= 3, quotas >= 60). viorb_extractor_max_keypoints is the sum of quota + 2, what
 * holds for every ordinary size; viorb_extractor_max_keypoints_for(width, height) is the exact bound for that image size (the same number
 * unless the image is of that kind) and the pitch of the handle's device results for it. A caller buffer that is too small makes the
 * extraction return VIORB_ERR_CAPACITY (never a silently shortened list). */
int viorb_extractor_max_keypoints(const viorb_extractor* h, int* cap);
int viorb_extractor_max_keypoints_for(const viorb_extractor* h, int width, int height, int* cap);
/* Images per k_fast_cells launch of a batched call (the FAST stage of a batch goes out as several launches over sub-ranges of the
 * batch; with a kernel selection the profiler times one of them per call, in rotation). For bench.py's bytes-per-launch figure. */
int viorb_extractor_fast_launch_images(int batch);

/* Drop-in for ORBextractor::operator()(image, mask [ignored], keypoints, descriptors) with host
 * buffers: 8-bit single-channel image, `stride` bytes per row. Writes min(*n, cap) keypoints and
 * 32-byte descriptors (row i of `desc` belongs to kps[i]); *n = number found. An empty image
 * (img NULL or w/h <= 0) returns VIORB_OK with *n = 0, like the reference's silent return. */
int viorb_extract(viorb_extractor* h, const uint8_t* img, int width, int height, int stride,
                  viorb_keypoint* kps, uint8_t* desc, int cap, int* n);

/* Batched, device-resident form: `d_images` points to `batch` images in device memory, image b at
 * d_images + b*image_pitch_bytes, rows `stride` bytes apart. Enqueues the whole extraction on
 * `stream` (a hipStream_t; NULL = the default stream) and returns without synchronising. Results
 * stay on the device inside the handle until the next call (see viorb_extractor_results_device). */
int viorb_extract_batch_device(viorb_extractor* h, const uint8_t* d_images, int batch, int width,
                               int height, int stride, size_t image_pitch_bytes, void* stream);

/* Device pointers to the results of the last batched call: d_kps[b*cap + i], d_desc[(b*cap + i)*32],
 * d_count[b] (keypoints found in image b), d_status[b] (VIORB_OK or VIORB_ERR_CAPACITY per image).
 * Valid once the stream the extraction was enqueued on has reached that point. */
int viorb_extractor_results_device(const viorb_extractor* h, const viorb_keypoint** d_kps,
                                   const uint8_t** d_desc, const int32_t** d_count,
                                   const int32_t** d_status, int* cap);

/* Synchronise and copy image b's results of the last batched call to host buffers. */
int viorb_extractor_download(viorb_extractor* h, int b, viorb_keypoint* kps, uint8_t* desc, int cap,
                             int* n);

/* Pyramid access — replaces the public member mvImagePyramid that Frame::ComputeStereoMatches reads
 * (reference include/ORBextractor.h:86, src/Frame.cc:653,743-760). Levels are stored WITHOUT the
 * reference's 19-px border (never read on this path). blurred != 0 selects the 7x7 sigma-2 blurred
 * plane the descriptors were sampled from. Pointers stay valid until the next extract call. */
int viorb_extractor_level_device(const viorb_extractor* h, int b, int level, int blurred,
                                 const uint8_t** d_ptr, int* width, int* height, int* stride);
/* Synchronise and copy one level to a host buffer of width*height bytes (rows packed). */
int viorb_extractor_level_download(viorb_extractor* h, int b, int level, int blurred, uint8_t* dst,
                                   int* width, int* height);

/* Frame::ComputeStereoMatches (reference src/Frame.cc:646-820) on the features and pyramids two extractor handles hold
 * after extraction (the reference runs two ORBextractor instances, src/Frame.cc:258-261): image left_index + p of L is
 * matched against image right_index + p of R (L and R may be the same batched handle). bf = Camera.bf, fx = Camera.fx.
 * Outputs per left keypoint: d_uright[p][cap], d_depth[p][cap] (-1 where unmatched, cap = viorb_extractor_max_keypoints)
 * and d_nmatched[p]. The sort keys of the right image's rows and the per-keypoint work arrays live in LDS for up to ~4000 features per
 * image (KITTI's setting is 2000) and in global memory beyond that (same result, slower). */
int viorb_stereo_match_device(const viorb_extractor* L, int left_index, const viorb_extractor* R, int right_index, int pairs,
                              float bf, float fx, float* d_uright, float* d_depth, int32_t* d_nmatched, void* stream);
/* Host-buffer form for the literal two-instance use: image 0 of both handles; uright/depth hold min(cap, max_keypoints). */
int viorb_stereo_match(viorb_extractor* L, viorb_extractor* R, float bf, float fx, float* uright, float* depth, int cap,
                       int* nmatched);

/* Stage introspection for parity tests: FAST candidates of one level before the quadtree
 * (ComputeKeyPointsOctTree's vToDistributeKeys, src/ORBextractor.cc:765-830), in the reference's
 * push order, as (x, y, response) int32 triples relative to the (16,16) border origin.
 * which = 0: candidates; which = 1: keypoints kept by DistributeOctTree, level coordinates. */
int viorb_extractor_debug_level_points(viorb_extractor* h, int b, int level, int which, int32_t* xyr,
                                       int cap, int* n);

/* ------------------------------------------------------------------------------------------------
 * Per-frame tracking front-end behind the extractor — replaces, for B independent camera streams at a
 * time, the calls Tracking::TrackWithIMU makes per frame (reference src/Tracking.cc:412-534):
 *   Frame::AssignFeaturesToGrid            (src/Frame.cc:410-425)      viorb_frontend_grid_device
 *   GetIMUPreIntSinceLastFrame + updateNS  (src/Frame.cc:41-110)       viorb_frontend_imu_predict_device
 *   ORBmatcher::SearchByProjection(F,F)    (src/ORBmatcher.cc:1328)    viorb_frontend_search_projection_device
 *   Optimizer::PoseOptimization (VI)       (src/Optimizer.cc:323,789)  viorb_frontend_pose_opt_device
 * All arrays are device pointers, row b of every array belongs to stream b; `cap` is the per-stream
 * keypoint capacity the handle was created with. Flat float64 layouts:
 *   navstate[22] = P3 V3 q4(x,y,z,w) bg3 ba3 dbg3 dba3        (NavState, src/IMU/NavState.h)
 *   preint[142]  = dP3 dV3 dR9 JPg9 JPa9 JVg9 JVa9 JRg9 cov81 dt  (IMUPreintegrator, row-major blocks)
 *   cam[16]      = fx fy cx cy Rbc9 Pbc3                        (Frame intrinsics + ConfigParam Tbc)
 *   imu[n][7]    = gyro3 acc3 t                                 (IMUData)
 *   obs[k][6]    = Pw3 u v invSigma2                            (one EdgeNavStatePVRPointXYZOnlyPose)
 * ---------------------------------------------------------------------------------------------- */
typedef struct viorb_frontend viorb_frontend;    /* opaque: scratch buffers for max_batch x cap */

typedef struct viorb_frontend_config {
    float   min_x, max_x, min_y, max_y;   /* Frame::mnMinX.. (image bounds after undistortion) */
    float   fx, fy, cx, cy;               /* pinhole intrinsics (float, as Frame stores them) */
    double  cam[16];                      /* fx fy cx cy Rbc9 Pbc3 in double for the solver */
    double  gravity[3];                   /* gw */
    float   scale_factors[16];            /* mvScaleFactors */
    float   inv_level_sigma2[16];         /* mvInvLevelSigma2 */
    int32_t nlevels;
    int32_t check_orientation;            /* ORBmatcher(nnratio, checkOri) second argument */
    double  gyr_meas_cov, acc_meas_cov;   /* IMUData::_gyrMeasCov / _accMeasCov diagonal; <= 0 selects */
    double  acc_bias_rw2;                 /* the reference constants (src/IMU/imudata.cpp:31-41)       */
    float   dist_coef[5];                 /* Frame::mDistCoef = k1 k2 p1 p2 k3 (Tracking.cc:105-117). dist_coef[0] == 0: the camera is
                                             treated as undistorted, mvKeysUn = mvKeys (Frame.cc:586-590) */
    int32_t reserved0;
} viorb_frontend_config;

/* cap = keypoints per frame the handle's arrays are pitched for (viorb_extractor_max_keypoints of the extractor that feeds it). The
 * projection and local-point searches keep a frame's keypoints and their work arrays in LDS up to cap = viorb_frontend_search_capacity()
 * (~4900; the reference's settings files use 1000-2000 features): above ~2460 they give up their LDS cache of the first candidates of every
 * point and take all candidates from the global list, above the capacity they keep the work arrays in global memory as well — same
 * result, slower each time. cap <= 65535 (16-bit keypoint indices). The global-memory work arrays of the over-size forms (searches,
 * SearchByBoW above ~7100 keypoints, stereo association above ~4000 features) are ONE scratch per handle (per calling thread for the handle-less
 * entry points): over-size calls on one handle / thread must be issued on one stream at a time; the LDS forms have no such restriction. */
int viorb_frontend_create(const viorb_frontend_config* cfg, int max_batch, int cap, int device, viorb_frontend** out);
int viorb_frontend_search_capacity(void);      /* largest cap whose search work arrays fit LDS (beyond it: global memory, slower) */
int viorb_frontend_destroy(viorb_frontend* h);

/* Frame::UndistortKeyPoints (reference src/Frame.cc:584-614) for a batch: kps_un[b][i] = kps[b][i] with pt replaced by
 * cv::undistortPoints(pt, mK, mDistCoef, R = Mat(), P = mK) — OpenCV 2.4: double inside, five fixed-point iterations of the
 * radial-tangential model, float out. Everything downstream of the extractor (grid, searches, edges) reads mvKeysUn. With
 * cfg.dist_coef[0] == 0 the records are copied unchanged. kps_un may alias kps. */
int viorb_frontend_undistort_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count, int batch,
                                    viorb_keypoint* kps_un, void* stream);
/* Host-buffer forms (run on the current HIP device): cv::undistortPoints on n float pixel pairs, intr4 = fx fy cx cy (Frame::mK),
 * dist5 = k1 k2 p1 p2 k3; and Frame::ComputeImageBounds (src/Frame.cc:616-644): bounds4 = mnMinX mnMaxX mnMinY mnMaxY from the four
 * undistorted image corners (0, width, 0, height when dist5[0] == 0). */
int viorb_undistort_points(const float* xy, int n, const float* intr4, const float* dist5, float* xy_out);
int viorb_image_bounds(int width, int height, const float* intr4, const float* dist5, float* bounds4);

/* cell_start[b][64*48+1], cell_idx[b][cap]: CSR of Frame::mGrid in storage order cell = ix*48 + iy. */
int viorb_frontend_grid_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count, int batch,
                               int32_t* cell_start, int32_t* cell_idx, void* stream);

/* Pre-integrate imu[b][n_imu][7] between t_last[b] and t_cur[b] with the biases of last_ns[b], predict
 * the current NavState (Converter::updateNS) and the float camera pose pose12[b] = Rcw(9) tcw(3). */
int viorb_frontend_imu_predict_device(viorb_frontend* h, const double* imu, int n_imu, const double* t_last,
                                      const double* t_cur, const double* last_ns, int batch, double* preint,
                                      double* cur_ns, float* pose12, void* stream);

/* SearchByProjection(CurrentFrame, LastFrame, th, bMono = true). last_flags: bit0 map point present,
 * bit1 outlier, bit2 the map point has observations. cur_match[b][i2] = index of the last-frame point
 * given to current keypoint i2 or -1 (starts empty, as Tracking clears mvpMapPoints first);
 * nmatches[b] = the function's return value; status[b] = VIORB_OK / VIORB_ERR_CAPACITY. */
int viorb_frontend_search_projection_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc,
                                            const int32_t* cur_count, const int32_t* cell_start, const int32_t* cell_idx,
                                            const float* pose12, const viorb_keypoint* last_kps, const int32_t* last_count,
                                            const uint8_t* last_flags, const float* last_Pw, const uint8_t* last_desc,
                                            float th, int batch, int32_t* cur_match, int32_t* nmatches, int32_t* status,
                                            void* stream);
/* The same search run again only for the streams whose nmatches[b] is below retry_below (the others keep their results):
 * TrackWithIMU's "if(nmatches<20) ... SearchByProjection(..., 2*th, ...)" (reference src/Tracking.cc:440-444) for a batch —
 * call with th = 2 * th and retry_below = 20 right after the first search. retry_below <= 0: every stream is searched. */
int viorb_frontend_search_projection_retry_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc,
                                            const int32_t* cur_count, const int32_t* cell_start, const int32_t* cell_idx,
                                            const float* pose12, const viorb_keypoint* last_kps, const int32_t* last_count,
                                            const uint8_t* last_flags, const float* last_Pw, const uint8_t* last_desc,
                                            float th, int retry_below, int batch, int32_t* cur_match, int32_t* nmatches, int32_t* status,
                                            void* stream);

/* The stereo / RGB-D branch of the same search (bMono == false; reference src/ORBmatcher.cc:1346-1349, 1385-1410): cur_uright[b][cap] =
 * CurrentFrame.mvuRight (<= 0: no right match), last_pose12[b] = LastFrame.mTcw, bf = mbf, mb = mb. When the camera has moved forward
 * (backward) along the last frame's optical axis by more than the baseline, the candidates come from the octaves >= (<=) the point's
 * octave instead of octave +- 1, and a candidate with a right coordinate must agree with the projected one within the window radius. */
int viorb_frontend_search_projection_stereo_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc,
                                                   const int32_t* cur_count, const float* cur_uright, const int32_t* cell_start,
                                                   const int32_t* cell_idx, const float* pose12, const float* last_pose12,
                                                   const viorb_keypoint* last_kps, const int32_t* last_count, const uint8_t* last_flags,
                                                   const float* last_Pw, const uint8_t* last_desc, float th, float bf, float mb,
                                                   int retry_below, int batch, int32_t* cur_match, int32_t* nmatches, int32_t* status,
                                                   void* stream);

/* Tracking::SearchLocalPoints (reference src/Tracking.cc:1904-1958): Frame::isInFrustum(pMP, 0.5) for every local map
 * point that is valid and not yet matched in this frame, then ORBmatcher::SearchByProjection(F, vpMapPoints, th)
 * (src/ORBmatcher.cc:45-129, ORBmatcher(nnratio)). pts_f[b][pcap][8] = Pw3 normal3 mfMinDistance mfMaxDistance,
 * pts_flags bit0 !isBad, bit1 mnLastFrameSeen == frame id (skip), bit2 Observations() > 0; pts_desc[b][pcap][32];
 * cur_owner_obs[b][cap] != 0 where the keypoint already holds a map point with observations. Outputs:
 * match[b][cap] = local point given to the keypoint by this call or -1, nmatches[b], and (optional)
 * frustum[b][pcap][5] = mbTrackInView mTrackProjX mTrackProjY mTrackViewCos mnTrackScaleLevel. */
int viorb_frontend_search_local_points_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc,
                                              const int32_t* cur_count, const int32_t* cell_start, const int32_t* cell_idx,
                                              const float* pose12, const float* pts_f, const uint8_t* pts_flags,
                                              const uint8_t* pts_desc, const int32_t* pts_count, int pcap, float th,
                                              float nnratio, const uint8_t* cur_owner_obs, int batch, int32_t* match,
                                              int32_t* nmatches, float* frustum, int32_t* status, void* stream);

/* The same search for a stereo / RGB-D frame (reference src/ORBmatcher.cc:91-97): cur_uright[b][cap] = F.mvuRight (<= 0: no right match), bf = F.mbf.
 * A candidate keypoint with a right coordinate is skipped when |mTrackProjXR - mvuRight| exceeds the window radius r * mvScaleFactors[level],
 * with mTrackProjXR = u - mbf * invz as Frame::isInFrustum stores it (src/Frame.cc:499). frustum_xr (may be NULL) [b][pcap] receives mTrackProjXR
 * (0 for a point that is not in view). KITTI-shaped TrackLocalMap goes through this entry. */
int viorb_frontend_search_local_points_stereo_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc,
                                                     const int32_t* cur_count, const float* cur_uright, float bf, const int32_t* cell_start,
                                                     const int32_t* cell_idx, const float* pose12, const float* pts_f, const uint8_t* pts_flags,
                                                     const uint8_t* pts_desc, const int32_t* pts_count, int pcap, float th, float nnratio,
                                                     const uint8_t* cur_owner_obs, int batch, int32_t* match, int32_t* nmatches, float* frustum,
                                                     float* frustum_xr, int32_t* status, void* stream);

/* Edge construction of PoseOptimization: one observation per matched keypoint, in keypoint order.
 * match[b][i] >= 0 selects point match_Pw[b][match[b][i]]. obs_index[b][k] = keypoint of obs k. */
int viorb_frontend_build_observations_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count,
                                             const int32_t* match, const float* match_Pw, int batch, double* obs,
                                             int32_t* obs_index, int32_t* n_obs, void* stream);

/* Optimizer::PoseOptimization with NavState edges. variant 0 = (Frame*, KeyFrame*): last is fixed, no
 * prior edge, obs_last ignored; variant 1 = (Frame*, Frame*): last is free, prior edge from prior_ns /
 * marg_cov_inv[b][144], reprojection edges of both frames. Outputs: out_ns (optimised current NavState),
 * out_last_ns (may be NULL), outlier_cur[b][cap] / outlier_last (mvbOutlier per observation),
 * info[b][4] = {return value nInitialCorrespondences - nBad, final robust chi2, LM iterations, 0},
 * marg_out[b][144] = mMargCovInv when compute_marg != 0. The solver keeps every edge's chi2 of its last evaluation (g2o's stored edge
 * errors) in a scratch of the handle: the solves of ONE handle go to one stream at a time (like its searches' work arrays). */
int viorb_frontend_pose_opt_device(viorb_frontend* h, int variant, int compute_marg, const double* cur_ns,
                                   const double* last_ns, const double* prior_ns, const double* marg_cov_inv,
                                   const double* preint, const double* obs_cur, const int32_t* n_cur,
                                   const double* obs_last, const int32_t* n_last, int batch, double* out_ns,
                                   double* out_last_ns, uint8_t* outlier_cur, uint8_t* outlier_last, double* marg_out,
                                   double* info, void* stream);

/* Shape of the visual-inertial pose solver's launches for the rest of the process: P problems per workgroup (advanced in lock step), WPP wavefronts
 * per problem; 0, 0 = chosen by batch size (default: 2, 2 from two problems per CU, 1, 4 / 1, 8 for small batches). A tuning / test hook: results are
 * the same up to the grouping of floating-point sums. */
int viorb_frontend_set_pose_shape(int problems_per_workgroup, int wavefronts_per_problem);

/* Optimizer::PoseOptimization(Frame*) — vision-only 6-DoF solve (reference src/Optimizer.cc:3749-3978) for a batch.
 * pose12[b] = Rcw(9) tcw(3) of pFrame->mTcw (float), obs7[b][cap][7] = Xw3 u v uRight invSigma2 in keypoint order
 * (uRight < 0: monocular edge, else stereo edge with baseline*fx = bf). Intrinsics are the handle's fx fy cx cy.
 * Outputs: out_pose12[b] (SetPose), outlier[b][cap] (mvbOutlier per observation), info[b][4] = {return value, final
 * robust chi2, LM iterations, 0}. Fewer than 3 observations: pose unchanged, return value 0. */
int viorb_frontend_pose_opt_se3_device(viorb_frontend* h, const float* pose12, const double* obs7, const int32_t* n_obs,
                                       double bf, int batch, float* out_pose12, uint8_t* outlier, double* info, void* stream);

/* ---- Tracking::TrackLocalMapWithIMU glue (reference src/Tracking.cc:228-346, 489-507) ----------------------------------------
 * viorb_frontend_discard_outliers_device: TrackWithIMU's "Discard outliers" loop (:489-507). outlier[b][k] / obs_index[b][k] /
 * n_obs[b] are PoseOptimization's outputs for the current frame; match[b][c] (index into the last frame's points) is set to -1
 * where the edge ended as an outlier. owner_obs[b][c] = 1 when keypoint c still holds a map point with Observations() > 0
 * (pt_flags bit 2 of that point) — the cur_owner_obs input of viorb_frontend_search_local_points_device; n_map[b] = nmatchesMap.
 * viorb_frontend_pose_from_navstate_device: Frame::UpdatePoseFromNS (src/Frame.cc:88-105), NavState -> float Tcw (pose12).
 * viorb_frontend_build_observations2_device: PoseOptimization's edge construction when the frame's map points come from the
 * last frame (match_a into Pw_a[b][cap][3]) or, where match_a < 0, from the local map (match_b into pts_b[b][stride_b][8],
 * the pts_f layout of the local-points search), in keypoint order. */
int viorb_frontend_discard_outliers_device(viorb_frontend* h, int32_t* match, const int32_t* obs_index, const uint8_t* outlier,
                                           const int32_t* n_obs, const uint8_t* pt_flags, int batch, uint8_t* owner_obs,
                                           int32_t* n_map, void* stream);
int viorb_frontend_pose_from_navstate_device(viorb_frontend* h, const double* ns, int batch, float* pose12, void* stream);
int viorb_frontend_build_observations2_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count,
                                              const int32_t* match_a, const float* Pw_a, const int32_t* match_b,
                                              const float* pts_b, int stride_b, int batch, double* obs, int32_t* obs_index,
                                              int32_t* n_obs, void* stream);

/* Workload support (no reference counterpart): the MapPoint fields Frame::isInFrustum reads (normal, min / max distance,
 * MapPoint::UpdateNormalAndDepth with one observation) for the points viorb_synth_plane_points_device created from a frame:
 * pts_f[b][cap][8] = Pw3 normal3 minDist maxDist. */
int viorb_synth_local_points_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count, const double* pose12,
                                    const float* Pw, int batch, float* pts_f, void* stream);

/* Workload support for bench.py / tests (no reference counterpart): map points of the synthetic plane
 * world (viorb_amd/synth.py) for all keypoints of a frame; pose12 = Rcw(9) tcw(3) in double per stream.
 * Writes Pw[b][cap][3] and flags[b][cap] = 1|4 (map point with observations), 0 beyond count[b]. */
/* "mLastFrame = Frame(mCurrentFrame)" for the batched harness in one launch: shifts the local-map tables ([b][local_frames][cap]:
 * pts_f, descriptors, flags; slot 0 = newest) by one frame and puts the outgoing last frame's points (last_pts_f, last_desc,
 * last_flags) into slot 0 when shift_local != 0, copies the current frame's keypoints / descriptors / count over the last frame's,
 * and carries ns_src -> last_ns and prior_ns, t_src -> t_last, marg_src -> marg_dst (both NULL: skipped). loc_* may be NULL. */
int viorb_frontend_roll_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const int32_t* cur_count,
                               viorb_keypoint* last_kps, uint8_t* last_desc, int32_t* last_count, const float* last_pts_f,
                               const uint8_t* last_flags, float* loc_pts_f, uint8_t* loc_desc, uint8_t* loc_flags,
                               int local_frames, int shift_local, const double* ns_src, double* last_ns, double* prior_ns,
                               const double* t_src, double* t_last, const double* marg_src, double* marg_dst, int batch,
                               void* stream);

int viorb_synth_plane_points_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count,
                                    const double* pose12, double z0, int batch, float* Pw, uint8_t* flags,
                                    int32_t* self_index /* [b][cap]: i for i < count[b], else -1; may be NULL */, void* stream);

/* The per-stream variant of viorb_frontend_pose_opt_device: variant[b] = 0 selects the (Frame, KeyFrame) overload for stream b
 * (Tracking's "if(mpLocalMapper->GetFirstVINSInited() || bMapUpdated)", reference src/Tracking.cc:454, :243), 1 the (Frame, Frame)
 * overload (NULL: all 1); streams with skip[b] != 0 (may be NULL) return at once with out_ns = cur_ns and info = 0, as
 * "if(nInitialCorrespondences<3) return 0" does. All the Frame-overload arrays must be given. obs arrays must be 16-byte aligned. */
int viorb_frontend_pose_opt_select_device(viorb_frontend* h, const uint8_t* variant, const uint8_t* skip, int compute_marg,
                                          const double* cur_ns, const double* last_ns, const double* prior_ns, const double* marg_cov_inv,
                                          const double* preint, const double* obs_cur, const int32_t* n_cur, const double* obs_last,
                                          const int32_t* n_last, int batch, double* out_ns, double* out_last_ns, uint8_t* outlier_cur,
                                          uint8_t* outlier_last, double* marg_out, double* info, void* stream);
/* self_index[b][i] = i where keypoint i of the last frame holds a map point (flags bit 0) and i < count[b], else -1: the `match`
 * input that makes viorb_frontend_build_observations_device build the LAST frame's own edges (reference src/Optimizer.cc:549-589). */
int viorb_frontend_self_index_device(viorb_frontend* h, const uint8_t* flags, const int32_t* count, int batch, int32_t* self_index,
                                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batched tracking sequence: what Tracking::TrackWithIMU followed by Tracking::TrackLocalMapWithIMU do for one
 * frame (reference src/Tracking.cc:412-534, :229-346) for `batch` independent mono-inertial streams, thresholds 20 / 10 / 15 / 30 and
 * the backup / revert decisions included, enqueued by ONE host call on two HIP streams of the tracker (extraction of frame k+1
 * overlaps the matching and pose solves of frame k). No host synchronisation inside a step. Per-stream outcome in state[b]:
 * ---------------------------------------------------------------------------------------------- */
#define VIORB_TRACK_OK           0   /* both stages returned true */
#define VIORB_TRACK_FEW_MATCHES  1   /* TrackWithIMU: nmatches < 20 after the 2*th retry: "return false" before any optimisation (:446-447) */
#define VIORB_TRACK_REVERT_1     2   /* TrackWithIMU: nmatchesMap < 10: frames reverted to the backup (IMU prediction), false (:518-533) */
#define VIORB_TRACK_REVERT_2     3   /* TrackLocalMapWithIMU: mnMatchesInliers < 15: reverted to the state it started from, false (:333-342) */
#define VIORB_TRACK_RELOC_FEW    4   /* TrackLocalMapWithIMU: relocalised recently and mnMatchesInliers < 30: false, no revert (:330-331) */

typedef struct viorb_tracker viorb_tracker;   /* opaque */
typedef struct viorb_tracker_config {
    viorb_extractor_params extractor;
    viorb_frontend_config  frontend;      /* bounds (Frame::ComputeImageBounds from width / height / dist_coef), scale tables and nlevels are filled in */
    int32_t width, height, batch, device;
    float   th_projection;                /* SearchByProjection window: 15 mono, 7 stereo (src/Tracking.cc:427-431) */
    int32_t track_local_map;              /* > 0: TrackWithIMU + TrackLocalMapWithIMU; 0: TrackWithIMU only; < 0: extract + grid + IMU prediction + SearchByProjection
                                             only ("ORB extract + match": no pose solve, the frame hands on its IMU prediction) */
    int32_t local_frames;                 /* local map = the points of this many frames before the last one (1..8) */
    int32_t compute_marg;                 /* bComputeMarg of the last solve of a frame */
    int32_t max_steps_ahead;              /* host runs at most this many steps in front of the device (<= 0: 8) */
    double  synth_plane_z0;               /* plane of the synthetic world (workload support; used with d_synth_pose12) */
} viorb_tracker_config;
typedef struct viorb_tracker_inputs {     /* device pointers; row b belongs to stream b */
    const uint8_t* d_images; int32_t image_stride; size_t image_pitch_bytes;   /* batch images, as viorb_extract_batch_device */
    const double*  d_imu; int32_t n_imu;  /* [b][n_imu][7] gyro3 acc3 t since the last frame */
    const double*  d_t_cur;               /* [b] frame stamps */
    const uint8_t* d_map_updated;         /* [b] != 0: mbMapUpdated -> PoseOptimization(Frame, KeyFrame) with the last frame as the
                                             key frame it was just promoted to; NULL: never */
    const uint8_t* d_recent_reloc;        /* [b] != 0: mCurrentFrame.mnId < mnLastRelocFrameId + mMaxFrames; NULL: never */
    const double*  d_t_next_last;         /* [b] stamp the frame carries as "last frame" (NULL: d_t_cur; periodic synthetic streams) */
    const double*  d_reset_ns;            /* [b][22] harness key-frame boundary: the next frame starts from this state (NULL: chained) */
    const double*  d_reset_marg;          /* [b][144] ... and this prior information (NULL: chained) */
    const double*  d_synth_pose12;        /* [b][12] double Rcw tcw: map points of the new last frame from the synthetic plane world
                                             (NULL: the caller supplies them with viorb_tracker_set_last_points_device before the next step) */
    const uint8_t* h_images;              /* live feed: the batch's images in HOST memory (same stride / pitch; page-locked for the copy to be
                                             asynchronous). When set, d_images is ignored: the step uploads them on the tracker's copy stream
                                             into a ring of max_steps_ahead + 2 device buffers and the extraction waits for that upload only —
                                             the upload of frame k + 1 overlaps the extraction and tracking of frame k. The host buffer may be
                                             reused once viorb_tracker_step has been called max_steps_ahead + 1 more times (or after _sync). */
} viorb_tracker_inputs;
typedef struct viorb_tracker_results {    /* device pointers into the tracker, valid after viorb_tracker_sync until the next step */
    int32_t cap;
    const int32_t *state, *status, *nmatches, *n_map, *n_loc, *inliers, *n_obs, *n_obs2, *cur_match, *loc_match, *last_count;
    const double *info, *info2, *pred_ns, *ns_stage1, *ns_stage2, *final_ns, *final_marg, *last_ns;
    const uint8_t *outlier_cur, *outlier_cur2, *last_flags;
    const float *last_Pw, *last_pts_f;
    const viorb_extractor* extractor;     /* the handle that holds the frame's keypoints / descriptors / pyramid */
} viorb_tracker_results;
int viorb_tracker_create(const viorb_tracker_config* cfg, viorb_tracker** out);
int viorb_tracker_destroy(viorb_tracker* h);
int viorb_tracker_capacity(const viorb_tracker* h, int* cap);
/* First frame of every stream: extract and adopt as last frame with the given state / prior information. Synchronises. */
int viorb_tracker_bootstrap(viorb_tracker* h, const uint8_t* d_images, int stride, size_t image_pitch_bytes, const double* d_ns0,
                            const double* d_t0, const double* d_marg_cov_inv, const double* d_synth_pose12, void* caller_stream);
/* Map points of the current last frame from the caller's map: Pw[b][cap][3], flags[b][cap] (bit0 point, bit1 outlier, bit2 has
 * observations), pts_f[b][cap][8] (isInFrustum fields; NULL without the local-map stage). */
int viorb_tracker_set_last_points_device(viorb_tracker* h, const float* d_Pw, const uint8_t* d_flags, const float* d_pts_f, void* caller_stream);
/* One frame for every stream. The inputs must stay valid until the step has completed (viorb_tracker_sync, or max_steps_ahead
 * later calls): they are read on the tracker's own streams. */
int viorb_tracker_step(viorb_tracker* h, const viorb_tracker_inputs* in, void* caller_stream);
int viorb_tracker_sync(viorb_tracker* h);
int viorb_tracker_results_device(const viorb_tracker* h, viorb_tracker_results* out);
/* Host cost of the steps since the last reset: seconds spent enqueueing (launch calls only) and seconds blocked in the
 * max_steps_ahead throttle. */
int viorb_tracker_host_stats(viorb_tracker* h, double* enqueue_s, double* throttle_s, long long* steps, int reset);
/* Blocking host <-> device copies for callers that hold raw device addresses. */
int viorb_memcpy_dtoh(void* dst_host, const void* src_device, size_t bytes);
int viorb_memcpy_htod(void* dst_device, const void* src_host, size_t bytes);

/* hipMemcpyAsync(device -> device) on `stream`, for callers that hold raw device addresses. */
int viorb_memcpy_dtod_async(void* dst, const void* src, size_t bytes, void* stream);

/* Per-kernel timing with HIP events recorded on the stream each kernel is launched on. Off by default.
 * viorb_profile_read synchronises the device; names_buf gets the kernel names separated by newlines,
 * total_ms[i] / calls[i] the summed duration and launch count of name i since the last reset. */
int viorb_profile_enable(int on);
int viorb_profile_reset(void);
/* Restrict the timing to one kernel name or a comma-separated list of names (NULL or "" = every kernel). An event pair around a kernel costs about 8 us of stream
 * time, so timing all ~30 launches of a tracking step slows the step by ~5 %; bench.py times only its roofline kernel. */
int viorb_profile_select(const char* kernel_name);
int viorb_profile_read(char* names_buf, int names_cap, double* total_ms, int* calls, int cap, int* n);
/* Start / end of every recorded interval (ms since the first record; both streams on one clock), in recording order; slot[i]
 * indexes the names viorb_profile_read returns. */
int viorb_profile_timeline(double* start_ms, double* end_ms, int* slot, int cap, int* n);

/* Host-buffer drop-ins for single calls (they stage through the device and include the PCIe copies). */
int viorb_descriptor_distance(const uint8_t* a, const uint8_t* b);       /* ORBmatcher::DescriptorDistance */
/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono = true) with host arrays: builds the
 * current frame's grid and runs the search on the device. bounds4 = mnMinX mnMaxX mnMinY mnMaxY, pose12 =
 * Rcw(9) tcw(3) of CurrentFrame.mTcw, intr4 = fx fy cx cy, scale_factors[nlevels]; last_flags as above.
 * cur_match[ncur] and *nmatches receive the assignment and the function's return value. */
int viorb_search_by_projection_frame(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, int ncur, const float bounds4[4],
                                     const float pose12[12], const float intr4[4], const float* scale_factors, int nlevels,
                                     const viorb_keypoint* last_kps, int nlast, const uint8_t* last_flags, const float* last_Pw,
                                     const uint8_t* last_desc, float th, int check_orientation, int32_t* cur_match, int* nmatches);
/* Tracking::SearchLocalPoints' matcher call (reference src/Tracking.cc:1904-1958): Frame::isInFrustum(pMP, 0.5) for every local map point
 * (src/Frame.cc:449-505) followed by ORBmatcher(nnratio).SearchByProjection(F, vpMapPoints, th) (src/ORBmatcher.cc:45-129), host arrays.
 * Layouts as viorb_frontend_search_local_points_device: pts_f[p][8] = Pw3 normal3 mfMinDistance mfMaxDistance, pts_flags bit0 !isBad,
 * bit1 mnLastFrameSeen == frame id, bit2 Observations() > 0, cur_owner_obs[i] != 0 where keypoint i already holds a map point with
 * observations. match[i] = local point given to keypoint i or -1; frustum5 (may be NULL) [p][5] = mbTrackInView mTrackProjX mTrackProjY
 * mTrackViewCos mnTrackScaleLevel. */
int viorb_search_by_projection_points(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, int ncur, const float bounds4[4],
                                      const float pose12[12], const float intr4[4], const float* scale_factors, int nlevels,
                                      const float* pts_f, const uint8_t* pts_flags, const uint8_t* pts_desc, int npts, float th,
                                      float nnratio, const uint8_t* cur_owner_obs, int32_t* match, int* nmatches, float* frustum5);
/* ... for a frame with right coordinates (stereo / RGB-D): cur_uright = F.mvuRight, bf = F.mbf, proj_xr (may be NULL) [npts] = mTrackProjXR; see
 * viorb_frontend_search_local_points_stereo_device. With cur_uright all <= 0 the matches equal viorb_search_by_projection_points'. */
int viorb_search_by_projection_points_stereo(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, float bf, int ncur,
                                             const float bounds4[4], const float pose12[12], const float intr4[4], const float* scale_factors, int nlevels,
                                             const float* pts_f, const uint8_t* pts_flags, const uint8_t* pts_desc, int npts, float th, float nnratio,
                                             const uint8_t* cur_owner_obs, int32_t* match, int* nmatches, float* frustum5, float* proj_xr);
/* ... and with bMono = false (stereo / RGB-D, Tracking.cc:432 with mSensor != MONOCULAR): see viorb_frontend_search_projection_stereo_device. */
int viorb_search_by_projection_frame_stereo(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int ncur,
                                            const float bounds4[4], const float pose12[12], const float last_pose12[12], const float intr4[4],
                                            float bf, float mb, const float* scale_factors, int nlevels, const viorb_keypoint* last_kps, int nlast,
                                            const uint8_t* last_flags, const float* last_Pw, const uint8_t* last_desc, float th,
                                            int check_orientation, int32_t* cur_match, int* nmatches);
int viorb_preintegrate(const double* imu, int n_imu, const double bg[3], const double ba[3], double t_last,
                       double t_cur, double* preint142);
int viorb_pose_opt_vi(int variant, int compute_marg, const double cur_ns[22], const double last_ns[22],
                      const double prior_ns[22], const double* marg_cov_inv144, const double preint[142],
                      const double gw[3], const double cam[16], const double* obs_cur, int n_cur, const double* obs_last,
                      int n_last, double out_ns[22], double out_last_ns[22], uint8_t* outlier_cur, uint8_t* outlier_last,
                      double* marg_out144, double info[4]);

/* Host-buffer form of the vision-only solve; intr5 = fx fy cx cy bf (float, as Frame stores them). */
int viorb_pose_opt_se3(const float pose12[12], const float intr5[5], const double* obs7, int n, float out_pose12[12],
                       uint8_t* outlier, double info[4]);

/* Optimizer::LocalBundleAdjustmentNavState (reference src/Optimizer.cc:1690-2241): the local-mapping window solve.
 * Vertices: n_local free key frames (PVR 9 + accelerometer-bias 3, src/IMU/g2otypes.h:16-75), fixed key frames
 * [n_local, nk), np marginalised points (Thirdparty/g2o/g2o/types/types_sba.h VertexSBAPointXYZ). Factors: one
 * EdgeNavStatePVR + EdgeNavStateBias per local key frame i between its predecessor (i-1, or prev_kf for i == 0;
 * prev_kf == -1: none) and i (src/Optimizer.cc:1872-1930), one EdgeNavStatePVRPointXYZ per observation
 * (src/IMU/g2otypes.h:129-203). Solver: g2o Levenberg with the Schur complement of the point block
 * (Thirdparty/g2o/g2o/core/block_solver.hpp:367-486), optimize(5), chi2 > 5.991 / depth <= 0 edges to level 1 and the
 * mono kernel dropped, optimize(10) (src/Optimizer.cc:2027-2063); erase[k] = 1 for observations the caller must remove
 * (:2105-2118). kfs [nk][22] (local window first, chronological), preint [n_local][142] of the interval ending at local
 * key frame i, points [np][3], edge_idx [ne][2] = (point, key frame) sorted by point, edge_obs [ne][3] = u v invSigma2.
 * stop: the reference's pbStopFlag (may be NULL). g2o polls it once per iteration and once per LM trial; here the waiting host thread
 * mirrors it into a page-locked word that the device-side Levenberg control reads after every trial, so a raised flag ends the solve
 * after the trial in flight (a rejected one is restored first), as "&& !terminate()" does in the reference. Outputs: kfs_out [n_local][22],
 * points_out [np][3], erase [ne], info = chi2 after optimize(5), final chi2, iterations of both runs, 0, 0.
 * Host buffers in and out (the caller is the LocalMapping thread); all arithmetic runs on the GPU in FP64, on the device
 * chosen with viorb_local_ba_set_device (default: the calling thread's current HIP device). Re-entrant: concurrent callers get their own stream and device arena. */
/* Device of all window solves of this process: >= 0 fixes it, < 0 (default) = the calling thread's current HIP device. */
int viorb_local_ba_set_device(int device);
int viorb_local_ba_navstate(const double* kfs, int nk, int n_local, int prev_kf, const double* preint,
                            const double* points, int np, const int32_t* edge_idx, const double* edge_obs, int ne,
                            const double gw[3], const double cam[16], const volatile int* stop, double* kfs_out,
                            double* points_out, uint8_t* erase, double info[6]);

/* Several windows at once (one per camera stream in a multi-stream deployment; the reference runs one LocalMapping thread per system):
 * each entry carries the arguments of one viorb_local_ba_navstate call and receives its return code in `status`. All windows advance in
 * LOCK STEP, max_in_flight at a time (<= 0: 128; at most 512): one launch per solver step covers every window of the group and each
 * window's Levenberg / two-phase state machine lives in its control block on the device — a single window is a chain of small
 * latency-bound launches that leaves most of the GPU idle. The windows of a group are prepared (checks, graph bookkeeping, upload) by a
 * few host threads. With VIORB_LBA_STREAMS=1 the round-1 driver runs instead (one HIP stream and host LM loop per window, max_in_flight
 * of them, <= 0: 16). When a group fails (a HIP error), every window of it that has no result yet and every later window gets that
 * error in `status`. Results are those of the individual calls (to rounding: the order of the terms of a Schur block's sum is not
 * fixed from run to run). */
typedef struct viorb_lba_window {
    const double* kfs; int32_t nk, n_local, prev_kf; const double* preint; const double* points; int32_t np;
    const int32_t* edge_idx; const double* edge_obs; int32_t ne; const double* gw; const double* cam; const volatile int* stop;
    double* kfs_out; double* points_out; uint8_t* erase; double* info /* [6] */; int32_t status;
} viorb_lba_window;
int viorb_local_ba_navstate_batch(viorb_lba_window* windows, int n_windows, int max_in_flight);

/* Vision-only Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:3980-4311; BlockSolver_6_3): kfs [nk][7] = g2o::SE3Quat of
 * each key frame's Tcw as qx qy qz qw tx ty tz (Converter::toSE3Quat), the n_local free ones first, then the fixed ones (lFixedCameras,
 * and key frame 0 when it is local: :4055); points [np][3]; edge_idx [ne][2] = (point, key frame) sorted by point; edge_obs [ne][4] =
 * u v uRight invSigma2 with uRight < 0 for a monocular observation (EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ,
 * Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:66-250); intr5 = fx fy cx cy bf. Same optimize(5) / gate / optimize(10) scheme
 * and outputs as viorb_local_ba_navstate (chi-square gates 5.991 mono, 7.815 stereo). n_local <= 40 (a 240 x 240 reduced system). */
int viorb_local_ba_se3(const double* kfs, int nk, int n_local, const double* points, int np, const int32_t* edge_idx,
                       const double* edge_obs, int ne, const double intr5[5], const volatile int* stop, double* kfs_out,
                       double* points_out, uint8_t* erase, double info[6]);

/* The batch form of viorb_local_ba_se3, as viorb_local_ba_navstate_batch is for the NavState window. */
typedef struct viorb_lba_se3_window {
    const double* kfs; int32_t nk, n_local; const double* points; int32_t np; const int32_t* edge_idx; const double* edge_obs; int32_t ne;
    const double* intr5; const volatile int* stop; double* kfs_out; double* points_out; uint8_t* erase; double* info /* [6] */; int32_t status;
} viorb_lba_se3_window;
int viorb_local_ba_se3_batch(viorb_lba_se3_window* windows, int n_windows, int max_in_flight);

/* ---- Bag of words: DBoW2 vocabulary-tree descent and ORBmatcher::SearchByBoW -------------------------------------
 * viorb_vocabulary replaces ORBVocabulary (= DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, reference
 * include/ORBVocabulary.h:30-31) for the one call the trackers make, transform(features, BowVector, FeatureVector, 4)
 * (src/Frame.cc:575-582, src/KeyFrame.cc ComputeBoW). Flat tree: node 0 is the root; the children of node n are
 * child_ids[child_start[n] .. child_start[n+1]) in file order (the order TemplatedVocabulary::m_nodes[n].children holds,
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1351-1508 loadFromTextFile); a node without children is a word with
 * word_id >= 0 and its TF-IDF weight; desc[n][32] is the node descriptor. The arrays are copied to the device. */
typedef struct viorb_vocabulary viorb_vocabulary;   /* opaque */
int viorb_vocabulary_create(int n_nodes, int L, const int32_t* child_start, const int32_t* child_ids, const uint8_t* desc,
                            const int32_t* word_id, const double* weight, viorb_vocabulary** out);
int viorb_vocabulary_destroy(viorb_vocabulary* v);
/* The two file formats of the reference's vocabulary: text = TemplatedVocabulary::loadFromTextFile /
 * saveToTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1351-1460: "k L scoring weighting", then "parent isLeaf d0..d31 weight"
 * per node), binary = loadFromBinaryFile / saveToBinaryFile (:1462-1533, the file tools/bin_vocabulary.cc writes: u32 nb_nodes,
 * u32 size_node, i32 k, L, scoring, weighting, then { i32 parent; u8 desc[32]; f32 weight; u8 is_leaf } per node). Node ids are file
 * order, children keep file order, word ids count the leaves in file order. viorb_vocabulary_read_file parses into malloc'ed flat
 * arrays (the layout of viorb_vocabulary_create; no GPU needed; release with viorb_vocabulary_flat_free); the load functions parse and
 * create the device vocabulary; the save functions write a flat tree whose ids are in file order. */
typedef struct viorb_vocabulary_flat {
    int32_t n_nodes, k, L, n_words;
    int32_t *child_start, *child_ids, *word_id; uint8_t* desc; double* weight;
} viorb_vocabulary_flat;
int viorb_vocabulary_read_file(const char* path, int binary, viorb_vocabulary_flat* out);
void viorb_vocabulary_flat_free(viorb_vocabulary_flat* f);
int viorb_vocabulary_load_text(const char* path, viorb_vocabulary** out);
int viorb_vocabulary_load_binary(const char* path, viorb_vocabulary** out);
int viorb_vocabulary_save_text(const char* path, int n_nodes, int k, int L, const int32_t* child_start, const int32_t* child_ids,
                               const uint8_t* desc, const double* weight);
int viorb_vocabulary_save_binary(const char* path, int n_nodes, int k, int L, const int32_t* child_start, const int32_t* child_ids,
                                 const uint8_t* desc, const double* weight);

/* TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (TemplatedVocabulary.h:1231-1272) for every
 * descriptor: word[i], weight[i] of the leaf reached and node[i] = the ancestor at level L - levelsup (0 = root when
 * L - levelsup <= 0). The caller builds BowVector (sum of weights per word over features with weight > 0, L1-normalised,
 * BowVector.cpp:36-85) and FeatureVector (features with weight > 0 grouped by node, FeatureVector.cpp:31-45) from them;
 * viorb_amd/shim and viorb_amd/frontend.py show how. Device form: desc[(b*cap+i)*32], count[b], outputs [b*cap+i]. */
int viorb_bow_transform_device(const viorb_vocabulary* v, const uint8_t* desc, const int32_t* count, int cap, int batch,
                               int levelsup, int32_t* word, double* weight, int32_t* node, void* stream);
int viorb_bow_transform(const viorb_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word,
                        double* weight, int32_t* node);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:159-288).
 * kf_node / f_node: the FeatureVector node of each feature (-1: feature absent from the FeatureVector, i.e. a stopped
 * word); kf_has_point[i] = pKF's map point i exists and !isBad(); angles are read from the keypoints (mvKeysUn / mvKeys
 * carry the same angle). match[iF] = key-frame feature index whose map point the frame feature receives, or -1;
 * nnratio = mfNNratio (0.7 in Tracking.cc:1500). Device form: all arrays [b*cap + i], counts per pair. */
int viorb_search_by_bow_device(const viorb_keypoint* kf_kps, const uint8_t* kf_desc, const int32_t* kf_node,
                               const uint8_t* kf_has_point, const int32_t* kf_count, const viorb_keypoint* f_kps,
                               const uint8_t* f_desc, const int32_t* f_node, const int32_t* f_count, int cap, int batch,
                               float nnratio, int check_orientation, int32_t* match, int32_t* nmatches, void* stream);
int viorb_search_by_bow(const viorb_keypoint* kf_kps, const uint8_t* kf_desc, const int32_t* kf_node,
                        const uint8_t* kf_has_point, int nkf, const viorb_keypoint* f_kps, const uint8_t* f_desc,
                        const int32_t* f_node, int nf, float nnratio, int check_orientation, int32_t* match,
                        int* nmatches);

/* ---- Brute-force Hamming matcher (north_star "Hamming brute-force"; SURVEY.md §8b viorb_match_bruteforce) ---------------------
 * For every query descriptor q[i] (32 bytes): best[i] / second[i] = the smallest and second-smallest ORBmatcher::DescriptorDistance
 * (reference src/ORBmatcher.cc:1648-1664) over ALL candidates c[0..nc) and idx[i] = the FIRST candidate at the smallest distance —
 * the bestDist1 / bestDist2 / bestIdx scan of the reference's searches (e.g. src/ORBmatcher.cc:204-222: "if(dist<bestDist1){
 * bestDist2=bestDist1; bestDist1=dist; bestIdx=i;} else if(dist<bestDist2) bestDist2=dist;", both initialised to 256) without a
 * window or a vocabulary node restricting the candidates. nc == 0: best = second = 256, idx = -1. The caller applies its own
 * TH_LOW / TH_HIGH / nnratio gates. Device form: q_desc[(b*qcap + i)*32], nq[b], c_desc[(b*ccap + j)*32], nc[b]; outputs [b*qcap + i]. */
int viorb_match_bruteforce_device(const uint8_t* q_desc, const int32_t* nq, int qcap, const uint8_t* c_desc, const int32_t* nc, int ccap,
                                  int batch, int32_t* best, int32_t* second, int32_t* idx, void* stream);
int viorb_match_bruteforce(const uint8_t* q, int nq, const uint8_t* c, int nc, int32_t* best, int32_t* second, int32_t* idx);

/* ---- Key-frame side matchers LocalMapping runs around the local BA (reference src/LocalMapping.cc:1296, 1522, 1547) ------------
 * ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (src/ORBmatcher.cc:657-823): node1 / node2 are
 * the FeatureVector nodes per feature (-1 absent), has_point = GetMapPoint(i) != NULL, uright = mvuRight (< 0 mono), F12 row-major
 * (LocalMapping::ComputeF12), Cw1 = pKF1->GetCameraCenter(), pose12_2 = pKF2's Tcw, intr4 / scale_factors2 / level_sigma2_2 =
 * pKF2's fx fy cx cy, mvScaleFactors, mvLevelSigma2. match12[i1] = i2 or -1 (the vMatchedPairs list is its non-negative entries
 * in i1 order). Device form: arrays [b*cap + i], n1[b] / n2[b] counts, F12 [b][9], Cw1 [b][3], pose12_2 [b][12]. */
int viorb_search_for_triangulation_device(const viorb_keypoint* k1, const uint8_t* d1, const uint8_t* has_point1,
                                          const float* uright1, const int32_t* node1, const int32_t* n1,
                                          const viorb_keypoint* k2, const uint8_t* d2, const uint8_t* has_point2,
                                          const float* uright2, const int32_t* node2, const int32_t* n2, const float* F12,
                                          const float* Cw1, const float* pose12_2, const float intr4[4],
                                          const float* scale_factors2, const float* level_sigma2_2, int nlevels, int only_stereo,
                                          int check_orientation, int cap, int batch, int32_t* match12, int32_t* nmatches,
                                          void* stream);
int viorb_search_for_triangulation(const viorb_keypoint* k1, const uint8_t* d1, const uint8_t* has_point1, const float* uright1,
                                   const int32_t* node1, int n1, const viorb_keypoint* k2, const uint8_t* d2,
                                   const uint8_t* has_point2, const float* uright2, const int32_t* node2, int n2,
                                   const float F12[9], const float Cw1[3], const float pose12_2[12], const float intr4[4],
                                   const float* scale_factors2, const float* level_sigma2_2, int nlevels, int only_stereo,
                                   int check_orientation, int32_t* match12, int* nmatches);

/* ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th) (src/ORBmatcher.cc:825-975): pts_f[p][8] = Pw3 normal3
 * minDist maxDist (GetWorldPos, GetNormal, mfMinDistance, mfMaxDistance), pts_valid[p] = pMP && !isBad() && !IsInKeyFrame(pKF),
 * pts_desc = GetDescriptor(); uright = pKF->mvuRight, bf = pKF->mbf, cell_start / cell_idx = the key frame's 64x48 grid
 * (viorb_frontend_grid_device). best_idx[p] = the feature the point fuses with (descriptor distance <= TH_LOW) or -1; the caller
 * then replaces / adds observations in point order as :956-972 do. nfused = number of non-negative entries. */
int viorb_frontend_fuse_device(viorb_frontend* h, const viorb_keypoint* kps, const uint8_t* desc, const float* uright,
                               const int32_t* count, const int32_t* cell_start, const int32_t* cell_idx, const float* pose12,
                               const float* pts_f, const uint8_t* pts_valid, const uint8_t* pts_desc, const int32_t* pts_count,
                               int pcap, float th, float bf, int batch, int32_t* best_idx, int32_t* nfused, void* stream);
int viorb_fuse(const viorb_keypoint* kps, const uint8_t* desc, const float* uright, int n, const float bounds4[4],
               const float pose12[12], const float intr5[5], const float* scale_factors, const float* inv_level_sigma2, int nlevels,
               const float* pts_f, const uint8_t* pts_valid, const uint8_t* pts_desc, int npts, float th, int32_t* best_idx,
               int* nfused);

/* Host-only test hooks (no GPU needed; used by the CPU test-suite to compare product host code with
 * the oracle): the flat-array formulation of DistributeOctTree that the device kernel mirrors
 * (keys packed x | y<<12 | score<<24, border-relative), and the scalar math shared with the kernels. */
int viorb_debug_octree_host(const uint32_t* keys, int n, int width, int height, int N, uint32_t* out,
                            int cap, int* nout);
float viorb_debug_fast_atan2(float y, float x);
void viorb_debug_sincos(float radians, float* s, float* c);
/* vio_core.h (the FP64 edge/state functions the solver kernel uses) compiled for the host. J layouts:
 * pvr edge 9x21 = [d/d i(P V Phi) | d/d j(P V Phi) | d/d bias_i acc]; proj edge 2x3 (dP) then 2x3 (dPhi);
 * prior edge 12x12 = [d/d(P V Phi) | d/d bias acc]; small60 = dP3 dV3 dR9 JPg9 JPa9 JVg9 JVa9 JRg9. */
void viorb_debug_pvr_edge(const double* i22, const double* j22, const double* b22, const double* preint142,
                          const double* gw, double* e9, double* J189);
void viorb_debug_proj_edge(const double* ns22, const double* cam16, const double* obs6, double* e2, double* J12);
void viorb_debug_prior_edge(const double* pvr22, const double* bias22, const double* prior22, double* e12, double* J144);
void viorb_debug_update_ns(const double* ns22, const double* preint142, const double* gw, const double* cam16,
                           double* out22, float* pose12);
void viorb_debug_preint_step(double* small60, const double* omega, const double* acc, double dt);

#ifdef __cplusplus
}
#endif
#endif /* VIORB_H */
