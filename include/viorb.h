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
 * changes. max_batch = 1 gives the literal per-camera object of the reference. */
int viorb_extractor_create(const viorb_extractor_params* params, int max_batch, int device,
                           viorb_extractor** out);
int viorb_extractor_destroy(viorb_extractor* h);

/* Scale tables (GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares, reference include/ORBextractor.h:66-84) and per-level feature quotas
 * (mnFeaturesPerLevel). Any pointer may be NULL. Arrays hold nlevels entries. */
int viorb_extractor_tables(const viorb_extractor* h, float* scale, float* inv_scale, float* sigma2,
                           float* inv_sigma2, int32_t* features_per_level);

/* Upper bound on keypoints per image (sum of per-level quota + 2): size output buffers with it. */
int viorb_extractor_max_keypoints(const viorb_extractor* h, int* cap);

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

/* Stage introspection for parity tests: FAST candidates of one level before the quadtree
 * (ComputeKeyPointsOctTree's vToDistributeKeys, src/ORBextractor.cc:765-830), in the reference's
 * push order, as (x, y, response) int32 triples relative to the (16,16) border origin.
 * which = 0: candidates; which = 1: keypoints kept by DistributeOctTree, level coordinates. */
int viorb_extractor_debug_level_points(viorb_extractor* h, int b, int level, int which, int32_t* xyr,
                                       int cap, int* n);

/* Host-only test hooks (no GPU needed; used by the CPU test-suite to compare product host code with
 * the oracle): the flat-array formulation of DistributeOctTree that the device kernel mirrors
 * (keys packed x | y<<12 | score<<24, border-relative), and the scalar math shared with the kernels. */
int viorb_debug_octree_host(const uint32_t* keys, int n, int width, int height, int N, uint32_t* out,
                            int cap, int* nout);
float viorb_debug_fast_atan2(float y, float x);
void viorb_debug_sincos(float radians, float* s, float* c);

#ifdef __cplusplus
}
#endif
#endif /* VIORB_H */
