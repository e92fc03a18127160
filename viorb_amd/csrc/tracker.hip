// viorb_amd/csrc/tracker.hip — the batched tracking-sequence harness in C++: for B independent mono-inertial
// streams per GPU, one call enqueues what Tracking::TrackWithIMU followed by Tracking::TrackLocalMapWithIMU do for a frame
// (reference src/Tracking.cc:412-534, :229-346), including their thresholds and backup / revert decisions, as per-stream selects on
// the device — no host synchronisation inside a step:
//
//   extract (stream s_ex)                      ORBextractor::operator()                        src/Frame.cc:427-433
//   IMU pre-integration + prediction           Tracking::PredictNavStateByIMU                  src/Tracking.cc:348-410
//   SearchByProjection(th), again with 2*th    "if(nmatches<20)"                               :432-444
//   nmatches < 20 -> state FEW_MATCHES         "if(nmatches<20) return false;"                 :446-447  (no optimisation)
//   PoseOptimization(Frame, KeyFrame | Frame)  mbMapUpdated selects the KeyFrame overload      :454-469  (per stream: map_updated[b])
//   discard outliers, nmatchesMap              :489-507
//   nmatchesMap < 10 -> revert, state REVERT_1 "mCurrentFrame = backupCurrentFrame"            :518-533
//   SearchLocalPoints                          Tracking::SearchLocalPoints                     :1904-1958
//   PoseOptimization(..., bComputeMarg = true) :275 / :296
//   mnMatchesInliers                           map points that are inliers and have observations   :307-325
//   recent relocalisation && inliers < 30 -> state RELOC_FEW (no revert)                       :330-331
//   inliers < 15 -> revert to the state TrackLocalMapWithIMU started from, state REVERT_2      :333-342
//
// A revert on the device is a select: the NavState (and marginal) the frame hands to the next one is the backup's — the IMU prediction
// after a stage-1 failure, the stage-1 result after a stage-2 revert; later kernels of a failed stream are skipped through the
// solver's skip flags. What Tracking::Track does with a `false` (IMU-only tracking, relocalisation, reset: :1036-1114) is the
// caller's state machine and stays outside: the per-stream state code is an output.
//
// Two HIP streams per tracker: the extraction of frame k+1 (chip-filling) overlaps the matching + pose solves of frame k
// (latency-bound, one workgroup per stream); two extractor handles alternate. Map management is not on this path (LocalMapping does
// it in the reference): the points of the new last frame come either from the caller (viorb_tracker_set_last_points_device) or from
// the synthetic plane world of viorb_amd/synth.py (workload support for bench.py / tests).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdlib>
#include <deque>
#include <vector>
#include "viorb_common.h"

namespace viorb {

// per-stream decisions --------------------------------------------------------------------------------------------------------
// stage 0: skip1 = nmatches < 20
__global__ void k_track_gate0(const int* __restrict__ nmatches, int batch, uint8_t* __restrict__ skip1, const uint8_t* __restrict__ map_updated,
                              uint8_t* __restrict__ variant) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    skip1[b] = nmatches[b] < 20;
    variant[b] = (map_updated && map_updated[b]) ? 0 : 1;              // KeyFrame overload when the map was updated
}
// stage 1: state after TrackWithIMU, the NavState TrackLocalMapWithIMU starts from, skip flag of stage 2
__global__ void k_track_gate1(const int* __restrict__ nmatches, const int* __restrict__ n_map, const double* __restrict__ pred_ns,
                              const double* __restrict__ opt_ns, int batch, int* __restrict__ state, double* __restrict__ ns1,
                              uint8_t* __restrict__ skip2) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= batch) return;
    const int st = nmatches[b] < 20 ? VIORB_TRACK_FEW_MATCHES : (n_map[b] < 10 ? VIORB_TRACK_REVERT_1 : VIORB_TRACK_OK);
    if (t < 22) ns1[(size_t)b * 22 + t] = (st ? pred_ns : opt_ns)[(size_t)b * 22 + t];
    if (t == 0) { state[b] = st; skip2[b] = st != 0; }
}
// mnMatchesInliers of TrackLocalMapWithIMU: inlier edges of the second solve whose map point has observations (:307-325)
__global__ __launch_bounds__(256) void k_track_count_inliers(const uint8_t* __restrict__ outlier2, const int* __restrict__ idx2, const int* __restrict__ n2,
                                                             const int* __restrict__ match_a, const uint8_t* __restrict__ flags_a,
                                                             const int* __restrict__ match_b, const uint8_t* __restrict__ flags_b, int stride_b,
                                                             int cap, int* __restrict__ inliers) {
    __shared__ int s_cnt;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int c = 0;
    const int n = min(n2[b], cap);
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        if (outlier2[(size_t)b * cap + k]) continue;
        const int kp = idx2[(size_t)b * cap + k];
        const int ma = match_a[(size_t)b * cap + kp];
        uint8_t f;
        if (ma >= 0) f = flags_a[(size_t)b * cap + ma];
        else { const int mb = match_b ? match_b[(size_t)b * cap + kp] : -1; f = (mb >= 0 && flags_b) ? flags_b[(size_t)b * stride_b + mb] : 0; }
        c += (f & 4) != 0;
    }
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) inliers[b] = s_cnt;
}
// stage 2 + what the frame hands to the next one: NavState and prior information
__global__ void k_track_final(int* __restrict__ state, const int* __restrict__ inliers, const uint8_t* __restrict__ recent_reloc,
                              const double* __restrict__ ns1, const double* __restrict__ ns2, const double* __restrict__ marg_new,
                              const double* __restrict__ marg_old, const double* __restrict__ reset_ns, const double* __restrict__ reset_marg,
                              int two_stage, int have_marg, int batch, double* __restrict__ final_ns, double* __restrict__ final_marg) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= batch) return;
    int st = state[b];
    if (two_stage && st == VIORB_TRACK_OK) {
        const int inl = inliers[b];
        if (recent_reloc && recent_reloc[b] && inl < 30) st = VIORB_TRACK_RELOC_FEW;        // "return false" without the revert
        else if (inl < 15) st = VIORB_TRACK_REVERT_2;
    }
    // the optimised second-stage state stands unless that stage was reverted or never ran
    const bool keep2 = two_stage && (st == VIORB_TRACK_OK || st == VIORB_TRACK_RELOC_FEW);
    const bool tracked = st == VIORB_TRACK_OK || st == VIORB_TRACK_RELOC_FEW;
    if (t < 22) final_ns[(size_t)b * 22 + t] = reset_ns ? reset_ns[(size_t)b * 22 + t] : (keep2 ? ns2 : ns1)[(size_t)b * 22 + t];
    if (t < 144) {
        const double* src = reset_marg ? reset_marg : ((have_marg && tracked) ? marg_new : marg_old);
        final_marg[(size_t)b * 144 + t] = src[(size_t)b * 144 + t];
    }
    __syncthreads();                                  // every wave has read the stage-1 state
    if (t == 0) state[b] = st;
}
__global__ void k_track_merge_status(const int* __restrict__ a, const int* __restrict__ b2, const int* __restrict__ c, int batch, int* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    int s = a ? a[b] : 0;
    if (s == 0 && b2) s = b2[b];
    if (s == 0 && c) s = c[b];
    out[b] = s;
}

struct DevArena {
    std::vector<void*> ptrs;
    hipError_t err = hipSuccess;
    template <class T> T* get(size_t n) {
        void* p = nullptr;
        if (err != hipSuccess) return nullptr;
        err = hipMalloc(&p, sizeof(T) * (n ? n : 1));
        if (err != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, sizeof(T) * (n ? n : 1));
        ptrs.push_back(p);
        return (T*)p;
    }
    void release() { for (void* p : ptrs) (void)hipFree(p); ptrs.clear(); }
};

} // namespace viorb

using namespace viorb;

struct viorb_tracker {
    viorb_tracker_config cfg;
    int B = 0, cap = 0, device = 0, nlevels = 0;
    viorb_extractor* ex[2] = {nullptr, nullptr};
    viorb_frontend* fe = nullptr;
    hipStream_t s_ex = nullptr, s_ex2 = nullptr, s_tr = nullptr;       // s_ex2: the second extractor handle's stream (VIORB_TRACKER_EX_STREAMS=2)
    hipEvent_t ev_in = nullptr, ev_ex[2] = {nullptr, nullptr}, ev_tr[2] = {nullptr, nullptr};
    bool ev_tr_valid[2] = {false, false};
    std::deque<hipEvent_t> in_flight; std::vector<hipEvent_t> ev_pool;
    long long k = 0, rolls = 0;
    int cur_slot = 0;
    DevArena mem;
    // last frame
    viorb_keypoint* last_kps; uint8_t* last_desc; int* last_count; uint8_t* last_flags; float* last_Pw; float* last_pts_f; int* last_self;
    double *last_ns, *prior_ns, *marg_cov_inv, *t_last;
    // local map: [b][local_frames][cap]
    float* loc_pts_f; uint8_t* loc_flags; uint8_t* loc_desc; int* loc_count;
    // per step
    int *cell_start, *cell_idx; double *preint, *cur_ns; float *pose12, *pose12_b;
    int *cur_match, *nmatches, *status_s1, *status_s2, *status; double *obs_cur, *obs_last, *obs_cur2; int *idx_cur, *idx_last, *idx_cur2;
    int *n_cur, *n_last, *n_cur2; double *out_ns, *out_last_ns, *out_ns2, *ns1, *final_ns, *final_marg, *marg_out, *info, *info2;
    uint8_t *outlier_cur, *outlier_last, *outlier_cur2, *owner_obs, *skip1, *skip2, *variant; int *n_map, *loc_match, *n_loc, *state, *inliers;
    bool undistort = false; viorb_keypoint* kps_un = nullptr;      // mvKeysUn of the frame in flight (Frame::UndistortKeyPoints)
    // live feed: host images -> ring of device buffers on a copy stream (the third stream of the tracker)
    hipStream_t s_cp = nullptr; std::vector<uint8_t*> stage; std::vector<hipEvent_t> ev_cp; size_t stage_bytes = 0; double upload_bytes = 0;
    // host statistics
    double enqueue_s = 0, throttle_s = 0; long long steps = 0;
};

#define TR_TRY(x) do { int _rc = (x); if (_rc != VIORB_OK) return _rc; } while (0)

static int tracker_roll(viorb_tracker* h, const viorb_keypoint* kps, const uint8_t* desc, const int* count, const double* ns_src, const double* t_src,
                        const double* marg_src, const double* synth_pose12, hipStream_t st) {
    const bool tlm = h->cfg.track_local_map > 0;
    TR_TRY(viorb_frontend_roll_device(h->fe, kps, desc, count, h->last_kps, h->last_desc, h->last_count, tlm ? h->last_pts_f : nullptr, h->last_flags,
                                      tlm ? h->loc_pts_f : nullptr, tlm ? h->loc_desc : nullptr, tlm ? h->loc_flags : nullptr, h->cfg.local_frames,
                                      (tlm && h->rolls > 0) ? 1 : 0, ns_src, h->last_ns, h->prior_ns, t_src, h->t_last, marg_src,
                                      marg_src ? h->marg_cov_inv : nullptr, h->B, st));
    if (synth_pose12) {
        TR_TRY(viorb_synth_plane_points_device(h->fe, h->last_kps, h->last_count, synth_pose12, h->cfg.synth_plane_z0, h->B, h->last_Pw, h->last_flags,
                                               h->last_self, st));
        if (tlm) TR_TRY(viorb_synth_local_points_device(h->fe, h->last_kps, h->last_count, synth_pose12, h->last_Pw, h->B, h->last_pts_f, st));
    }
    h->rolls++;
    return VIORB_OK;
}

extern "C" {

int viorb_tracker_create(const viorb_tracker_config* cfg, viorb_tracker** out) {
    VIORB_REQUIRE(cfg && out, "null cfg/out");
    VIORB_REQUIRE(cfg->batch >= 1 && cfg->width > 0 && cfg->height > 0, "batch >= 1, width, height > 0");
    VIORB_REQUIRE(cfg->track_local_map <= 0 || (cfg->local_frames >= 1 && cfg->local_frames <= 8), "1 <= local_frames <= 8");
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    VIORB_HIP_TRY(hipSetDevice(cfg->device));
    viorb_tracker* h = new viorb_tracker();
    struct Guard { viorb_tracker* h; ~Guard() { if (h) viorb_tracker_destroy(h); } } guard{h};
    h->cfg = *cfg; h->B = cfg->batch; h->device = cfg->device; h->nlevels = cfg->extractor.nlevels;
    for (int i = 0; i < 2; i++) TR_TRY(viorb_extractor_create(&cfg->extractor, cfg->batch, cfg->device, &h->ex[i]));
    TR_TRY(viorb_extractor_max_keypoints_for(h->ex[0], cfg->width, cfg->height, &h->cap));      // the pitch of the extractor's results for this image size
    viorb_frontend_config fc = cfg->frontend;
    float sf[16], is2[16];
    TR_TRY(viorb_extractor_tables(h->ex[0], sf, nullptr, nullptr, is2, nullptr));
    for (int i = 0; i < 16; i++) { fc.scale_factors[i] = sf[i < h->nlevels ? i : h->nlevels - 1]; fc.inv_level_sigma2[i] = is2[i < h->nlevels ? i : h->nlevels - 1]; }
    fc.nlevels = h->nlevels;
    {   // Frame::ComputeImageBounds (src/Frame.cc:616-644): the four undistorted corners when the camera is distorted
        const float intr4[4] = {fc.fx, fc.fy, fc.cx, fc.cy};
        float b4[4];
        TR_TRY(viorb_image_bounds(cfg->width, cfg->height, intr4, fc.dist_coef, b4));
        fc.min_x = b4[0]; fc.max_x = b4[1]; fc.min_y = b4[2]; fc.max_y = b4[3];
    }
    h->undistort = fc.dist_coef[0] != 0.0f;
    TR_TRY(viorb_frontend_create(&fc, cfg->batch, h->cap, cfg->device, &h->fe));
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);                  // hi = numerically lowest = highest priority
        const char* pr = getenv("VIORB_TRACK_PRIORITY");                  // experiment switch: 1 = tracking stream above extraction, 2 = below
        const int p_tr = pr && pr[0] == '1' ? hi : (pr && pr[0] == '2' ? lo : 0), p_ex = pr && pr[0] == '1' ? lo : (pr && pr[0] == '2' ? hi : 0);
        VIORB_HIP_TRY(hipStreamCreateWithPriority(&h->s_ex, hipStreamNonBlocking, p_ex));
        { const char* e2 = getenv("VIORB_TRACKER_EX_STREAMS"); if (e2 && atoi(e2) == 2) VIORB_HIP_TRY(hipStreamCreateWithPriority(&h->s_ex2, hipStreamNonBlocking, p_ex)); }
        VIORB_HIP_TRY(hipStreamCreateWithPriority(&h->s_tr, hipStreamNonBlocking, p_tr));
    }
    VIORB_HIP_TRY(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) { VIORB_HIP_TRY(hipEventCreateWithFlags(&h->ev_ex[i], hipEventDisableTiming)); VIORB_HIP_TRY(hipEventCreateWithFlags(&h->ev_tr[i], hipEventDisableTiming)); }
    const size_t B = (size_t)h->B, cap = (size_t)h->cap, R = (size_t)(cfg->track_local_map > 0 ? cfg->local_frames : 1);
    DevArena& M = h->mem;
    h->last_kps = M.get<viorb_keypoint>(B * cap); h->last_desc = M.get<uint8_t>(B * cap * 32); h->last_count = M.get<int>(B);
    h->last_flags = M.get<uint8_t>(B * cap); h->last_Pw = M.get<float>(B * cap * 3); h->last_pts_f = M.get<float>(B * cap * 8); h->last_self = M.get<int>(B * cap);
    h->last_ns = M.get<double>(B * 22); h->prior_ns = M.get<double>(B * 22); h->marg_cov_inv = M.get<double>(B * 144); h->t_last = M.get<double>(B);
    h->loc_pts_f = M.get<float>(B * R * cap * 8); h->loc_flags = M.get<uint8_t>(B * R * cap); h->loc_desc = M.get<uint8_t>(B * R * cap * 32); h->loc_count = M.get<int>(B);
    h->cell_start = M.get<int>(B * (64 * 48 + 1)); h->cell_idx = M.get<int>(B * cap); h->preint = M.get<double>(B * 142); h->cur_ns = M.get<double>(B * 22);
    h->pose12 = M.get<float>(B * 12); h->pose12_b = M.get<float>(B * 12); h->cur_match = M.get<int>(B * cap); h->nmatches = M.get<int>(B);
    h->status_s1 = M.get<int>(B); h->status_s2 = M.get<int>(B); h->status = M.get<int>(B);
    h->obs_cur = M.get<double>(B * cap * 6); h->obs_last = M.get<double>(B * cap * 6); h->obs_cur2 = M.get<double>(B * cap * 6);
    h->idx_cur = M.get<int>(B * cap); h->idx_last = M.get<int>(B * cap); h->idx_cur2 = M.get<int>(B * cap);
    h->n_cur = M.get<int>(B); h->n_last = M.get<int>(B); h->n_cur2 = M.get<int>(B);
    h->out_ns = M.get<double>(B * 22); h->out_last_ns = M.get<double>(B * 22); h->out_ns2 = M.get<double>(B * 22); h->ns1 = M.get<double>(B * 22);
    h->final_ns = M.get<double>(B * 22); h->final_marg = M.get<double>(B * 144); h->marg_out = M.get<double>(B * 144);
    h->info = M.get<double>(B * 4); h->info2 = M.get<double>(B * 4);
    h->outlier_cur = M.get<uint8_t>(B * cap); h->outlier_last = M.get<uint8_t>(B * cap); h->outlier_cur2 = M.get<uint8_t>(B * cap); h->owner_obs = M.get<uint8_t>(B * cap);
    h->skip1 = M.get<uint8_t>(B); h->skip2 = M.get<uint8_t>(B); h->variant = M.get<uint8_t>(B);
    if (h->undistort) h->kps_un = M.get<viorb_keypoint>(B * cap);
    h->n_map = M.get<int>(B); h->loc_match = M.get<int>(B * cap); h->n_loc = M.get<int>(B); h->state = M.get<int>(B); h->inliers = M.get<int>(B);
    if (M.err != hipSuccess) { set_error("device allocation failed: %s", hipGetErrorString(M.err)); return VIORB_ERR_HIP; }
    {
        std::vector<int> lc(B, (int)(R * cap));
        VIORB_HIP_TRY(hipMemcpy(h->loc_count, lc.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    }
    guard.h = nullptr;
    *out = h;
    return VIORB_OK;
}

int viorb_tracker_destroy(viorb_tracker* h) {
    if (!h) return VIORB_OK;
    (void)hipSetDevice(h->device);
    if (h->s_ex) (void)hipStreamSynchronize(h->s_ex);
    if (h->s_ex2) (void)hipStreamSynchronize(h->s_ex2);
    if (h->s_tr) (void)hipStreamSynchronize(h->s_tr);
    for (hipEvent_t e : h->in_flight) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    for (int i = 0; i < 2; i++) { if (h->ev_ex[i]) (void)hipEventDestroy(h->ev_ex[i]); if (h->ev_tr[i]) (void)hipEventDestroy(h->ev_tr[i]); }
    if (h->s_ex) (void)hipStreamDestroy(h->s_ex);
    if (h->s_ex2) (void)hipStreamDestroy(h->s_ex2);
    if (h->s_tr) (void)hipStreamDestroy(h->s_tr);
    if (h->s_cp) { (void)hipStreamSynchronize(h->s_cp); (void)hipStreamDestroy(h->s_cp); }
    for (uint8_t* p : h->stage) if (p) (void)hipFree(p);
    for (hipEvent_t e : h->ev_cp) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; i++) viorb_extractor_destroy(h->ex[i]);
    viorb_frontend_destroy(h->fe);
    h->mem.release();
    delete h;
    return VIORB_OK;
}

int viorb_tracker_capacity(const viorb_tracker* h, int* cap) {
    VIORB_REQUIRE(h && cap, "null argument");
    *cap = h->cap;
    return VIORB_OK;
}

int viorb_tracker_bootstrap(viorb_tracker* h, const uint8_t* d_images, int stride, size_t image_pitch_bytes, const double* d_ns0,
                            const double* d_t0, const double* d_marg_cov_inv, const double* d_synth_pose12, void* caller_stream) {
    VIORB_REQUIRE(h && d_images && d_ns0 && d_t0 && d_marg_cov_inv, "null argument");
    VIORB_HIP_TRY(hipSetDevice(h->device));
    VIORB_HIP_TRY(hipStreamSynchronize((hipStream_t)caller_stream));
    VIORB_HIP_TRY(hipStreamSynchronize(h->s_ex)); if (h->s_ex2) VIORB_HIP_TRY(hipStreamSynchronize(h->s_ex2)); VIORB_HIP_TRY(hipStreamSynchronize(h->s_tr));
    h->rolls = 0; h->k = 0; h->ev_tr_valid[0] = h->ev_tr_valid[1] = false; h->cur_slot = 0;
    TR_TRY(viorb_extract_batch_device(h->ex[0], d_images, h->B, h->cfg.width, h->cfg.height, stride, image_pitch_bytes, h->s_tr));
    const viorb_keypoint* kps; const uint8_t* desc; const int32_t* count; const int32_t* st; int cap;
    TR_TRY(viorb_extractor_results_device(h->ex[0], &kps, &desc, &count, &st, &cap));
    if (h->undistort) { TR_TRY(viorb_frontend_undistort_device(h->fe, kps, count, h->B, h->kps_un, h->s_tr)); kps = h->kps_un; }
    VIORB_HIP_TRY(hipMemcpyAsync(h->marg_cov_inv, d_marg_cov_inv, sizeof(double) * 144 * h->B, hipMemcpyDeviceToDevice, h->s_tr));
    TR_TRY(tracker_roll(h, kps, desc, count, d_ns0, d_t0, nullptr, d_synth_pose12, h->s_tr));
    VIORB_HIP_TRY(hipStreamSynchronize(h->s_tr));
    return VIORB_OK;
}

int viorb_tracker_set_last_points_device(viorb_tracker* h, const float* d_Pw, const uint8_t* d_flags, const float* d_pts_f, void* caller_stream) {
    VIORB_REQUIRE(h && d_Pw && d_flags, "null argument");
    VIORB_HIP_TRY(hipSetDevice(h->device));
    VIORB_HIP_TRY(hipEventRecord(h->ev_in, (hipStream_t)caller_stream));
    VIORB_HIP_TRY(hipStreamWaitEvent(h->s_tr, h->ev_in, 0));
    const size_t n = (size_t)h->B * h->cap;
    VIORB_HIP_TRY(hipMemcpyAsync(h->last_Pw, d_Pw, sizeof(float) * 3 * n, hipMemcpyDeviceToDevice, h->s_tr));
    VIORB_HIP_TRY(hipMemcpyAsync(h->last_flags, d_flags, n, hipMemcpyDeviceToDevice, h->s_tr));
    if (d_pts_f) VIORB_HIP_TRY(hipMemcpyAsync(h->last_pts_f, d_pts_f, sizeof(float) * 8 * n, hipMemcpyDeviceToDevice, h->s_tr));
    // last_self: keypoint i holds map point i where flags bit 0 is set — filled by a tiny kernel through the synth path is not available here:
    // the caller's flags decide; build_observations reads match = last_self, so write the identity / -1 on the host side of the stream
    return viorb_frontend_self_index_device(h->fe, h->last_flags, h->last_count, h->B, h->last_self, h->s_tr);
}

int viorb_tracker_step(viorb_tracker* h, const viorb_tracker_inputs* in, void* caller_stream) {
    VIORB_REQUIRE(h && in && (in->d_images || in->h_images) && in->d_imu && in->d_t_cur && in->n_imu >= 1, "null argument");
    VIORB_HIP_TRY(hipSetDevice(h->device));
    const auto t_begin = std::chrono::steady_clock::now();
    const int B = h->B, slot = (int)(h->k & 1);
    const uint8_t* d_images = in->d_images;
    hipEvent_t ev_upload = nullptr;
    if (in->h_images) {
        // Host -> device staging of a live feed (the reference's Frame constructor takes a host cv::Mat, src/Frame.cc:427-433): one
        // asynchronous copy per step on its own stream; buffer k % n is free again because the throttle below keeps the host at most
        // max_steps_ahead steps in front of the device.
        const int ahead_ = h->cfg.max_steps_ahead > 0 ? h->cfg.max_steps_ahead : 8;
        const size_t bytes = (size_t)B * in->image_pitch_bytes;
        if (h->stage.empty() || h->stage_bytes < bytes) {
            VIORB_HIP_TRY(hipDeviceSynchronize());
            for (uint8_t* p : h->stage) if (p) (void)hipFree(p);
            h->stage.assign((size_t)ahead_ + 2, nullptr); h->stage_bytes = 0;
            for (auto& p : h->stage) VIORB_HIP_TRY(hipMalloc(&p, bytes));
            h->stage_bytes = bytes;
            if (h->ev_cp.empty()) { h->ev_cp.assign(h->stage.size(), nullptr); for (auto& e : h->ev_cp) VIORB_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
            if (!h->s_cp) VIORB_HIP_TRY(hipStreamCreateWithFlags(&h->s_cp, hipStreamNonBlocking));
        }
        const size_t sl = (size_t)(h->k % (long long)h->stage.size());
        VIORB_HIP_TRY(hipMemcpyAsync(h->stage[sl], in->h_images, bytes, hipMemcpyHostToDevice, h->s_cp));
        VIORB_HIP_TRY(hipEventRecord(h->ev_cp[sl], h->s_cp));
        d_images = h->stage[sl]; ev_upload = h->ev_cp[sl];
        h->upload_bytes += (double)bytes;
    }
    const bool tlm = h->cfg.track_local_map > 0;
    const bool match_only = h->cfg.track_local_map < 0;       // extraction + grid + IMU prediction + SearchByProjection only: the frame hands on its IMU prediction
    const int marg = h->cfg.compute_marg != 0;
    viorb_extractor* ex = h->ex[slot];
    // inputs were produced on the caller's stream
    VIORB_HIP_TRY(hipEventRecord(h->ev_in, (hipStream_t)caller_stream));
    hipStream_t sx = (slot && h->s_ex2) ? h->s_ex2 : h->s_ex;
    VIORB_HIP_TRY(hipStreamWaitEvent(sx, h->ev_in, 0));
    VIORB_HIP_TRY(hipStreamWaitEvent(h->s_tr, h->ev_in, 0));
    if (h->ev_tr_valid[slot]) VIORB_HIP_TRY(hipStreamWaitEvent(sx, h->ev_tr[slot], 0));     // this handle's previous results have been consumed
    if (ev_upload) VIORB_HIP_TRY(hipStreamWaitEvent(sx, ev_upload, 0));
    TR_TRY(viorb_extract_batch_device(ex, d_images, B, h->cfg.width, h->cfg.height, in->image_stride, in->image_pitch_bytes, sx));
    VIORB_HIP_TRY(hipEventRecord(h->ev_ex[slot], sx));
    hipStream_t st = h->s_tr;
    // ---- what does not need the new frame's keypoints: IMU pre-integration + prediction, the last frame's own observations
    TR_TRY(viorb_frontend_imu_predict_device(h->fe, in->d_imu, in->n_imu, h->t_last, in->d_t_cur, h->last_ns, B, h->preint, h->cur_ns, h->pose12, st));
    TR_TRY(viorb_frontend_build_observations_device(h->fe, h->last_kps, h->last_count, h->last_self, h->last_Pw, B, h->obs_last, h->idx_last, h->n_last, st));
    VIORB_HIP_TRY(hipStreamWaitEvent(st, h->ev_ex[slot], 0));
    const viorb_keypoint* kps; const uint8_t* desc; const int32_t* count; const int32_t* ex_status; int cap;
    TR_TRY(viorb_extractor_results_device(ex, &kps, &desc, &count, &ex_status, &cap));
    h->cur_slot = slot;
    // Frame::UndistortKeyPoints (Frame.cc:171): grid, searches, edges and the frame handed on all read mvKeysUn
    if (h->undistort) { TR_TRY(viorb_frontend_undistort_device(h->fe, kps, count, B, h->kps_un, st)); kps = h->kps_un; }
    // ---- TrackWithIMU
    TR_TRY(viorb_frontend_grid_device(h->fe, kps, count, B, h->cell_start, h->cell_idx, st));
    TR_TRY(viorb_frontend_search_projection_device(h->fe, kps, desc, count, h->cell_start, h->cell_idx, h->pose12, h->last_kps, h->last_count, h->last_flags,
                                                   h->last_Pw, h->last_desc, h->cfg.th_projection, B, h->cur_match, h->nmatches, h->status_s1, st));
    TR_TRY(viorb_frontend_search_projection_retry_device(h->fe, kps, desc, count, h->cell_start, h->cell_idx, h->pose12, h->last_kps, h->last_count,
                                                         h->last_flags, h->last_Pw, h->last_desc, 2 * h->cfg.th_projection, 20, B, h->cur_match, h->nmatches,
                                                         h->status_s1, st));
    TR_TRY(viorb_frontend_build_observations_device(h->fe, kps, count, h->cur_match, h->last_Pw, B, h->obs_cur, h->idx_cur, h->n_cur, st));
    if (match_only) {
        // "ORB extract + match" (BASELINE north_star's single-stream figure): no pose solve; state = VIORB_TRACK_OK, the NavState handed on is the
        // IMU prediction (what Tracking does while vision is lost, src/Tracking.cc:1036-1114), the prior information stays
        VIORB_HIP_TRY(hipMemsetAsync(h->state, 0, sizeof(int32_t) * B, st));
        VIORB_HIP_TRY(hipMemcpyAsync(h->final_ns, h->cur_ns, sizeof(double) * 22 * B, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_track_merge_status, dim3((B + 255) / 256), dim3(256), 0, st, ex_status, h->status_s1, (const int32_t*)nullptr, B, h->status);
        VIORB_HIP_TRY(hipGetLastError());
        TR_TRY(tracker_roll(h, kps, desc, count, in->d_reset_ns ? in->d_reset_ns : h->final_ns, in->d_t_next_last ? in->d_t_next_last : in->d_t_cur, nullptr,
                            in->d_synth_pose12, st));
    } else {
    hipLaunchKernelGGL(k_track_gate0, dim3((B + 255) / 256), dim3(256), 0, st, h->nmatches, B, h->skip1, in->d_map_updated, h->variant);
    TR_TRY(viorb_frontend_pose_opt_select_device(h->fe, h->variant, h->skip1, marg && !tlm, h->cur_ns, h->last_ns, h->prior_ns, h->marg_cov_inv, h->preint,
                                                 h->obs_cur, h->n_cur, h->obs_last, h->n_last, B, h->out_ns, h->out_last_ns, h->outlier_cur,
                                                 h->outlier_last, h->marg_out, h->info, st));
    TR_TRY(viorb_frontend_discard_outliers_device(h->fe, h->cur_match, h->idx_cur, h->outlier_cur, h->n_cur, h->last_flags, B, h->owner_obs, h->n_map, st));
    hipLaunchKernelGGL(k_track_gate1, dim3(B), dim3(64), 0, st, h->nmatches, h->n_map, h->cur_ns, h->out_ns, B, h->state, h->ns1, h->skip2);
    if (tlm) {
        // ---- TrackLocalMapWithIMU
        TR_TRY(viorb_frontend_pose_from_navstate_device(h->fe, h->ns1, B, h->pose12_b, st));
        TR_TRY(viorb_frontend_search_local_points_device(h->fe, kps, desc, count, h->cell_start, h->cell_idx, h->pose12_b, h->loc_pts_f, h->loc_flags, h->loc_desc,
                                                         h->loc_count, h->cfg.local_frames * h->cap, 1.0f, 0.8f, h->owner_obs, B, h->loc_match, h->n_loc, nullptr,
                                                         h->status_s2, st));
        TR_TRY(viorb_frontend_build_observations2_device(h->fe, kps, count, h->cur_match, h->last_Pw, h->loc_match, h->loc_pts_f, h->cfg.local_frames * h->cap, B,
                                                         h->obs_cur2, h->idx_cur2, h->n_cur2, st));
        TR_TRY(viorb_frontend_pose_opt_select_device(h->fe, h->variant, h->skip2, marg, h->ns1, h->last_ns, h->prior_ns, h->marg_cov_inv, h->preint, h->obs_cur2,
                                                     h->n_cur2, h->obs_last, h->n_last, B, h->out_ns2, h->out_last_ns, h->outlier_cur2, h->outlier_last,
                                                     h->marg_out, h->info2, st));
        hipLaunchKernelGGL(k_track_count_inliers, dim3(B), dim3(256), 0, st, h->outlier_cur2, h->idx_cur2, h->n_cur2, h->cur_match, h->last_flags, h->loc_match,
                           h->loc_flags, h->cfg.local_frames * h->cap, h->cap, h->inliers);
    }
    hipLaunchKernelGGL(k_track_final, dim3(B), dim3(192), 0, st, h->state, h->inliers, in->d_recent_reloc, h->ns1, h->out_ns2, h->marg_out, h->marg_cov_inv,
                       in->d_reset_ns, in->d_reset_marg, tlm ? 1 : 0, marg, B, h->final_ns, h->final_marg);
    hipLaunchKernelGGL(k_track_merge_status, dim3((B + 255) / 256), dim3(256), 0, st, ex_status, h->status_s1, tlm ? h->status_s2 : nullptr, B, h->status);
    VIORB_HIP_TRY(hipGetLastError());
    // ---- mLastFrame = Frame(mCurrentFrame)
    TR_TRY(tracker_roll(h, kps, desc, count, h->final_ns, in->d_t_next_last ? in->d_t_next_last : in->d_t_cur, h->final_marg, in->d_synth_pose12, st));
    }
    VIORB_HIP_TRY(hipEventRecord(h->ev_tr[slot], st));
    h->ev_tr_valid[slot] = true;
    h->k++;
    // keep the host at most max_steps_ahead steps in front of the device (a live system never queues more: frames arrive one at a time)
    hipEvent_t ev;
    if (!h->ev_pool.empty()) { ev = h->ev_pool.back(); h->ev_pool.pop_back(); }
    else VIORB_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    VIORB_HIP_TRY(hipEventRecord(ev, st));
    h->in_flight.push_back(ev);
    const auto t_enq = std::chrono::steady_clock::now();
    const int ahead = h->cfg.max_steps_ahead > 0 ? h->cfg.max_steps_ahead : 8;
    while ((int)h->in_flight.size() > ahead) {
        hipEvent_t e = h->in_flight.front(); h->in_flight.pop_front();
        VIORB_HIP_TRY(hipEventSynchronize(e));
        h->ev_pool.push_back(e);
    }
    const auto t_end = std::chrono::steady_clock::now();
    h->enqueue_s += std::chrono::duration<double>(t_enq - t_begin).count();
    h->throttle_s += std::chrono::duration<double>(t_end - t_enq).count();
    h->steps++;
    return VIORB_OK;
}

int viorb_tracker_sync(viorb_tracker* h) {
    VIORB_REQUIRE(h, "null handle");
    VIORB_HIP_TRY(hipSetDevice(h->device));
    if (h->s_cp) VIORB_HIP_TRY(hipStreamSynchronize(h->s_cp));
    VIORB_HIP_TRY(hipStreamSynchronize(h->s_ex));
    if (h->s_ex2) VIORB_HIP_TRY(hipStreamSynchronize(h->s_ex2));
    VIORB_HIP_TRY(hipStreamSynchronize(h->s_tr));
    while (!h->in_flight.empty()) { h->ev_pool.push_back(h->in_flight.front()); h->in_flight.pop_front(); }
    return VIORB_OK;
}

int viorb_tracker_results_device(const viorb_tracker* h, viorb_tracker_results* r) {
    VIORB_REQUIRE(h && r, "null argument");
    r->cap = h->cap;
    r->state = h->state; r->status = h->status; r->nmatches = h->nmatches; r->n_map = h->n_map; r->n_loc = h->n_loc; r->inliers = h->inliers;
    r->info = h->info; r->info2 = h->info2; r->pred_ns = h->cur_ns; r->ns_stage1 = h->out_ns; r->ns_stage2 = h->out_ns2; r->final_ns = h->final_ns;
    r->final_marg = h->final_marg; r->cur_match = h->cur_match; r->loc_match = h->loc_match; r->outlier_cur = h->outlier_cur; r->outlier_cur2 = h->outlier_cur2;
    r->n_obs = h->n_cur; r->n_obs2 = h->n_cur2; r->last_ns = h->last_ns; r->last_Pw = h->last_Pw; r->last_pts_f = h->last_pts_f; r->last_flags = h->last_flags; r->last_count = h->last_count;
    r->extractor = h->ex[h->cur_slot];
    return VIORB_OK;
}

int viorb_tracker_host_stats(viorb_tracker* h, double* enqueue_s, double* throttle_s, long long* steps, int reset) {
    VIORB_REQUIRE(h, "null handle");
    if (enqueue_s) *enqueue_s = h->enqueue_s;
    if (throttle_s) *throttle_s = h->throttle_s;
    if (steps) *steps = h->steps;
    if (reset) { h->enqueue_s = h->throttle_s = 0; h->steps = 0; }
    return VIORB_OK;
}

int viorb_memcpy_dtoh(void* dst_host, const void* src_device, size_t bytes) {
    VIORB_REQUIRE(dst_host && src_device, "null pointer");
    VIORB_HIP_TRY(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return VIORB_OK;
}
int viorb_memcpy_htod(void* dst_device, const void* src_host, size_t bytes) {
    VIORB_REQUIRE(dst_device && src_host, "null pointer");
    VIORB_HIP_TRY(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
    return VIORB_OK;
}

} // extern "C"
