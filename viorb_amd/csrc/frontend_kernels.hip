// viorb_amd/csrc/frontend_kernels.hip — the per-frame tracking kernels behind the extractor, batched
// over independent camera streams (one workgroup per stream):
//   k_frame_grid          Frame::AssignFeaturesToGrid            reference src/Frame.cc:410-425,562-572
//   k_search_projection   ORBmatcher::SearchByProjection(F,F)    reference src/ORBmatcher.cc:1328-1471
//   k_imu_predict         IMUPreintegrator::update xN, Converter::updateNS, Frame::UpdatePoseFromNS
//                         reference src/IMU/IMUPreintegrator.cpp:86-153, src/Frame.cc:41-105
//   k_build_observations  the edge-construction loops of PoseOptimization  src/Optimizer.cc:493-589
//   k_pose_opt_vi_mp      Optimizer::PoseOptimization(Frame*, KeyFrame*|Frame*, ...) with g2o's LM (pose_opt_mp.inc)
//                         reference src/Optimizer.cc:323-1112 + Thirdparty/g2o (see vio_core.h)
#include <hip/hip_runtime.h>
#include <vector>
#include <atomic>
#include <algorithm>
#include "viorb_common.h"
#include "orb_math.h"
#include "vio_core.h"

namespace viorb {

enum { GRID_COLS = 64, GRID_ROWS = 48, GRID_CELLS = GRID_COLS * GRID_ROWS, TH_HIGH = 100, HISTO_LENGTH = 30, CAND_CAP = 128 };

// ---------------------------------------------------------------------------------------------
// Frame grid as CSR. Cell index = ix*48 + iy (the reference's mGrid[ix][iy] storage order), entries
// of a cell in keypoint-index order (push_back order). Built by sorting (cell << 16 | index).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_bitonic_sort(uint32_t* a, int m) {
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (m >> 1); t += blockDim.x) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const uint32_t x = a[lo], y = a[hi];
                if ((x > y) == up) { a[lo] = y; a[hi] = x; }
            }
            __syncthreads();
        }
}

// Frame::UndistortKeyPoints (reference src/Frame.cc:584-614): cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK) of OpenCV 2.4
// (cvUndistortPoints, modules/imgproc/src/undistort.cpp): camera matrix and coefficients widened to double, per point the normalised
// coordinates, FIVE fixed-point iterations of the radial-tangential model, re-projection with RR = K * I in homogeneous form, one
// rounding to float. Operation order as published (no contraction: the library is built with -ffp-contract=off). One thread per keypoint record: the record is copied with pt replaced (mvKeysUn[i] = mvKeys[i]
// with kp.pt changed). n == nullptr: a flat array of `cap` records (the host-buffer forms).
struct UndistortArgs { double fx, fy, cx, cy, k[5]; };
__device__ __forceinline__ void undistort_pt(const UndistortArgs& A, float u, float v, float* ou, float* ov) {
    const double ifx = 1. / A.fx, ify = 1. / A.fy;
    double x = u, y = v;
    const double x0 = x = (x - A.cx) * ifx;
    const double y0 = y = (y - A.cy) * ify;
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1. / (1 + ((A.k[4] * r2 + A.k[1]) * r2 + A.k[0]) * r2);     // numerator 1 + ((k6 r2 + k5) r2 + k4) r2 with k4 = k5 = k6 = 0: exactly 1
        const double deltaX = 2 * A.k[2] * x * y + A.k[3] * (r2 + 2 * x * x);
        const double deltaY = A.k[2] * (r2 + 2 * y * y) + 2 * A.k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    // RR = K: xx = fx*x + 0*y + cx, yy = 0*x + fy*y + cy, ww = 1./(0*x + 0*y + 1) = 1 — the zero products are +-0 and change no finite sum
    *ou = (float)(A.fx * x + A.cx); *ov = (float)(A.fy * y + A.cy);
}
__global__ __launch_bounds__(256) void k_undistort(const viorb_keypoint* __restrict__ kps, const int* __restrict__ count, int cap, UndistortArgs A,
                                                   int enabled, viorb_keypoint* __restrict__ out) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = count ? min(count[b], cap) : cap;
    if (i >= n) return;
    viorb_keypoint kp = kps[(size_t)b * cap + i];
    if (enabled) undistort_pt(A, kp.x, kp.y, &kp.x, &kp.y);
    out[(size_t)b * cap + i] = kp;
}

__global__ __launch_bounds__(256) void k_frame_grid(const viorb_keypoint* __restrict__ kps, const int* __restrict__ count, int cap,
                                                    float minX, float minY, float wInv, float hInv,
                                                    int* __restrict__ cell_start, int* __restrict__ cell_idx, int sort_n) {
    extern __shared__ uint32_t s_keys[];
    const int b = blockIdx.x;
    const int n = min(count[b], cap);
    const viorb_keypoint* kp = kps + (size_t)b * cap;
    for (int i = threadIdx.x; i < sort_n; i += blockDim.x) {
        uint32_t key = 0xffffffffu;
        if (i < n) {
            // PosInGrid: round() = half away from zero
            const int px = (int)roundf((kp[i].x - minX) * wInv), py = (int)roundf((kp[i].y - minY) * hInv);
            if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) key = ((uint32_t)(px * GRID_ROWS + py) << 16) | (uint32_t)i;
        }
        s_keys[i] = key;
    }
    __syncthreads();
    block_bitonic_sort(s_keys, sort_n);
    int* cs = cell_start + (size_t)b * (GRID_CELLS + 1);
    int* ci = cell_idx + (size_t)b * cap;
    // entries: sorted position p holds keypoint (key & 0xffff) of cell (key >> 16)
    for (int p = threadIdx.x; p < sort_n; p += blockDim.x) {
        const uint32_t key = s_keys[p];
        const int cell = key == 0xffffffffu ? GRID_CELLS : (int)(key >> 16);
        if (key != 0xffffffffu && p < cap) ci[p] = (int)(key & 0xffff);
        const int prev_cell = p == 0 ? -1 : (s_keys[p - 1] == 0xffffffffu ? GRID_CELLS : (int)(s_keys[p - 1] >> 16));
        // every cell in (prev_cell, cell] starts at p
        for (int c = prev_cell + 1; c <= cell && c <= GRID_CELLS; c++) cs[c] = p;
        if (p == sort_n - 1) for (int c = cell + 1; c <= GRID_CELLS; c++) cs[c] = sort_n;   // only when all slots are real entries
    }
}

// ---------------------------------------------------------------------------------------------
// SearchByProjection(CurrentFrame, LastFrame, th, mono)
//   phase A (parallel over last-frame points): project, walk the grid window in the reference's
//     order (ix outer, iy inner, insertion order inside a cell), store (candidate, Hamming) lists;
//   phase B: the reference's greedy loop "for i in order: take the closest candidate not yet owned by
//     a point with observations" is reproduced by a parallel fixed-point iteration — after k sweeps
//     the choices of the first k points are final, sweeps stop when nothing changes;
//   phase C: rotation histogram (the reference's factor 1/30 binning), three maxima, rejection.
// last_flags: bit0 has map point, bit1 outlier, bit2 the point has observations.
// ---------------------------------------------------------------------------------------------
struct SearchArgs {
    const viorb_keypoint* cur_kps; const uint8_t* cur_desc; const int* cur_count;
    const int* cell_start; const int* cell_idx;
    const float* pose12; const viorb_keypoint* last_kps; const int* last_count; const uint8_t* last_flags;
    const float* last_Pw; const uint8_t* last_desc;
    int* cur_match; int* nmatches; int* status;
    uint32_t* cand;           // [B][CAND_CAP][cap] = dist<<16 | idx (SearchByProjection; entry k of point i at k * cap + i), [B][CAND_CAP][pcap] = dist << 20 | level << 16 | idx (SearchLocalPoints)
    int* cand_n;              // [B][cap]
    int cap;
    int slot_n;               // candidates per point cached in LDS: SEARCH_SLOT, or 0 when the frame's arrays leave no room (cap > ~2400)
    unsigned char* work; size_t work_bytes;   // k_search_projection<true>: the per-stream work arrays in global memory (a frame with more keypoints than LDS holds)
    float minX, maxX, minY, maxY, wInv, hInv, fx, fy, cx, cy, th;
    float scale[16];
    int check_ori;
    int skip_if_at_least;     // > 0: leave streams whose nmatches[b] is already >= this untouched (TrackWithIMU's retry with 2*th)
    // stereo / RGB-D branch (bMono == false, reference src/ORBmatcher.cc:1346-1349, 1385-1410); all NULL / 0 for the monocular call
    const float* cur_uright;  // [B][cap] CurrentFrame.mvuRight (<= 0: monocular keypoint)
    const float* last_pose12; // [B][12] LastFrame.mTcw as Rlw(9) tlw(3)
    float bf, mb;             // CurrentFrame.mbf, CurrentFrame.mb
};

// LDS plan (dynamic, per workgroup): the current frame's grid (u16 CSR), keypoint x/y/octave/angle, the first
// SLOT candidates of every last-frame point, and the four per-keypoint work arrays. Candidates beyond SLOT
// spill to the global scratch list.
#ifndef SEARCH_SLOT
#define SEARCH_SLOT 8       // a multiple of 4 (the candidate lists are read four entries at a time, from one address space)
#endif
__host__ __device__ inline size_t search_lds_bytes(int cap, int slot_n = SEARCH_SLOT) {
    return (size_t)(GRID_CELLS + 1) * 2 + 2 /*pad*/ + (size_t)cap * (8 + 4 + 4) + (size_t)cap * slot_n * 4 + (size_t)cap * 4 * 4 + 64;
}
// the slot cache is dropped (every candidate goes through the global list) when the arrays would not fit LDS with it
__host__ inline int search_slot_n(int cap) { return search_lds_bytes(cap, SEARCH_SLOT) <= 160 * 1024 ? SEARCH_SLOT : 0; }

// 1024 threads per stream: the kernel is a chain of LDS / L2 latencies per point (grid walk, descriptor fetch), so it wants every
// point of a ~1000-point frame on its own thread and 16 waves per CU to hide them (256 threads: 0.32 ms per 256 streams)
#ifndef SEARCH_THREADS
#define SEARCH_THREADS 1024
#endif
#ifndef SCAN_W
#define SCAN_W 4            // entries of a grid column tested per step of the window scan
#endif
// -DVIORB_SEARCH_TIMING: thread 0 of workgroup 0 prints the s_memtime ticks (100 MHz) of each phase of the two projection searches
#ifdef VIORB_SEARCH_TIMING
#define SPT_DECL unsigned long long spt_acc[6] = {0, 0, 0, 0, 0, 0}, spt_t0 = __builtin_amdgcn_s_memtime(); int spt_sweeps = 0
#define SPT_LAP(k) do { const unsigned long long spt_now = __builtin_amdgcn_s_memtime(); spt_acc[k] += spt_now - spt_t0; spt_t0 = spt_now; } while (0)
#define SPT_PRINT(name) do { if (threadIdx.x == 0 && blockIdx.x == 0) printf("SPT %s stage %llu walk %llu dist %llu sweeps %llu out %llu nsweeps %d\n", name, spt_acc[0], spt_acc[4], spt_acc[1], spt_acc[2], spt_acc[3], spt_sweeps); } while (0)
#else
#define SPT_DECL
#define SPT_LAP(k)
#define SPT_PRINT(name)
#endif
// GW = false: the work arrays in LDS (every frame of up to viorb_frontend_search_capacity() keypoints); GW = true: the same arrays in a
// per-stream slice of global memory — same code, same result, for frames LDS cannot hold (the reference has no limit)
template <bool GW>
__global__ __launch_bounds__(SEARCH_THREADS) void k_search_projection(SearchArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_lds[];
    unsigned char* s_raw = GW ? A.work + (size_t)blockIdx.x * A.work_bytes : s_lds;
    const int b = blockIdx.x, cap = A.cap, t = threadIdx.x, lane = t & 63;
    if (A.skip_if_at_least > 0 && A.nmatches[b] >= A.skip_if_at_least) return;       // uniform per workgroup, before any barrier
    const int ncur = min(A.cur_count[b], cap), nlast = min(A.last_count[b], cap);
    SPT_DECL;
    // carve LDS (4-byte aligned sections first)
    const int slot_n = A.slot_n;
    uint32_t* slot = reinterpret_cast<uint32_t*>(s_raw);                 // [slot_n][cap] dist<<16 | idx: entry k of point i at k * cap + i (lanes = points: conflict-free)
    int* choice = reinterpret_cast<int*>(slot + (size_t)cap * slot_n);   // [cap] chosen current keypoint of last point i, or -1 (touched by the thread of point i only, until phase C)
    int* taken = choice + cap;            // [cap] per current keypoint: smallest i (with observations) that chose it
    int* owner = taken + cap;             // [cap] the second buffer of `taken` during the sweeps; phase C: largest i that chose the keypoint
    int* rej = owner + cap;               // [cap] phase C: per current keypoint, chosen by a point of a rejected bin
    float2* pxy = reinterpret_cast<float2*>(rej + cap);                  // [cap] position of the keypoint at CSR entry p (the walk reads entries, not keypoints: no index hop)
    float* cang = reinterpret_cast<float*>(pxy + cap);                   // [cap] by keypoint index
    uint32_t* pio = reinterpret_cast<uint32_t*>(cang + cap);             // [cap] keypoint index | octave << 16 of CSR entry p
    uint16_t* cs = reinterpret_cast<uint16_t*>(pio + cap);               // [GRID_CELLS + 1] (+1 pad)
    __shared__ int s_changed[2], s_hist[HISTO_LENGTH], s_keep[HISTO_LENGTH], s_nm;
#ifdef VIORB_SEARCH_TIMING
    __shared__ int s_spt[4];        // [2], [3]: the slowest and the fastest wave's ticks in the walk
    if (t < 4) s_spt[t] = t == 3 ? 0x7fffffff : 0;
#endif
    const float* P = A.pose12 + (size_t)b * 12;
    const viorb_keypoint* ck = A.cur_kps + (size_t)b * cap;
    const viorb_keypoint* lk = A.last_kps + (size_t)b * cap;
    const uint8_t* lf = A.last_flags + (size_t)b * cap;
    uint32_t* cand = A.cand + (size_t)b * cap * CAND_CAP;
    int* cand_n = A.cand_n + (size_t)b * cap;
    // stereo: does the camera move forward / backward by more than the baseline? (tlc = Rlw * twc + tlw, twc = -Rcw^T tcw; :1339-1349)
    int motion = 0;
    const float* cur_ur = A.cur_uright ? A.cur_uright + (size_t)b * cap : nullptr;
    if (A.last_pose12) {
        const float* Lp = A.last_pose12 + (size_t)b * 12;
        float twc[3];
#pragma unroll
        for (int r = 0; r < 3; r++) twc[r] = -(P[r] * P[9] + P[3 + r] * P[10] + P[6 + r] * P[11]);
        const float tz = (Lp[6] * twc[0] + Lp[7] * twc[1] + Lp[8] * twc[2]) + Lp[11];
        motion = tz > A.mb ? 1 : (-tz > A.mb ? 2 : 0);
    }
    // The search window of last-frame point i (:1351-1395): projection, radius by octave, level range, grid window. Needs no LDS: the
    // window of a thread's first point is computed BEFORE the frame is staged, so its chain of dependent global loads (flags ->
    // position -> octave -> scale factor) runs beside the staging loads instead of after the barrier.
    struct Window { float u, v, invz, radius; int x0, x1, y0, y1, minL, maxL; bool ok; };
    auto project = [&](int i) -> Window {
        Window W; W.ok = false;
        W.u = W.v = W.invz = W.radius = 0.0f; W.x0 = W.y0 = 0; W.x1 = W.y1 = -1; W.minL = 0; W.maxL = -1;
        const int fl = lf[i];
        if ((fl & 1) && !(fl & 2)) {
            const float* X = A.last_Pw + ((size_t)b * cap + i) * 3;
            float pc[3];
#pragma unroll
            for (int r = 0; r < 3; r++) { const float tt = P[3 * r] * X[0] + P[3 * r + 1] * X[1] + P[3 * r + 2] * X[2]; pc[r] = tt + P[9 + r]; }
            const float invz = 1.0f / pc[2];
            const float u = A.fx * pc[0] * invz + A.cx, v = A.fy * pc[1] * invz + A.cy;
            if (!(invz < 0) && !(u < A.minX || u > A.maxX) && !(v < A.minY || v > A.maxY)) {
                const int oct = lk[i].octave;
                const float radius = A.th * A.scale[oct];
                // bForward: levels >= octave; bBackward: levels <= octave; else octave +- 1 (:1385-1390)
                W.minL = motion == 1 ? oct : (motion == 2 ? 0 : oct - 1); W.maxL = motion == 1 ? -1 : (motion == 2 ? oct : oct + 1);
                W.x0 = max(0, (int)floorf((u - A.minX - radius) * A.wInv));
                W.x1 = min((int)GRID_COLS - 1, (int)ceilf((u - A.minX + radius) * A.wInv));
                W.y0 = max(0, (int)floorf((v - A.minY - radius) * A.hInv));
                W.y1 = min((int)GRID_ROWS - 1, (int)ceilf((v - A.minY + radius) * A.hInv));
                W.u = u; W.v = v; W.invz = invz; W.radius = radius;
                W.ok = W.x0 < GRID_COLS && W.x1 >= 0 && W.y0 < GRID_ROWS && W.y1 >= 0;
            }
        }
        return W;
    };
    Window W0; W0.ok = false;
    if (t < nlast) W0 = project(t);
    {
        const int* gcs = A.cell_start + (size_t)b * (GRID_CELLS + 1);
        const int* gci = A.cell_idx + (size_t)b * cap;
        for (int i = t; i <= GRID_CELLS; i += blockDim.x) cs[i] = (uint16_t)gcs[i];
        for (int i = t; i < ncur; i += blockDim.x) cang[i] = ck[i].angle;
        const int ngrid = min(gcs[GRID_CELLS], ncur);                    // entries of the CSR (keypoints outside the grid have none)
        for (int p = t; p < ngrid; p += blockDim.x) {
            const int i2 = gci[p];
            const viorb_keypoint k = ck[i2];
            pxy[p] = make_float2(k.x, k.y); pio[p] = (uint32_t)i2 | ((uint32_t)(k.octave & 0xff) << 16);
        }
        for (int p = ngrid + t; p < cap; p += blockDim.x) { pxy[p] = make_float2(0.0f, 0.0f); pio[p] = 0; }   // read (and masked) by the clamped loads of the walk
    }
    if (t == 0) s_nm = 0;
    for (int i = t; i < HISTO_LENGTH; i += blockDim.x) { s_hist[i] = 0; s_keep[i] = 0; }
    __syncthreads();
    SPT_LAP(0);
    // Candidates of the window in the reference's order (GetFeaturesInArea: columns outer, rows inner, insertion order inside a cell):
    // g(k, index). Cells (ix, y0..y1) of a column are contiguous in the CSR, so a column is one range of entries. ONE loop with a
    // straight-line body: a step tests up to FOUR entries of the current column and reads the bounds of the next; every LDS read of
    // the step is issued up front (clamped addresses, results masked), the state moves by selects — a wave-step costs one LDS round
    // trip whatever its lanes are doing, and a column of <= 4 entries is one step. (Written as column loop / entry loop, each entry
    // cost three dependent LDS reads and the wave ran the sum over columns of the per-column maxima.)
    auto scan = [&](const Window& W, auto&& g) -> int {
        int nc = 0;
        if (!W.ok) return 0;
        const float u = W.u, v = W.v, radius = W.radius;
        const int x1 = W.x1, y0 = W.y0, y1 = W.y1, minL = W.minL, maxL = W.maxL;
        const bool check_levels = (minL > 0) || (maxL >= 0);
        int ix = W.x0, p = 0, pend = 0;
        for (;;) {
            const int col = min(ix, x1) * GRID_ROWS;
            const int cbeg = cs[col + y0], cend = cs[col + y1 + 1];
            uint32_t e[SCAN_W]; float2 q[SCAN_W];
#pragma unroll
            for (int j = 0; j < SCAN_W; j++) { const int pp = min(p + j, cap - 1); e[j] = pio[pp]; q[j] = pxy[pp]; }
#pragma unroll
            for (int j = 0; j < SCAN_W; j++) {
                const int i2 = (int)(e[j] & 0xffff), o2 = (int)(e[j] >> 16);
                bool pass = p + j < pend && fabsf(q[j].x - u) < radius && fabsf(q[j].y - v) < radius;
                if (check_levels) pass = pass && !(o2 < minL) && !(maxL >= 0 && o2 > maxL);
                if (cur_ur && pass) {                                      // "if(CurrentFrame.mvuRight[i2]>0)" (:1404-1410)
                    const float r2 = cur_ur[i2];
                    if (r2 > 0) { const float ur = u - A.bf * W.invz; if (fabsf(ur - r2) > radius) pass = false; }
                }
                if (pass) { g(nc, i2); nc++; }
            }
            const bool adv = p + SCAN_W >= pend;                                // the column is done: enter the next, or stop after the last
            if (adv && ix > x1) break;
            p = adv ? cbeg : p + SCAN_W;
            pend = adv ? cend : pend;
            ix += adv ? 1 : 0;
        }
        return nc;
    };
    // the same with each candidate's Hamming distance, f(k, dist << 16 | index): only for a point with more than CAND_CAP candidates
    // (a window of half the image), which every sweep of phase B enumerates again instead of cutting the list short — no capacity limit
    auto enumerate = [&](int i, auto&& f) -> int {
        const uint4* dl = reinterpret_cast<const uint4*>(A.last_desc + ((size_t)b * cap + i) * 32);
        const uint4 da = dl[0], db = dl[1];
        return scan(project(i), [&](int k, int i2) {
            const uint4* dc = reinterpret_cast<const uint4*>(A.cur_desc + ((size_t)b * cap + i2) * 32);
            const uint4 ea = dc[0], eb = dc[1];
            const int dist = __popc(da.x ^ ea.x) + __popc(da.y ^ ea.y) + __popc(da.z ^ ea.z) + __popc(da.w ^ ea.w) +
                             __popc(db.x ^ eb.x) + __popc(db.y ^ eb.y) + __popc(db.z ^ eb.z) + __popc(db.w ^ eb.w);
            f(k, ((uint32_t)dist << 16) | (uint32_t)i2);
        });
    };
    // stored candidate k of point i: the first slot_n in LDS, the rest in the stream's global list; both with lanes (= points) adjacent
    auto set_entry = [&](int i, int k, uint32_t e) { if (k < slot_n) slot[(size_t)k * cap + i] = e; else cand[(size_t)k * cap + i] = e; };
    // four of them from k0 (a multiple of 4, like slot_n: one address space per call), indices clamped to n - 1
    auto entries4 = [&](int i, int k0, int n, uint32_t* e) {
        if (k0 < slot_n) {
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = slot[(size_t)min(k0 + u, n - 1) * cap + i];
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = cand[(size_t)min(k0 + u, n - 1) * cap + i];
        }
    };
    // ---- phase A, in two passes per point: (1) the scan collects the candidates' indices (LDS reads only); (2) their descriptors are
    // fetched EIGHT at a time (sixteen 16-byte loads in flight) and the Hamming distances written beside the indices
#ifdef VIORB_SEARCH_TIMING
    const unsigned long long spt_w0 = __builtin_amdgcn_s_memtime();
#endif
    for (int i = t; i < nlast; i += blockDim.x) {
        const Window W = i == t ? W0 : project(i);
        const int nc = scan(W, [&](int k, int i2) { if (k < CAND_CAP) set_entry(i, k, (uint32_t)i2); });   // the true count, also beyond CAND_CAP
        cand_n[i] = nc;
        choice[i] = -1;
    }
#ifdef VIORB_SEARCH_TIMING
    { const int dt = (int)(__builtin_amdgcn_s_memtime() - spt_w0); if (lane == 0) { atomicMax(&s_spt[2], dt); atomicMin(&s_spt[3], dt); } }
    __syncthreads();
    SPT_LAP(4);
#endif
    for (int i = t; i < nlast; i += blockDim.x) {                    // (every thread reads back its own points' lists: no barrier)
        const int nc = cand_n[i];
        const int ns = min(nc, (int)CAND_CAP);
        if (ns > 0) {
            const uint4* dl = reinterpret_cast<const uint4*>(A.last_desc + ((size_t)b * cap + i) * 32);
            const uint4 da = dl[0], db = dl[1];
            for (int k0 = 0; k0 < ns; k0 += 8) {
                uint32_t id[8]; uint4 ea[8], eb[8];
                entries4(i, k0, ns, id);
                if (k0 + 4 < ns) entries4(i, k0 + 4, ns, id + 4);       // (never an index from a list position that was not written)
                else { id[4] = id[0]; id[5] = id[0]; id[6] = id[0]; id[7] = id[0]; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint4* dc = reinterpret_cast<const uint4*>(A.cur_desc + ((size_t)b * cap + (id[u] & 0xffff)) * 32);
                    ea[u] = dc[0]; eb[u] = dc[1];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (k0 + u >= ns) break;
                    const int dist = __popc(da.x ^ ea[u].x) + __popc(da.y ^ ea[u].y) + __popc(da.z ^ ea[u].z) + __popc(da.w ^ ea[u].w) +
                                     __popc(db.x ^ eb[u].x) + __popc(db.y ^ eb[u].y) + __popc(db.z ^ eb[u].z) + __popc(db.w ^ eb[u].w);
                    set_entry(i, k0 + u, ((uint32_t)dist << 16) | id[u]);
                }
            }
        }
    }
    SPT_LAP(1);
    // ---- phase B: fixed-point sweeps. `taken` is double-buffered (one buffer holds this sweep's owners, the other is cleared for the
    // next while this one is read) and choice[i] belongs to thread i: two barriers per sweep.
    for (int c = t; c < ncur; c += blockDim.x) taken[c] = 0x7fffffff;
    if (t < 2) s_changed[t] = 0;
    __syncthreads();
    int cur = 0;
    for (int sweep = 0; sweep <= nlast; sweep++, cur ^= 1) {
#ifdef VIORB_SEARCH_TIMING
        spt_sweeps++;
#endif
        int* tkc = cur ? owner : taken; int* tkn = cur ? taken : owner;
        for (int i = t; i < nlast; i += blockDim.x)
            if (choice[i] >= 0 && (lf[i] & 4)) atomicMin(&tkc[choice[i]], i);
        __syncthreads();
        bool changed = false;
        for (int i = t; i < nlast; i += blockDim.x) {
            const int nc = cand_n[i];
            int best = 256, bidx = -1;
            auto consider = [&](int, uint32_t e) {
                const int i2 = (int)(e & 0xffff), dist = (int)(e >> 16);
                if (tkc[i2] < i) return;                          // owned by an earlier point that has observations
                if (dist < best) { best = dist; bidx = i2; }
            };
            if (nc <= CAND_CAP) {
                for (int k0 = 0; k0 < nc; k0 += 4) {               // four entries, then their four owners, in flight together; the
                    uint32_t e[4]; int ow[4];                      // repeats of the last entry past nc change nothing (strict <)
                    entries4(i, k0, nc, e);
#pragma unroll
                    for (int u = 0; u < 4; u++) ow[u] = tkc[e[u] & 0xffff];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int dist = (int)(e[u] >> 16);
                        if (!(ow[u] < i) && dist < best) { best = dist; bidx = (int)(e[u] & 0xffff); }
                    }
                }
            } else enumerate(i, consider);                        // more candidates than the stored list holds: walk the grid again
            const int nw = best <= TH_HIGH ? bidx : -1;
            changed = changed || (nw != choice[i]);
            choice[i] = nw;
        }
        for (int c = t; c < ncur; c += blockDim.x) tkn[c] = 0x7fffffff;
        if (t == 0) s_changed[cur ^ 1] = 0;
        if (__any(changed) && lane == 0) s_changed[cur] = 1;
        __syncthreads();
        if (!s_changed[cur]) break;
    }
    SPT_LAP(2);
    // ---- phase C: histogram, maxima, final ownership
    for (int c = t; c < ncur; c += blockDim.x) { owner[c] = -1; rej[c] = 0; }
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    {
        int my_n = 0;
        for (int i = t; i < nlast; i += blockDim.x) {
            const int c = choice[i];
            int bin = -1;
            if (c >= 0) {
                atomicMax(&owner[c], i);
                my_n++;
                if (A.check_ori) {
                    float rot = lk[i].angle - cang[c];
                    if (rot < 0.0f) rot += 360.0f;
                    bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                }
            }
            taken[i] = bin;                                        // remember the bin of point i (taken[] is free now)
            if (A.check_ori) {
                // wave-aggregated histogram: one atomic per distinct bin per wave
                for (int hb = 0; hb < 13; hb++) { const unsigned long long m = __ballot(bin == hb); if (m && lane == (int)__ffsll((long long)m) - 1) atomicAdd(&s_hist[hb], __popcll(m)); }
                const unsigned long long mo = __ballot(bin > 12);   // bins above 12 cannot occur with factor 1/30; kept for safety
                if (mo && bin > 12) atomicAdd(&s_hist[bin], 1);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) my_n += __shfl_xor(my_n, d);
        if (lane == 0) atomicAdd(&s_nm, my_n);
    }
    __syncthreads();
    if (A.check_ori) {
        if (t == 0) {            // ComputeThreeMaxima
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < HISTO_LENGTH; i++) {
                const int sz = s_hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
            int removed = 0;
            for (int i = 0; i < HISTO_LENGTH; i++) { const int keep = (i == ind1 || i == ind2 || i == ind3); s_keep[i] = keep; if (!keep) removed += s_hist[i]; }
            s_nm -= removed;
        }
        __syncthreads();
        for (int i = t; i < nlast; i += blockDim.x) {
            const int c = choice[i];
            if (c >= 0 && !s_keep[taken[i]]) rej[c] = 1;
        }
        __syncthreads();
    }
    int* out = A.cur_match + (size_t)b * cap;
    for (int c = t; c < cap; c += blockDim.x) out[c] = (c < ncur && !rej[c]) ? owner[c] : -1;
    if (t == 0) { A.nmatches[b] = s_nm; A.status[b] = VIORB_OK; }
    SPT_LAP(3);
    SPT_PRINT("projection");
#ifdef VIORB_SEARCH_TIMING
    if (t == 0 && b == 0) printf("SPT walk: %d wave-iterations %d lane-iterations, slowest wave %d fastest %d ticks, nlast %d ncur %d\n", s_spt[0], s_spt[1], s_spt[2], s_spt[3], nlast, ncur);
#endif
}

// ---------------------------------------------------------------------------------------------
// Tracking::SearchLocalPoints: Frame::isInFrustum for every local map point (reference src/Frame.cc:449-505,
// MapPoint::PredictScale src/MapPoint.cc:408-424) followed by ORBmatcher::SearchByProjection(F, vpMapPoints, th)
// (reference src/ORBmatcher.cc:45-129): best / second-best Hamming among unowned grid candidates of levels
// [lvl-1, lvl], same-level ratio test, greedy ownership in point order (same fixed-point scheme as above).
// pts_f[i] = Pw3 normal3 minDist maxDist; pts_flags bit0 valid, bit1 skip (already matched in this frame),
// bit2 has observations. cur_owner_obs[c] != 0: keypoint c already holds a map point with observations.
// ---------------------------------------------------------------------------------------------
#define LOCAL_SLOT 4
struct LocalSearchArgs {
    const viorb_keypoint* cur_kps; const uint8_t* cur_desc; const int* cur_count;
    const int* cell_start; const int* cell_idx; const float* pose12;
    const float* pts_f; const uint8_t* pts_flags; const uint8_t* pts_desc; const int* pts_count;
    const uint8_t* cur_owner_obs;
    const float* cur_uright; float bf;   // stereo / RGB-D frames (mvuRight, mbf): the right-coordinate gate of ORBmatcher.cc:91-97; NULL = monocular
    float* frustum_xr;                    // optional [batch][pcap]: mTrackProjXR = u - mbf * invz (Frame.cc:499)
    int* match; int* nmatches; int* status; float* frustum;
    uint32_t* cand; int* cand_n;
    int cap, pcap, slot_n;    // slot_n: candidates per point cached in LDS (LOCAL_SLOT or 0)
    unsigned char* work; size_t work_bytes;   // k_search_local_points<true>: the per-stream work arrays in global memory
    float minX, maxX, minY, maxY, wInv, hInv, fx, fy, cx, cy, th, nnratio, log_sf;
    float scale[16];
    int nlevels;
};
__host__ __device__ inline size_t local_search_lds_bytes(int cap, int pcap, int slot_n = LOCAL_SLOT) {
    return (size_t)pcap * (slot_n * 4 + 4) + (size_t)cap * (4 + 4 + 8 + 4) + (size_t)(GRID_CELLS + 2) * 2 + 64;
}

template <bool GW>
__global__ __launch_bounds__(SEARCH_THREADS) void k_search_local_points(LocalSearchArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_lds[];
    unsigned char* s_raw = GW ? A.work + (size_t)blockIdx.x * A.work_bytes : s_lds;
    const int b = blockIdx.x, cap = A.cap, pcap = A.pcap, t = threadIdx.x, lane = t & 63;
    const int ncur = min(A.cur_count[b], cap), npts = min(A.pts_count[b], pcap);
    const int slot_n = A.slot_n;
    SPT_DECL;
    float2* pxy = reinterpret_cast<float2*>(s_raw);                              // [cap] position of the keypoint at CSR entry p (8-byte entries first: pcap may be odd)
    uint32_t* slot = reinterpret_cast<uint32_t*>(pxy + cap);                     // [slot_n][pcap] dist << 20 | level << 16 | idx: entry k of point i at k * pcap + i
    int* choice = reinterpret_cast<int*>(slot + (size_t)pcap * slot_n);         // [pcap] (touched by the thread of point i only, until the output)
    int* taken = choice + pcap;                                                  // [cap]
    int* taken2 = taken + cap;                                                   // [cap] the second buffer of `taken` during the sweeps
    uint32_t* pio = reinterpret_cast<uint32_t*>(taken2 + cap);                   // [cap] keypoint index | octave << 16 | (holds a point with observations) << 24 of CSR entry p
    uint16_t* cs = reinterpret_cast<uint16_t*>(pio + cap);                       // [GRID_CELLS + 2]
    __shared__ int s_changed[2], s_nm;
    const float* P = A.pose12 + (size_t)b * 12;
    const viorb_keypoint* ck = A.cur_kps + (size_t)b * cap;
    const uint8_t* pf = A.pts_flags + (size_t)b * pcap;
    uint32_t* cand = A.cand + (size_t)b * pcap * CAND_CAP;
    int* cand_n = A.cand_n + (size_t)b * pcap;
    // mOw = -Rcw^T tcw
    float Ow[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { const float tt = P[r] * P[9] + P[3 + r] * P[10] + P[6 + r] * P[11]; Ow[r] = -tt; }
    const bool bFactor = A.th != 1.0f;
    const float* cur_ur = A.cur_uright ? A.cur_uright + (size_t)b * cap : nullptr;
    // Frame::isInFrustum of local point i and its search window (ORBmatcher.cc:60-75); fr (when not null) receives mbTrackInView,
    // mTrackProjX, mTrackProjY, mTrackViewCos, mnTrackScaleLevel. Needs no LDS: a thread's first point is done BEFORE the frame is
    // staged, its chain of dependent global loads beside the staging loads (see k_search_projection).
    struct Window { float u, v, xr, radius; int x0, x1, y0, y1, minL, maxL; bool ok; };
    auto frustum = [&](int i, float* fr) -> Window {
        Window W; W.ok = false;
        W.u = W.v = W.xr = W.radius = 0.0f; W.x0 = W.y0 = 0; W.x1 = W.y1 = -1; W.minL = 0; W.maxL = -1;
        const int fl = pf[i];
        float fr_in = 0, fr_u = 0, fr_v = 0, fr_cos = 0, fr_lvl = 0, fr_xr = 0;
        if ((fl & 1) && !(fl & 2)) {
            const float* X = A.pts_f + ((size_t)b * pcap + i) * 8;
            float pc[3];
#pragma unroll
            for (int r = 0; r < 3; r++) { const float tt = P[3 * r] * X[0] + P[3 * r + 1] * X[1] + P[3 * r + 2] * X[2]; pc[r] = tt + P[9 + r]; }
            bool ok = !(pc[2] < 0.0f);
            const float invz = 1.0f / pc[2];
            const float u = A.fx * pc[0] * invz + A.cx, v = A.fy * pc[1] * invz + A.cy;
            ok = ok && !(u < A.minX || u > A.maxX) && !(v < A.minY || v > A.maxY);
            const float maxD = 1.2f * X[7], minD = 0.8f * X[6];
            const float PO0 = X[0] - Ow[0], PO1 = X[1] - Ow[1], PO2 = X[2] - Ow[2];
            const float dist = (float)sqrt((double)PO0 * PO0 + (double)PO1 * PO1 + (double)PO2 * PO2);
            ok = ok && !(dist < minD || dist > maxD);
            const float viewCos = (float)(((double)PO0 * X[3] + (double)PO1 * X[4] + (double)PO2 * X[5]) / (double)dist);
            ok = ok && !(viewCos < 0.5f);
            if (ok) {
                const float ratio = X[7] / dist;
                int lvl = (int)ceilf(viorb_logf(ratio) / A.log_sf);
                lvl = lvl < 0 ? 0 : (lvl >= A.nlevels ? A.nlevels - 1 : lvl);
                const float xr = u - A.bf * invz;               // mTrackProjXR (Frame.cc:499); bf = 0 for a monocular frame
                fr_in = 1; fr_u = u; fr_v = v; fr_cos = viewCos; fr_lvl = (float)lvl; fr_xr = xr;
                float rr = viewCos > 0.998 ? 2.5f : 4.0f;
                if (bFactor) rr *= A.th;
                const float radius = rr * A.scale[lvl];
                W.minL = lvl - 1; W.maxL = lvl;
                W.x0 = max(0, (int)floorf((u - A.minX - radius) * A.wInv));
                W.x1 = min((int)GRID_COLS - 1, (int)ceilf((u - A.minX + radius) * A.wInv));
                W.y0 = max(0, (int)floorf((v - A.minY - radius) * A.hInv));
                W.y1 = min((int)GRID_ROWS - 1, (int)ceilf((v - A.minY + radius) * A.hInv));
                W.u = u; W.v = v; W.xr = xr; W.radius = radius;
                W.ok = W.x0 < GRID_COLS && W.x1 >= 0 && W.y0 < GRID_ROWS && W.y1 >= 0;
            }
        }
        if (fr) { fr[0] = fr_in; fr[1] = fr_u; fr[2] = fr_v; fr[3] = fr_cos; fr[4] = fr_lvl; if (A.frustum_xr) A.frustum_xr[(size_t)b * pcap + i] = fr_xr; }
        return W;
    };
    float fr_tmp[5];
    auto fr_of = [&](int i) -> float* { return A.frustum ? A.frustum + ((size_t)b * pcap + i) * 5 : (A.frustum_xr ? fr_tmp : nullptr); };
    Window W0; W0.ok = false;
    if (t < npts) W0 = frustum(t, fr_of(t));
    {
        const int* gcs = A.cell_start + (size_t)b * (GRID_CELLS + 1);
        const int* gci = A.cell_idx + (size_t)b * cap;
        for (int i = t; i <= GRID_CELLS; i += blockDim.x) cs[i] = (uint16_t)gcs[i];
        const int ngrid = min(gcs[GRID_CELLS], ncur);                    // entries of the CSR (keypoints outside the grid have none)
        for (int p = t; p < ngrid; p += blockDim.x) {
            const int i2 = gci[p];
            const viorb_keypoint k = ck[i2];
            pxy[p] = make_float2(k.x, k.y);
            pio[p] = (uint32_t)i2 | ((uint32_t)(k.octave & 0xff) << 16) | (A.cur_owner_obs[(size_t)b * cap + i2] ? 1u << 24 : 0u);
        }
        for (int p = ngrid + t; p < cap; p += blockDim.x) { pxy[p] = make_float2(0.0f, 0.0f); pio[p] = 0; }   // read (and masked) by the clamped loads of the scan
    }
    if (t == 0) s_nm = 0;
    __syncthreads();
    SPT_LAP(0);
    // Candidates of the window in the reference's order, g(k, level << 16 | index): the single-loop scan of k_search_projection (four entries
    // of the current column and the bounds of the next per step, every LDS read issued up front, state moved by selects)
    auto scan = [&](const Window& W, auto&& g) -> int {
        int nc = 0;
        if (!W.ok) return 0;
        const float u = W.u, v = W.v, radius = W.radius;
        const int x1 = W.x1, y0 = W.y0, y1 = W.y1, minL = W.minL, maxL = W.maxL;
        const bool check_levels = (minL > 0) || (maxL >= 0);
        int ix = W.x0, p = 0, pend = 0;
        for (;;) {
            const int col = min(ix, x1) * GRID_ROWS;
            const int cbeg = cs[col + y0], cend = cs[col + y1 + 1];
            uint32_t e[SCAN_W]; float2 q[SCAN_W];
#pragma unroll
            for (int j = 0; j < SCAN_W; j++) { const int pp = min(p + j, cap - 1); e[j] = pio[pp]; q[j] = pxy[pp]; }
#pragma unroll
            for (int j = 0; j < SCAN_W; j++) {
                const int i2 = (int)(e[j] & 0xffff), o2 = (int)((e[j] >> 16) & 0xff);
                bool pass = p + j < pend && fabsf(q[j].x - u) < radius && fabsf(q[j].y - v) < radius;
                if (check_levels) pass = pass && !(o2 < minL) && !(maxL >= 0 && o2 > maxL);
                pass = pass && !(e[j] >> 24);                              // held by a point with observations: never available
                if (cur_ur && pass) {                  // "if(F.mvuRight[idx]>0) { er = fabs(mTrackProjXR - mvuRight[idx]); if(er > r*sf) continue; }"
                    const float ur = cur_ur[i2];
                    if (ur > 0 && fabsf(W.xr - ur) > radius) pass = false;
                }
                if (pass) { g(nc, e[j] & 0xfffffu); nc++; }
            }
            const bool adv = p + SCAN_W >= pend;                                // the column is done: enter the next, or stop after the last
            if (adv && ix > x1) break;
            p = adv ? cbeg : p + SCAN_W;
            pend = adv ? cend : pend;
            ix += adv ? 1 : 0;
        }
        return nc;
    };
    // the same with each candidate's Hamming distance, f(k, dist << 20 | level << 16 | index): only for a point with more than CAND_CAP
    // candidates, which every sweep of phase B enumerates again instead of cutting the list short — no capacity limit
    auto enumerate = [&](int i, auto&& f) -> int {
        const uint4* dl = reinterpret_cast<const uint4*>(A.pts_desc + ((size_t)b * pcap + i) * 32);
        const uint4 da = dl[0], db = dl[1];
        return scan(frustum(i, nullptr), [&](int k, uint32_t li) {
            const uint4* dc = reinterpret_cast<const uint4*>(A.cur_desc + ((size_t)b * cap + (li & 0xffff)) * 32);
            const uint4 ea = dc[0], eb = dc[1];
            const int dist2 = __popc(da.x ^ ea.x) + __popc(da.y ^ ea.y) + __popc(da.z ^ ea.z) + __popc(da.w ^ ea.w) +
                              __popc(db.x ^ eb.x) + __popc(db.y ^ eb.y) + __popc(db.z ^ eb.z) + __popc(db.w ^ eb.w);
            f(k, ((uint32_t)dist2 << 20) | li);
        });
    };
    auto set_entry = [&](int i, int k, uint32_t e) { if (k < slot_n) slot[(size_t)k * pcap + i] = e; else cand[(size_t)k * pcap + i] = e; };
    // four stored candidates from k0 (a multiple of 4, like slot_n: one address space per call), indices clamped to n - 1
    auto entries4 = [&](int i, int k0, int n, uint32_t* e) {
        if (k0 < slot_n) {
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = slot[(size_t)min(k0 + u, n - 1) * pcap + i];
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = cand[(size_t)min(k0 + u, n - 1) * pcap + i];
        }
    };
    // ---- phase A: frustum + candidate indices, then the Hamming distances eight candidates at a time
    for (int i = t; i < npts; i += blockDim.x) {
        const Window W = i == t ? W0 : frustum(i, fr_of(i));
        const int nc = scan(W, [&](int k, uint32_t li) { if (k < CAND_CAP) set_entry(i, k, li); });       // the true count, also beyond CAND_CAP
        cand_n[i] = nc;
        choice[i] = -1;
    }
#ifdef VIORB_SEARCH_TIMING
    __syncthreads();
    SPT_LAP(4);
#endif
    for (int i = t; i < npts; i += blockDim.x) {                     // (every thread reads back its own points' lists: no barrier)
        const int ns = min(cand_n[i], (int)CAND_CAP);
        if (ns > 0) {
            const uint4* dl = reinterpret_cast<const uint4*>(A.pts_desc + ((size_t)b * pcap + i) * 32);
            const uint4 da = dl[0], db = dl[1];
            for (int k0 = 0; k0 < ns; k0 += 8) {
                uint32_t id[8]; uint4 ea[8], eb[8];
                entries4(i, k0, ns, id);
                if (k0 + 4 < ns) entries4(i, k0 + 4, ns, id + 4);       // (never an index from a list position that was not written)
                else { id[4] = id[0]; id[5] = id[0]; id[6] = id[0]; id[7] = id[0]; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint4* dc = reinterpret_cast<const uint4*>(A.cur_desc + ((size_t)b * cap + (id[u] & 0xffff)) * 32);
                    ea[u] = dc[0]; eb[u] = dc[1];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (k0 + u >= ns) break;
                    const int dist2 = __popc(da.x ^ ea[u].x) + __popc(da.y ^ ea[u].y) + __popc(da.z ^ ea[u].z) + __popc(da.w ^ ea[u].w) +
                                      __popc(db.x ^ eb[u].x) + __popc(db.y ^ eb[u].y) + __popc(db.z ^ eb[u].z) + __popc(db.w ^ eb[u].w);
                    set_entry(i, k0 + u, ((uint32_t)dist2 << 20) | (id[u] & 0xfffffu));
                }
            }
        }
    }
    SPT_LAP(1);
    // ---- phase B: fixed-point sweeps over the greedy ownership; `taken` double-buffered, choice[i] owned by thread i: two barriers a sweep
    for (int c = t; c < ncur; c += blockDim.x) taken[c] = 0x7fffffff;
    if (t < 2) s_changed[t] = 0;
    __syncthreads();
    int cur = 0;
    for (int sweep = 0; sweep <= npts; sweep++, cur ^= 1) {
#ifdef VIORB_SEARCH_TIMING
        spt_sweeps++;
#endif
        int* tkc = cur ? taken2 : taken; int* tkn = cur ? taken : taken2;
        for (int i = t; i < npts; i += blockDim.x)
            if (choice[i] >= 0 && (pf[i] & 4)) atomicMin(&tkc[choice[i]], i);
        __syncthreads();
        bool changed = false;
        for (int i = t; i < npts; i += blockDim.x) {
            const int nc = cand_n[i];
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            auto rank = [&](uint32_t e, int ow) {
                const int i2 = (int)(e & 0xffff), dist = (int)(e >> 20), lv = (int)((e >> 16) & 0xf);
                if (ow < i) return;
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = lv; bestIdx = i2; }
                else if (dist < bestDist2) { bestLevel2 = lv; bestDist2 = dist; }
            };
            if (nc <= CAND_CAP) {
                for (int k0 = 0; k0 < nc; k0 += 4) {               // four entries, then their four owners, in flight together
                    uint32_t e[4]; int ow[4];
                    entries4(i, k0, nc, e);
#pragma unroll
                    for (int u = 0; u < 4; u++) ow[u] = tkc[e[u] & 0xffff];
#pragma unroll
                    for (int u = 0; u < 4; u++) if (k0 + u < nc) rank(e[u], ow[u]);
                }
            } else enumerate(i, [&](int, uint32_t e) { rank(e, tkc[e & 0xffff]); });      // more candidates than the stored list holds: walk the grid again
            int nw = -1;
            if (bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > A.nnratio * (float)bestDist2)) nw = bestIdx;
            changed = changed || (nw != choice[i]);
            choice[i] = nw;
        }
        for (int c = t; c < ncur; c += blockDim.x) tkn[c] = 0x7fffffff;
        if (t == 0) s_changed[cur ^ 1] = 0;
        if (__any(changed) && lane == 0) s_changed[cur] = 1;
        __syncthreads();
        if (!s_changed[cur]) break;
    }
    SPT_LAP(2);
    // ---- output: last assigner of every keypoint
    for (int c = t; c < ncur; c += blockDim.x) taken[c] = -1;
    __syncthreads();
    {
        int my_n = 0;
        for (int i = t; i < npts; i += blockDim.x) if (choice[i] >= 0) { atomicMax(&taken[choice[i]], i); my_n++; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) my_n += __shfl_xor(my_n, d);
        if (lane == 0) atomicAdd(&s_nm, my_n);
    }
    __syncthreads();
    int* out = A.match + (size_t)b * cap;
    for (int c = t; c < cap; c += blockDim.x) out[c] = c < ncur ? taken[c] : -1;
    if (t == 0) A.nmatches[b] = s_nm;
    SPT_LAP(3);
    SPT_PRINT("local points");
}

// ---------------------------------------------------------------------------------------------
// IMU: pre-integrate the samples between two frames, predict the NavState, derive the float pose.
// One workgroup (128 threads) per stream; the 9x9 covariance product is spread over 81 lanes, the
// 3x3 state is carried redundantly by every lane.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_imu_predict(const double* __restrict__ imu, int n_imu, const double* __restrict__ t_last,
                                                     const double* __restrict__ t_cur, const double* __restrict__ last_ns,
                                                     const double* __restrict__ gw3, const double* __restrict__ cam16,
                                                     double gyr_cov, double acc_cov, double* __restrict__ preint_out,
                                                     double* __restrict__ cur_ns, float* __restrict__ pose12) {
    __shared__ double s_cov[81], s_tmp[81], s_A[81], s_N[81];
    const int b = blockIdx.x, t = threadIdx.x;
    const double* ns = last_ns + (size_t)b * 22;
    const double* S = imu + (size_t)b * n_imu * 7;
    const d3 bg = ld3(ns + 10), ba = ld3(ns + 13);
    preint_small M;
    M.dP = mk3(0, 0, 0); M.dV = mk3(0, 0, 0); M.dR = eye3();
    M.JPg = zero3(); M.JPa = zero3(); M.JVg = zero3(); M.JVa = zero3(); M.JRg = zero3(); M.dt = 0;
    if (t < 81) s_cov[t] = 0;
    __syncthreads();
    const int r = t / 9, c = t % 9;
    for (int step = 0; step <= n_imu && n_imu > 0; step++) {
        // step 0: first sample over [t_last, t_imu0]; step k>=1: sample k-1 until the next stamp / the frame
        const int si = step == 0 ? 0 : step - 1;
        double d;
        if (step == 0) d = S[6] - t_last[b];
        else d = (si == n_imu - 1 ? t_cur[b] : S[(si + 1) * 7 + 6]) - S[si * 7 + 6];
        const d3 om = ld3(S + si * 7) - bg, ac = ld3(S + si * 7 + 3) - ba;
        const preint_cov_blocks C = preint_step(M, om, ac, d);
        if (t < 81) {
            // A = I + blocks; N = Bg*Sg*Bg^T + Ca*Sa*Ca^T
            double a = (r == c) ? 1.0 : 0.0;
            if (r >= 6 && c >= 6) a = m33_at(C.A66, r - 6, c - 6);
            else if (r >= 3 && r < 6 && c >= 6) a = m33_at(C.A36, r - 3, c - 6);
            else if (r < 3 && c >= 6) a = m33_at(C.A06, r, c - 6);
            else if (r < 3 && c >= 3 && c < 6) a = (r == c - 3) ? C.dt : 0.0;
            s_A[t] = a;
            double nn = 0;
            if (r >= 6 && c >= 6) { for (int k = 0; k < 3; k++) nn += m33_at(C.Bg, r - 6, k) * m33_at(C.Bg, c - 6, k); nn *= gyr_cov; }
            else if (r < 6 && c < 6) {
                const m33& Cr = r < 3 ? C.Ca0 : C.Ca3; const m33& Cc = c < 3 ? C.Ca0 : C.Ca3;
                for (int k = 0; k < 3; k++) nn += m33_at(Cr, r % 3, k) * m33_at(Cc, c % 3, k);
                nn *= acc_cov;
            }
            s_N[t] = nn;
        }
        __syncthreads();
        if (t < 81) { double s = 0; for (int k = 0; k < 9; k++) s += s_A[r * 9 + k] * s_cov[k * 9 + c]; s_tmp[t] = s; }
        __syncthreads();
        if (t < 81) { double s = 0; for (int k = 0; k < 9; k++) s += s_tmp[r * 9 + k] * s_A[c * 9 + k]; s_cov[t] = s + s_N[t]; }
        __syncthreads();
    }
    double* po = preint_out + (size_t)b * 142;
    if (t == 0) {
        st3(po, M.dP); st3(po + 3, M.dV); stm(po + 6, M.dR); stm(po + 15, M.JPg); stm(po + 24, M.JPa);
        stm(po + 33, M.JVg); stm(po + 42, M.JVa); stm(po + 51, M.JRg); po[141] = M.dt;
        // SetInitialNavStateAndBias(last) + UpdateNavState + UpdatePoseFromNS
        const pvr pred = update_ns(ld_pvr(ns), M.dP, M.dV, M.dR, M.dt, ld3(gw3));
        double* co = cur_ns + (size_t)b * 22;
        st_pvr(co, pred);
        // Frame::SetInitialNavStateAndBias (src/Frame.cc:117-125): bias <- bias + delta bias, delta <- 0
        for (int k = 10; k < 16; k++) co[k] = ns[k] + ns[k + 6];
        for (int k = 16; k < 22; k++) co[k] = 0.0;
        pose_from_navstate_f32(pred, cam16, pose12 + (size_t)b * 12);
    }
    if (t < 81) po[60 + t] = s_cov[t];
}

// Observations of the matched keypoints, compacted in keypoint order (the order of the reference's
// edge-construction loop): obs[k] = Pw(3) u v invSigma2, obs_index[k] = keypoint index.
__global__ __launch_bounds__(256) void k_build_observations(const viorb_keypoint* __restrict__ kps, const int* __restrict__ count,
                                                            const int* __restrict__ match, const float* __restrict__ match_Pw,
                                                            const float* __restrict__ inv_sigma2, int cap,
                                                            double* __restrict__ obs, int* __restrict__ obs_index, int* __restrict__ n_obs) {
    __shared__ int s_base;
    const int b = blockIdx.x;
    const int n = min(count[b], cap);
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int m = i < n ? match[(size_t)b * cap + i] : -1;
        const bool has = m >= 0;
        // block-wide ordered compaction: wave ballots + wave offsets through LDS
        __shared__ int s_wave[4];
        const unsigned long long bm = __ballot(has);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) s_wave[wv] = __popcll(bm);
        __syncthreads();
        int off = s_base;
        for (int k = 0; k < wv; k++) off += s_wave[k];
        if (has) {
            const int pos = off + __popcll(bm & ((1ull << lane) - 1ull));
            const viorb_keypoint kp = kps[(size_t)b * cap + i];
            const float* X = match_Pw + ((size_t)b * cap + m) * 3;
            double* o = obs + ((size_t)b * cap + pos) * 6;
            o[0] = X[0]; o[1] = X[1]; o[2] = X[2]; o[3] = kp.x; o[4] = kp.y; o[5] = inv_sigma2[kp.octave];
            obs_index[(size_t)b * cap + pos] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) n_obs[b] = s_base;
}

// TrackLocalMap's edge construction: as k_build_observations, with the frame's map points coming from two tables — the last
// frame's points (match_a, stride cap) for the keypoints matched by SearchByProjection(Frame, Frame) and the local map
// (match_b, stride stride_b) for those added by SearchLocalPoints. Keypoint order, as the loop over mvpMapPoints builds edges.
__global__ __launch_bounds__(256) void k_build_observations2(const viorb_keypoint* __restrict__ kps, const int* __restrict__ count,
                                                             const int* __restrict__ match_a, const float* __restrict__ Pw_a,
                                                             const int* __restrict__ match_b, const float* __restrict__ pts_b, int stride_b,
                                                             const float* __restrict__ inv_sigma2, int cap,
                                                             double* __restrict__ obs, int* __restrict__ obs_index, int* __restrict__ n_obs) {
    __shared__ int s_base, s_wave[4];
    const int b = blockIdx.x;
    const int n = min(count[b], cap);
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int ma = i < n ? match_a[(size_t)b * cap + i] : -1;
        const int mb = (i < n && ma < 0) ? match_b[(size_t)b * cap + i] : -1;
        const bool has = ma >= 0 || mb >= 0;
        const unsigned long long bm = __ballot(has);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) s_wave[wv] = __popcll(bm);
        __syncthreads();
        int off = s_base;
        for (int k = 0; k < wv; k++) off += s_wave[k];
        if (has) {
            const int pos = off + __popcll(bm & ((1ull << lane) - 1ull));
            const viorb_keypoint kp = kps[(size_t)b * cap + i];
            const float* X = ma >= 0 ? Pw_a + ((size_t)b * cap + ma) * 3 : pts_b + ((size_t)b * stride_b + mb) * 8;   // local points: pts_f[8], Pw first
            double* o = obs + ((size_t)b * cap + pos) * 6;
            o[0] = X[0]; o[1] = X[1]; o[2] = X[2]; o[3] = kp.x; o[4] = kp.y; o[5] = inv_sigma2[kp.octave];
            obs_index[(size_t)b * cap + pos] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) n_obs[b] = s_base;
}

// Tracking::TrackWithIMU's "Discard outliers" loop (reference src/Tracking.cc:489-507): matches whose edge ended as an outlier
// lose their map point; owner_obs[c] = the keypoint still holds a map point with observations (what SearchLocalPoints'
// SearchByProjection must not overwrite); n_map = nmatchesMap.
__global__ __launch_bounds__(256) void k_discard_outliers(int* __restrict__ match, const int* __restrict__ obs_index,
                                                          const uint8_t* __restrict__ outlier, const int* __restrict__ n_obs,
                                                          const uint8_t* __restrict__ pt_flags, int cap, uint8_t* __restrict__ owner_obs,
                                                          int* __restrict__ n_map) {
    __shared__ int s_n;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int n = min(n_obs[b], cap);
    for (int k = threadIdx.x; k < n; k += blockDim.x)
        if (outlier[(size_t)b * cap + k]) match[(size_t)b * cap + obs_index[(size_t)b * cap + k]] = -1;
    __syncthreads();
    int mine = 0;
    for (int c = threadIdx.x; c < cap; c += blockDim.x) {
        const int m = match[(size_t)b * cap + c];
        const uint8_t own = (m >= 0 && (pt_flags[(size_t)b * cap + m] & 4)) ? 1 : 0;
        owner_obs[(size_t)b * cap + c] = own;
        mine += own;
    }
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) n_map[b] = s_n;
}

// Frame::UpdatePoseFromNS (reference src/Frame.cc:88-105) for a batch of NavStates.
__global__ void k_pose_from_navstate(const double* __restrict__ ns, const double* __restrict__ cam16, int batch, float* __restrict__ pose12) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    pose_from_navstate_f32(ld_pvr(ns + (size_t)b * 22), cam16, pose12 + (size_t)b * 12);
}

// Workload support (not a reference function): the MapPoint fields Frame::isInFrustum reads, for points created from one frame
// of the synthetic world — MapPoint::UpdateNormalAndDepth with a single observation (reference src/MapPoint.cc:350-392):
// normal = (Pw - Ow) / |Pw - Ow|, mfMaxDistance = |Pw - Ow| * scaleFactor[octave], mfMinDistance = mfMaxDistance /
// scaleFactor[nlevels - 1]; evaluated in double from the float point and the double pose, rounded once.
__global__ void k_synth_local_points(const viorb_keypoint* __restrict__ kps, const int* __restrict__ count, int cap,
                                     const double* __restrict__ pose12, const float* __restrict__ Pw, const float* __restrict__ scale, int nlevels,
                                     float* __restrict__ pts_f) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const size_t o = (size_t)b * cap + i;
    float* out = pts_f + o * 8;
    if (i >= count[b]) { for (int k = 0; k < 8; k++) out[k] = 0.f; return; }
    const double* T = pose12 + (size_t)b * 12;
    const double ox = -(T[0] * T[9] + T[3] * T[10] + T[6] * T[11]), oy = -(T[1] * T[9] + T[4] * T[10] + T[7] * T[11]),
                 oz = -(T[2] * T[9] + T[5] * T[10] + T[8] * T[11]);
    const double px = Pw[3 * o], py = Pw[3 * o + 1], pz = Pw[3 * o + 2];
    const double dx = px - ox, dy = py - oy, dz = pz - oz, dist = sqrt(dx * dx + dy * dy + dz * dz);
    const double maxd = dist * (double)scale[kps[o].octave], mind = maxd / (double)scale[nlevels - 1];
    out[0] = Pw[3 * o]; out[1] = Pw[3 * o + 1]; out[2] = Pw[3 * o + 2];
    out[3] = (float)(dx / dist); out[4] = (float)(dy / dist); out[5] = (float)(dz / dist);
    out[6] = (float)mind; out[7] = (float)maxd;
}

// ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>&, th) (reference src/ORBmatcher.cc:825-975): one thread per map point —
// projection, image / distance / viewing-angle gates, MapPoint::PredictScale, then the best-Hamming key-frame feature of levels
// [l-1, l] inside the window that passes the chi-square reprojection gate (5.99 mono, 7.8 stereo). best_idx[p] = that feature when
// its distance is <= TH_LOW, else -1. The Replace / AddObservation bookkeeping is the caller's (map management). Candidates are
// visited in KeyFrame::GetFeaturesInArea order (src/KeyFrame.cc:906-945), so the first minimum wins as in the reference.
struct FuseArgs {
    const viorb_keypoint* kps; const uint8_t* desc; const float* uright; const int* count; const int* cell_start; const int* cell_idx;
    const float* pose12; const float* pts_f; const uint8_t* pts_valid; const uint8_t* pts_desc; const int* pts_count;
    int* best_idx; int* nfused;
    int cap, pcap, nlevels;
    float minX, maxX, minY, maxY, wInv, hInv, fx, fy, cx, cy, bf, th, log_sf;
    float scale[16], inv_sigma2[16];
};
__global__ __launch_bounds__(1024) void k_fuse(FuseArgs A) {
    __shared__ int s_n;
    const int b = blockIdx.x, t = threadIdx.x, cap = A.cap, pcap = A.pcap;
    const int npts = min(A.pts_count[b], pcap);
    const float* P = A.pose12 + (size_t)b * 12;
    const viorb_keypoint* kp = A.kps + (size_t)b * cap;
    const int* cs = A.cell_start + (size_t)b * (GRID_CELLS + 1);
    const int* ci = A.cell_idx + (size_t)b * cap;
    if (t == 0) s_n = 0;
    __syncthreads();
    float Ow[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { const float tt = P[r] * P[9] + P[3 + r] * P[10] + P[6 + r] * P[11]; Ow[r] = -tt; }
    int mine = 0;
    for (int i = t; i < pcap; i += blockDim.x) {
        int best = -1;
        if (i < npts && A.pts_valid[(size_t)b * pcap + i]) {
            const float* X = A.pts_f + ((size_t)b * pcap + i) * 8;
            float pc[3];
#pragma unroll
            for (int r = 0; r < 3; r++) { const float tt = P[3 * r] * X[0] + P[3 * r + 1] * X[1] + P[3 * r + 2] * X[2]; pc[r] = tt + P[9 + r]; }
            bool ok = !(pc[2] < 0.0f);
            const float invz = 1.0f / pc[2];
            const float x = pc[0] * invz, y = pc[1] * invz;
            const float u = A.fx * x + A.cx, v = A.fy * y + A.cy;
            ok = ok && (u >= A.minX && u < A.maxX && v >= A.minY && v < A.maxY);
            const float ur = u - A.bf * invz;
            const float maxD = 1.2f * X[7], minD = 0.8f * X[6];
            const float PO0 = X[0] - Ow[0], PO1 = X[1] - Ow[1], PO2 = X[2] - Ow[2];
            const float dist3D = (float)sqrt((double)PO0 * PO0 + (double)PO1 * PO1 + (double)PO2 * PO2);
            ok = ok && !(dist3D < minD || dist3D > maxD);
            const double dotp = (double)PO0 * X[3] + (double)PO1 * X[4] + (double)PO2 * X[5];
            ok = ok && !(dotp < 0.5 * (double)dist3D);
            if (ok) {
                const float ratio = X[7] / dist3D;
                int lvl = (int)ceilf(viorb_logf(ratio) / A.log_sf);
                lvl = lvl < 0 ? 0 : (lvl >= A.nlevels ? A.nlevels - 1 : lvl);
                const float radius = A.th * A.scale[lvl];
                const int x0 = max(0, (int)floorf((u - A.minX - radius) * A.wInv));
                const int x1 = min((int)GRID_COLS - 1, (int)ceilf((u - A.minX + radius) * A.wInv));
                const int y0 = max(0, (int)floorf((v - A.minY - radius) * A.hInv));
                const int y1 = min((int)GRID_ROWS - 1, (int)ceilf((v - A.minY + radius) * A.hInv));
                if (x0 < GRID_COLS && x1 >= 0 && y0 < GRID_ROWS && y1 >= 0) {
                    const uint4* dl = reinterpret_cast<const uint4*>(A.pts_desc + ((size_t)b * pcap + i) * 32);
                    const uint4 da = dl[0], db = dl[1];
                    int bestDist = 256;
                    for (int ix = x0; ix <= x1; ix++) {
                        const int pbeg = cs[ix * GRID_ROWS + y0], pend = cs[ix * GRID_ROWS + y1 + 1];
                        for (int q = pbeg; q < pend; q++) {
                            const int idx = ci[q];
                            const viorb_keypoint k = kp[idx];
                            if (!(fabsf(k.x - u) < radius && fabsf(k.y - v) < radius)) continue;
                            const int kl = k.octave;
                            if (kl < lvl - 1 || kl > lvl) continue;
                            const float kr = A.uright[(size_t)b * cap + idx];
                            const float ex = u - k.x, ey = v - k.y;
                            if (kr >= 0) {
                                const float er = ur - kr;
                                const float e2 = ex * ex + ey * ey + er * er;
                                if ((double)(e2 * A.inv_sigma2[kl]) > 7.8) continue;
                            } else {
                                const float e2 = ex * ex + ey * ey;
                                if ((double)(e2 * A.inv_sigma2[kl]) > 5.99) continue;
                            }
                            const uint4* dc = reinterpret_cast<const uint4*>(A.desc + ((size_t)b * cap + idx) * 32);
                            const uint4 ea = dc[0], eb = dc[1];
                            const int dist = __popc(da.x ^ ea.x) + __popc(da.y ^ ea.y) + __popc(da.z ^ ea.z) + __popc(da.w ^ ea.w) +
                                             __popc(db.x ^ eb.x) + __popc(db.y ^ eb.y) + __popc(db.z ^ eb.z) + __popc(db.w ^ eb.w);
                            if (dist < bestDist) { bestDist = dist; best = idx; }
                        }
                    }
                    if (bestDist > 50) best = -1;                       // TH_LOW
                }
            }
        }
        A.best_idx[(size_t)b * pcap + i] = best;
        mine += best >= 0;
    }
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (t == 0) A.nfused[b] = s_n;
}

// "mLastFrame = Frame(mCurrentFrame)" (reference src/Tracking.cc, end of Track()) for the batched harness, in ONE launch: the
// outgoing last frame's map points enter the local map (newest first, older slots shift back), the current frame's keypoints and
// descriptors become the last frame's, and the per-stream scalars (NavState, prior, time stamp, marginal) are carried over.
struct RollArgs {
    const viorb_keypoint* cur_kps; const uint8_t* cur_desc; const int* cur_count;
    viorb_keypoint* last_kps; uint8_t* last_desc; int* last_count;
    const float* last_pts_f; const uint8_t* last_flags;                  // the OUTGOING last frame's points (read before being replaced)
    float* loc_pts_f; uint8_t* loc_desc; uint8_t* loc_flags; int local_frames;      // [B][local_frames][cap] tables, or null
    const double* ns_src; double* last_ns; double* prior_ns; const double* t_src; double* t_last;
    const double* marg_src; double* marg_dst;                            // [B][144], or null
    int cap, shift_local;
};
__global__ __launch_bounds__(256) void k_roll(RollArgs A) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, cap = A.cap;
    if (i < cap) {
        const size_t o = (size_t)b * cap + i;
        if (A.loc_pts_f && A.shift_local) {
            const int R = A.local_frames;
            const size_t base = (size_t)b * R * cap + i;
            for (int r = R - 1; r >= 1; r--) {
                const size_t d = base + (size_t)r * cap, sidx = d - cap;
                for (int k = 0; k < 8; k++) A.loc_pts_f[d * 8 + k] = A.loc_pts_f[sidx * 8 + k];
                const uint4* sd = reinterpret_cast<const uint4*>(A.loc_desc + sidx * 32); uint4* dd = reinterpret_cast<uint4*>(A.loc_desc + d * 32);
                dd[0] = sd[0]; dd[1] = sd[1];
                A.loc_flags[d] = A.loc_flags[sidx];
            }
            for (int k = 0; k < 8; k++) A.loc_pts_f[base * 8 + k] = A.last_pts_f[o * 8 + k];
            const uint4* sd = reinterpret_cast<const uint4*>(A.last_desc + o * 32); uint4* dd = reinterpret_cast<uint4*>(A.loc_desc + base * 32);
            dd[0] = sd[0]; dd[1] = sd[1];
            A.loc_flags[base] = A.last_flags[o];
        }
        A.last_kps[o] = A.cur_kps[o];
        const uint4* sd = reinterpret_cast<const uint4*>(A.cur_desc + o * 32); uint4* dd = reinterpret_cast<uint4*>(A.last_desc + o * 32);
        dd[0] = sd[0]; dd[1] = sd[1];
    }
    if (blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t < 22) { const double v = A.ns_src[(size_t)b * 22 + t]; A.last_ns[(size_t)b * 22 + t] = v; A.prior_ns[(size_t)b * 22 + t] = v; }
        if (t == 32) { A.last_count[b] = A.cur_count[b]; A.t_last[b] = A.t_src[b]; }
        if (A.marg_src && t >= 64 && t < 64 + 144) A.marg_dst[(size_t)b * 144 + t - 64] = A.marg_src[(size_t)b * 144 + t - 64];
    }
}

// Workload support (not a reference function): map points for the keypoints of a frame of the
// synthetic plane world of viorb_amd/synth.py — intersects the pixel ray with the plane z = z0 using
// the given camera pose (Rcw, tcw) in double, writes float world points and flags = 1|4.
__global__ void k_self_index(const uint8_t* __restrict__ flags, const int* __restrict__ count, int cap, int* __restrict__ self_index) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    self_index[(size_t)b * cap + i] = (i < count[b] && (flags[(size_t)b * cap + i] & 1)) ? i : -1;
}

__global__ void k_synth_plane_points(const viorb_keypoint* __restrict__ kps, const int* __restrict__ count, int cap,
                                     const double* __restrict__ pose12, double fx, double fy, double cx, double cy, double z0,
                                     float* __restrict__ Pw, uint8_t* __restrict__ flags, int* __restrict__ self_index) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const size_t o = (size_t)b * cap + i;
    if (self_index) self_index[o] = i < count[b] ? i : -1;        // "keypoint i holds map point i": the match array of the frame's own points
    if (i >= count[b]) { flags[o] = 0; return; }
    const double* T = pose12 + (size_t)b * 12;
    const double dx = (kps[o].x - cx) / fx, dy = (kps[o].y - cy) / fy;
    // ray direction Rcw^T d and camera centre -Rcw^T t
    const double rx = T[0] * dx + T[3] * dy + T[6], ry = T[1] * dx + T[4] * dy + T[7], rz = T[2] * dx + T[5] * dy + T[8];
    const double ox = -(T[0] * T[9] + T[3] * T[10] + T[6] * T[11]), oy = -(T[1] * T[9] + T[4] * T[10] + T[7] * T[11]),
                 oz = -(T[2] * T[9] + T[5] * T[10] + T[8] * T[11]);
    const double s = (z0 - oz) / rz;
    Pw[3 * o] = (float)(ox + s * rx); Pw[3 * o + 1] = (float)(oy + s * ry); Pw[3 * o + 2] = (float)(oz + s * rz);
    flags[o] = 1 | 4;
}

// ---------------------------------------------------------------------------------------------
// Pose optimisation with NavState edges. One workgroup (256 threads) per problem.
// Unknowns: x = [cur PVR(9) | cur bias(3) | last PVR(9) | last bias(3)] (n = 24) for the Frame/Frame
// overload, x = [cur PVR(9) | cur bias(3)] (n = 12) for the Frame/KeyFrame overload (KF fixed).
// ---------------------------------------------------------------------------------------------
struct PoseOptArgs {
    int variant;                 // 0: last is a fixed KeyFrame; 1: last is a free Frame (+ prior edge)
    int compute_marg, cap;
    const double *cur_ns, *last_ns, *prior_ns, *marg_cov_inv, *preint, *gw, *cam;
    const double *obs_cur, *obs_last; const int *n_cur, *n_last;
    double *out_ns, *out_last_ns, *marg_out, *info;
    uint8_t *outlier_cur, *outlier_last;
    float* chi_store;            // [batch][2][cap] scratch of the VI solver: every mono edge's chi2 at the last evaluation (g2o's stored edge error)
    double acc_bias_rw2;
    const uint8_t* variant_arr;  // optional per-problem variant (overrides `variant`)
    const uint8_t* skip;         // optional: problems with skip[b] != 0 return at once like "fewer than 3 correspondences"
};


// hipcc's scheduler keeps register pressure low by pairing every LDS read with its use: a dot product fed from LDS becomes
// read -> s_waitcnt lgkmcnt(0) -> fma, once per term (a 27-term H entry took 2.6 k cycles). Where a phase is a handful of short chains
// fed by many LDS operands the operands are read into registers first and LDS_READS_DONE() keeps the reads above the arithmetic, so
// that they are all in flight together and the chain pays one LDS latency.
#define LDS_READS_DONE() __builtin_amdgcn_sched_barrier(0)
template <int N> __device__ __forceinline__ void lds_get(const double* p, double (&r)[N], int stride = 1) {
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = p[i * stride];
}

// single-wavefront LDS hand-offs: the LDS unit executes one wave's instructions in order, the fence only
// stops the compiler from moving LDS accesses across the hand-off
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// In-LDS right-looking Cholesky by ONE wavefront: eliminates the first `nsteps` columns of the n x n SPD
// matrix M (row-major, lower triangle used and overwritten by L); the trailing (n-nsteps)^2 block is then
// the Schur complement. Returns false (wave-uniform) on a non-positive pivot.
__device__ bool wave_cholesky(double* M, int n, int nsteps, int lane) {
    bool ok = true;
    for (int j = 0; j < nsteps; j++) {
        const double d = M[j * n + j];
        if (!(d > 0) || !isfinite(d)) { ok = false; break; }
        const double sd = sqrt(d);
        const int i = j + lane;
        if (i < n) M[i * n + j] = (i == j) ? sd : M[i * n + j] / sd;
        WAVE_LDS_SYNC();
        const int m = n - j - 1;
        for (int idx = lane; idx < m * m; idx += 64) {
            const int r = j + 1 + idx / m, c = j + 1 + idx % m;
            if (c <= r) M[r * n + c] -= M[r * n + j] * M[c * n + j];
        }
        WAVE_LDS_SYNC();
    }
    return ok;
}

__device__ __forceinline__ double readlane_d(double v, int l) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], l); u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return u.d;
}
// 1/sqrt(d) to full double precision: hardware estimate + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);          // y (1 + (1 - d y^2) / 2): three dependent operations per step instead of five
    y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);
    return y;
}
// Cholesky solve of (H + lambda I) x = b for ONE wavefront with lane i holding row i of the matrix in
// registers (N <= 24, fully unrolled: pivots and column entries travel by v_readlane, no LDS in the
// factorisation; L is transposed once through Lt for the back substitution). Entries above the diagonal
// hold don't-care values that never reach a valid result. Returns false on a non-positive pivot.
template <int N>
__device__ bool wave_solve_reg(const double* __restrict__ H, const double* __restrict__ bvec, double lambda,
                               double* __restrict__ Lt, double* __restrict__ x, int lane) {
    const int li = lane < N ? lane : N - 1;
    double a[N], invd[N];
#pragma unroll
    for (int c = 0; c < N; c++) a[c] = H[li * N + c];
    double rhs = bvec[li];
    LDS_READS_DONE();
#pragma unroll
    for (int c = 0; c < N; c++) a[c] += ((c == li) ? lambda : 0.0);
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; j++) {
        // Row j of the (symmetric, fully updated) trailing matrix sits in lane j's registers: A[j][k] = A[k][j] is the multiplier column
        // k needs, and reading it from lane j does not wait for this step's pivot arithmetic (reading l_kj from lane k would).
        const double d = readlane_d(a[j], j);
        double row[N];
#pragma unroll
        for (int k = j + 1; k < N; k++) row[k] = readlane_d(a[k], j);
        ok = ok && (d > 0.0) && isfinite(d);
        const double inv = rsqrt_nr(d);
        invd[j] = inv;
        const double lij = a[j] * inv;                           // lane j: d / sqrt(d)
        a[j] = lij;
        const double yj = readlane_d(rhs, j) * inv;              // forward substitution rides along
        if (lane == j) rhs = yj; else if (lane > j) rhs = fma(-lij, yj, rhs);
        const double m = lij * inv;                              // l_ij l_kj = (a_ij / d) a_jk
#pragma unroll
        for (int k = j + 1; k < N; k++) a[k] = fma(-m, row[k], a[k]);
    }
#pragma unroll
    for (int c = 0; c < N; c++) Lt[li * N + c] = a[c];
    WAVE_LDS_SYNC();
    double col[N];
#pragma unroll
    for (int j = 0; j < N; j++) col[j] = Lt[j * N + li];         // L[j][i]: column i of L
    LDS_READS_DONE();
    double acc = rhs;
#pragma unroll
    for (int j = N - 1; j >= 0; j--) {
        const double xj = readlane_d(acc, j) * invd[j];
        if (lane == j) acc = xj; else if (lane < j) acc = fma(-col[j], xj, acc);
    }
    if (lane < N) x[lane] = acc;
    return ok;
}

// Quaternion normalisation by ONE reciprocal square root (v_rsq_f64 + two Newton steps, ~1 ulp) and four multiplications instead of sqrt and
// four IEEE divisions (~150 dependent-ish f64 instructions): the scalar phases of the pose solver (IMU / prior factor pieces, state update) normalise
// five to eight quaternions per evaluation on a single lane, which was 40 % of their length. Used where the parity bar is a tolerance
// (solver state 1e-7, cost 1e-5), never in the bit-exact front-end arithmetic; DESIGN.md section 2, deviation 3.
__device__ __forceinline__ quat qnorm_f(quat q) {
    const double r = rsqrt_nr(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return mkq(q.x * r, q.y * r, q.z * r, q.w * r);
}
__device__ __forceinline__ quat so3_mul_f(quat a, quat b) { return qnorm_f(qmul(qnorm_f(a), b)); }
// sin and cos of the small angles of the solver's scalar phases (an LM step's rotation, a factor's rotation residual: |x| well below 0.5)
// by their Taylor series to x^17 / x^16 in Horner form — truncation below 3e-20, rounding ~1 ulp — instead of the library routines with
// their full argument reduction (~150 instructions each on one lane); any larger argument takes the library path.
__device__ __forceinline__ void sincos_small(double x, double* sn, double* cs) {
    if (fabs(x) < 0.5) {
        const double z = x * x;
        double ps = -1.0 / 355687428096000.0;                                            // -1/17!
        ps = fma(ps, z, 1.0 / 1307674368000.0); ps = fma(ps, z, -1.0 / 6227020800.0); ps = fma(ps, z, 1.0 / 39916800.0);
        ps = fma(ps, z, -1.0 / 362880.0); ps = fma(ps, z, 1.0 / 5040.0); ps = fma(ps, z, -1.0 / 120.0); ps = fma(ps, z, 1.0 / 6.0);
        *sn = fma(-(x * z), ps, x);
        double pc = 1.0 / 20922789888000.0;                                              // 1/16!
        pc = fma(pc, z, -1.0 / 87178291200.0); pc = fma(pc, z, 1.0 / 479001600.0); pc = fma(pc, z, -1.0 / 3628800.0);
        pc = fma(pc, z, 1.0 / 40320.0); pc = fma(pc, z, -1.0 / 720.0); pc = fma(pc, z, 1.0 / 24.0); pc = fma(pc, z, -0.5);
        *cs = fma(pc, z, 1.0);
    } else { *sn = sin(x); *cs = cos(x); }
}
__device__ __forceinline__ quat so3_exp_f(d3 w) {
    const double th = norm3(w), half = 0.5 * th;
    double sh, real; sincos_small(half, &sh, &real);
    double imag;
    if (th < 1e-10) { const double t2 = th * th, t4 = t2 * t2; imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4; }
    else imag = sh / th;
    return qnorm_f(mkq(imag * w.x, imag * w.y, imag * w.z, real));
}
// so3.cpp JacobianRInv with the small-angle sin / cos above
__device__ __forceinline__ m33 so3_jr_inv_f(d3 w) {
    const double th = norm3(w);
    if (th < 0.00001) return eye3();
    const m33 K = hat3(w * (1.0 / th));
    double sn, cs; sincos_small(th, &sn, &cs);
    return add(add(eye3(), scl(hat3(w), 0.5)), scl(mul(K, K), 1.0 - (1.0 + cs) * th / (2.0 * sn)));
}
__device__ __forceinline__ pvr sh_pvr(const double* p) { pvr s; s.P = ld3(p); s.V = ld3(p + 3); s.q = mkq(p[6], p[7], p[8], p[9]); return s; }
__device__ __forceinline__ void sh_put(double* p, const pvr& s) { st3(p, s.P); st3(p + 3, s.V); p[6] = s.q.x; p[7] = s.q.y; p[8] = s.q.z; p[9] = s.q.w; }


// The IMU factor of vio_core.h's pvr_edge() — same operations in the same order per value — cut into three independent pieces that
// the solver runs on the first lanes of three different waves (a single lane took ~15 k cycles per evaluation, as long as its wave's
// whole share of the reprojection edges). est_i / est_j: P V q (10 doubles); every operand is read from LDS where it is used.
// J (9 x 21, row-major) keeps its static zero pattern; each piece writes its own blocks.
__device__ __forceinline__ void imu_put(double* J, int r0, int c0, const m33& B, double s) {
    const double v[9] = {B.a00, B.a01, B.a02, B.a10, B.a11, B.a12, B.a20, B.a21, B.a22};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) J[(r0 + r) * 21 + c0 + c] = s * v[3 * r + c];
}
// Constant parts (they depend on the pre-integration and on the last frame's fixed gyro-bias delta only), computed once per solve:
// cpv = [dP + JPg*dbg | dV + JVg*dbg], corrT = (dRij * Exp(JRg*dbg))^-1 as pvr_edge() forms it.
__device__ __forceinline__ void imu_constants(const double* pre, const double* dbg_i_p, double* cpv, double* corrT) {
    const d3 dbg_i = ld3(dbg_i_p);
    st3(cpv, ld3(pre) + mulv(ldm(pre + 15), dbg_i));
    st3(cpv + 3, ld3(pre + 3) + mulv(ldm(pre + 33), dbg_i));
    const quat dRij = qnorm(mat2q(ldm(pre + 6)));
    const quat corr = so3_mul(dRij, so3_exp(mulv(ldm(pre + 51), dbg_i)));
    const quat c = qnorm(qconj(corr));
    corrT[0] = c.x; corrT[1] = c.y; corrT[2] = c.z; corrT[3] = c.w;
}
// piece 1: position and velocity residuals, their rotation-column blocks and the accelerometer-bias blocks
__device__ __forceinline__ void imu_piece_pv(const double* est_i, const double* est_j, const double* cpv, const double* dba_i_p, const double* pre,
                                             const double* gw_p, double* e, double* J) {
    double ei[10], ej[6], g3[3], db[3], ja[9], jv[9], cp[6];
    lds_get(est_i, ei); lds_get(est_j, ej); lds_get(gw_p, g3); lds_get(dba_i_p, db); lds_get(pre + 24, ja); lds_get(pre + 42, jv); lds_get(cpv, cp);
    const double dT = pre[60], dT2 = dT * dT;
    LDS_READS_DONE();
    const pvr si = sh_pvr(ei); const d3 Pj = ld3(ej), Vj = ld3(ej + 3), gw = ld3(g3), dba_i = ld3(db);
    const quat RiT = qnorm_f(qconj(si.q));
    const d3 aP = qrot(RiT, Pj - si.P - si.V * dT - gw * (0.5 * dT2));
    const d3 aV = qrot(RiT, Vj - si.V - gw * dT);
    const m33 JPa = ldm(ja), JVa = ldm(jv);
    const d3 rP = aP - (ld3(cp) + mulv(JPa, dba_i));
    const d3 rV = aV - (ld3(cp + 3) + mulv(JVa, dba_i));
    e[0] = rP.x; e[1] = rP.y; e[2] = rP.z; e[3] = rV.x; e[4] = rV.y; e[5] = rV.z;
    if (!J) return;
    imu_put(J, 0, 6, hat3(aP), 1); imu_put(J, 3, 6, hat3(aV), 1);
    imu_put(J, 0, 18, JPa, -1); imu_put(J, 3, 18, JVa, -1);
}
// piece 2: rotation residual and the two blocks that carry Jr^-1
__device__ __forceinline__ void imu_piece_rot(const double* est_i, const double* est_j, const double* corrT, double* e, double* J) {
    double qa[4], qb[4], ct[4];
    lds_get(est_i + 6, qa); lds_get(est_j + 6, qb); lds_get(corrT, ct);
    LDS_READS_DONE();
    const quat qi = mkq(qa[0], qa[1], qa[2], qa[3]), qj = mkq(qb[0], qb[1], qb[2], qb[3]);
    const quat RiT = qnorm_f(qconj(qi));
    const quat rR = so3_mul_f(so3_mul_f(mkq(ct[0], ct[1], ct[2], ct[3]), RiT), qj);
    const d3 rPhi = so3_log(rR);
    e[6] = rPhi.x; e[7] = rPhi.y; e[8] = rPhi.z;
    if (!J) return;
    const m33 JrInv = so3_jr_inv_f(rPhi);
    imu_put(J, 6, 15, JrInv, 1);
    imu_put(J, 6, 6, mul(mul(JrInv, tr(qmat(qj))), qmat(qi)), -1);
}
// the prior factor (vio_core.h prior_edge()) with its constant parts hoisted: pri = [P | V | conj(q) normalised | bias_acc + dbias_acc]
// of the prior NavState; J (12 x 12) keeps its static pattern, including the four constant -1 diagonals written at setup.
__device__ __forceinline__ void prior_constants(const double* prior22, double* pri) {
    const pvr pr = ld_pvr(prior22);
    st3(pri, pr.P); st3(pri + 3, pr.V);
    const quat c = qnorm(qconj(pr.q));
    pri[6] = c.x; pri[7] = c.y; pri[8] = c.z; pri[9] = c.w;
    st3(pri + 10, ld3(prior22 + 13) + ld3(prior22 + 19));
}
__device__ __forceinline__ void prior_piece(const double* est, d3 ba_plus_dba, const double* pri, double* e, double* J) {
    double es[10], pr[13];
    lds_get(est, es); lds_get(pri, pr);
    LDS_READS_DONE();
    const pvr s = sh_pvr(es);
    const d3 eP = ld3(pr) - s.P, eV = ld3(pr + 3) - s.V;
    const d3 eR = so3_log(so3_mul_f(mkq(pr[6], pr[7], pr[8], pr[9]), s.q));
    const d3 eB = ld3(pr + 10) - ba_plus_dba;
    e[0] = eP.x; e[1] = eP.y; e[2] = eP.z; e[3] = eV.x; e[4] = eV.y; e[5] = eV.z; e[6] = eR.x; e[7] = eR.y; e[8] = eR.z; e[9] = eB.x; e[10] = eB.y; e[11] = eB.z;
    if (!J) return;
    const m33 R = qmat(s.q), Ji = so3_jr_inv_f(eR);
    const double rv[9] = {R.a00, R.a01, R.a02, R.a10, R.a11, R.a12, R.a20, R.a21, R.a22};
    const double jv[9] = {Ji.a00, Ji.a01, Ji.a02, Ji.a10, Ji.a11, Ji.a12, Ji.a20, Ji.a21, Ji.a22};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) { J[r * 12 + c] = -rv[3 * r + c]; J[(6 + r) * 12 + 6 + c] = jv[3 * r + c]; }
}
// piece 3: the blocks built from Ri^T alone
__device__ __forceinline__ void imu_piece_blocks(const double* est_i, const double* est_j, const double* pre, double* J) {
    double qa[4], qb[4];
    lds_get(est_i + 6, qa); lds_get(est_j + 6, qb);
    const double dT = pre[60];
    LDS_READS_DONE();
    const quat qi = mkq(qa[0], qa[1], qa[2], qa[3]), qj = mkq(qb[0], qb[1], qb[2], qb[3]);
    const m33 RiTm = tr(qmat(qi));
    imu_put(J, 0, 0, eye3(), -1); imu_put(J, 0, 3, RiTm, -dT); imu_put(J, 3, 3, RiTm, -1); imu_put(J, 3, 12, RiTm, 1);
    imu_put(J, 0, 9, mul(RiTm, qmat(qj)), 1);
}

#include "pose_opt_mp.inc"

// ---------------------------------------------------------------------------------------------
// Vision-only pose optimisation, Optimizer::PoseOptimization(Frame*) (reference src/Optimizer.cc:3749-3978):
// one 6-DoF SE3 vertex (left-multiplicative update), mono and stereo only-pose edges, the same 4 x optimize(10)
// outlier scheme and g2o LM as the VI solve. One workgroup per problem.
// ---------------------------------------------------------------------------------------------
struct Se3Args {
    const float* pose12; const double* obs7; const int* n_obs; int cap;
    double fx, fy, cx, cy, bf;
    float* out_pose12; uint8_t* outlier; double* info;
};
struct Se3Shared { double H[36], Lm[36], b[6], x[6]; double red[4][28]; double est[7], bak[7], ev[7]; double sc[8]; int flag[4]; };   // ev: the estimate of the last computeActiveErrors

__global__ __launch_bounds__(256) void k_pose_opt_se3(Se3Args A) {
    __shared__ Se3Shared S;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, cap = A.cap;
    const int n = min(A.n_obs[b], cap);
    const double* ob = A.obs7 + (size_t)b * cap * 7;
    uint8_t* ol = A.outlier + (size_t)b * cap;
    const float* p0 = A.pose12 + (size_t)b * 12;
    const double d_mono = (double)(float)sqrt(5.991), d_stereo = (double)(float)sqrt(7.815);
    for (int i = t; i < n; i += blockDim.x) ol[i] = 0;
    if (t == 0) S.flag[1] = 0;
    __syncthreads();
    if (n < 3) {
        if (t == 0) { for (int k = 0; k < 12; k++) A.out_pose12[(size_t)b * 12 + k] = p0[k]; double* inf = A.info + (size_t)b * 4; inf[0] = inf[1] = inf[2] = inf[3] = 0; }
        return;
    }
    auto ld_est = [&]() { se3q s; s.r = mkq(S.est[0], S.est[1], S.est[2], S.est[3]); s.t = mk3(S.est[4], S.est[5], S.est[6]); return s; };
    int kernel_on = 1, nbad = 0;
    auto evaluate = [&](bool lin) -> double {
        const se3q s = ld_est();
        double a[28];
#pragma unroll
        for (int k = 0; k < 28; k++) a[k] = 0;
        for (int i = t; i < n; i += blockDim.x) {
            if (ol[i]) continue;
            const double* o = ob + 7 * i;
            double e[3], J[18];
            const int dim = se3_edge(s, ld3(o), o[3], o[4], o[5], A.fx, A.fy, A.cx, A.cy, A.bf, lin, e, J);
            const double is2 = o[6];
            const double chi = is2 * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            double r0 = chi, r1 = 1;
            if (kernel_on) huber(chi, dim == 3 ? d_stereo : d_mono, &r0, &r1);
            a[27] += r0;
            if (lin) {
                const double w = r1 * is2;
                int k = 0;
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = r; c < 6; c++) a[k++] += w * (J[r] * J[c] + J[6 + r] * J[6 + c] + J[12 + r] * J[12 + c]);
#pragma unroll
                for (int r = 0; r < 6; r++) a[21 + r] -= w * (J[r] * e[0] + J[6 + r] * e[1] + J[12 + r] * e[2]);
            }
        }
        if (lin) {
            double v[32];
#pragma unroll
            for (int k = 0; k < 28; k++) v[k] = a[k];
#pragma unroll
            for (int k = 28; k < 32; k++) v[k] = 0;
#pragma unroll
            for (int half = 16, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
                const bool up = (lane & bit) != 0;
#pragma unroll
                for (int i = 0; i < half; i++) { const double keep = up ? v[half + i] : v[i]; const double send = up ? v[i] : v[half + i]; v[i] = keep + __shfl_xor(send, bit); }
            }
            const double tot = v[0] + __shfl_xor(v[0], 1);
            const int idx = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            if ((lane & 1) == 0 && idx < 28) S.red[wave][idx] = tot;
        } else {
            double vv = a[27];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) vv += __shfl_xor(vv, d);
            if (lane == 0) S.red[wave][27] = vv;
        }
        __syncthreads();
        if (lin && t < 27) {
            const double v = S.red[0][t] + S.red[1][t] + S.red[2][t] + S.red[3][t];
            if (t < 21) {
                int kk = 0, rr = 0, cc = 0;
                for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) { if (kk == t) { rr = r; cc = c; } kk++; }
                S.H[rr * 6 + cc] = v; S.H[cc * 6 + rr] = v;
            } else S.b[t - 21] = v;
        }
        if (t == 32) S.sc[0] = S.red[0][27] + S.red[1][27] + S.red[2][27] + S.red[3][27];
        __syncthreads();
        return S.sc[0];
    };
    for (int round = 0; round < 4; round++) {
        if (t == 0) {        // Converter::toSE3Quat(pFrame->mTcw)
            const m33 R = mkm(p0[0], p0[1], p0[2], p0[3], p0[4], p0[5], p0[6], p0[7], p0[8]);
            const quat q = se3_norm_rot(mat2q(R));
            S.est[0] = q.x; S.est[1] = q.y; S.est[2] = q.z; S.est[3] = q.w; S.est[4] = p0[9]; S.est[5] = p0[10]; S.est[6] = p0[11];
        }
        __syncthreads();
        double lambda = 0, ni = 2; int nBadLM = 0;
        bool last_rejected = false;
        for (int it = 0; it < 10; it++) {
            double currentChi = evaluate(true);
            const double iniChi = currentChi;
            if (it == 0) { double mx = 0; for (int i = 0; i < 6; i++) mx = fmax(fabs(S.H[i * 7]), mx); lambda = 1e-5 * mx; ni = 2; nBadLM = 0; }
            double rho = 0; int qmax = 0;
            do {
                if (t < 7) S.bak[t] = S.est[t];
                if (wave == 0) {
                    const bool ok = wave_solve_reg<6>(S.H, S.b, lambda, S.Lm, S.x, lane);
                    if (!ok && lane < 6) S.x[lane] = 0;
                    if (lane == 0) S.flag[0] = ok ? 1 : 0;
                }
                __syncthreads();
                const int ok2 = S.flag[0];
                if (t == 0) {
                    const se3q nw = se3_mul(se3_exp(S.x), ld_est());
                    S.est[0] = nw.r.x; S.est[1] = nw.r.y; S.est[2] = nw.r.z; S.est[3] = nw.r.w; S.est[4] = nw.t.x; S.est[5] = nw.t.y; S.est[6] = nw.t.z;
                }
                __syncthreads();
                if (t < 7) S.ev[t] = S.est[t];
                double tempChi = evaluate(false);
                if (!ok2) tempChi = 1.7976931348623157e308;
                double scale = 0; for (int j = 0; j < 6; j++) scale += S.x[j] * (lambda * S.x[j] + S.b[j]);
                scale += 1e-3;
                rho = (currentChi - tempChi) / scale;
                last_rejected = !(rho > 0 && isfinite(tempChi));
                if (!last_rejected) { double alpha = 1. - pow(2 * rho - 1, 3); alpha = fmin(alpha, 2. / 3.); lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi; }
                else { lambda *= ni; ni *= 2; if (t < 7) S.est[t] = S.bak[t]; }
                __syncthreads();
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (t == 0) { S.flag[1]++; S.sc[7] = currentChi; }
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) break;
        }
        __syncthreads();
        {
            // Optimizer.cc:3880-3940: an edge flagged as outlier is re-evaluated at the new estimate ("if(pFrame->mvbOutlier[idx]) e->computeError()"),
            // an inlier is classified by its STORED error — the error of the last computeActiveErrors, i.e. of the last trial state, which is
            // not the estimate when that trial was rejected (optimization_algorithm_levenberg.cpp:143-147 restores the vertices only)
            const se3q s_est = ld_est();
            se3q s_ev; s_ev.r = mkq(S.ev[0], S.ev[1], S.ev[2], S.ev[3]); s_ev.t = mk3(S.ev[4], S.ev[5], S.ev[6]);
            int bad_local = 0;
            for (int i = t; i < n; i += blockDim.x) {
                const double* o = ob + 7 * i;
                double e[3];
                const se3q s = (last_rejected && !ol[i]) ? s_ev : s_est;
                const int dim = se3_edge(s, ld3(o), o[3], o[4], o[5], A.fx, A.fy, A.cx, A.cy, A.bf, false, e, nullptr);
                const float chi2 = (float)(o[6] * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]));
                const int bad = chi2 > (dim == 3 ? 7.815f : 5.991f);
                ol[i] = (uint8_t)bad; bad_local += bad;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) bad_local += __shfl_xor(bad_local, d);
            if (t == 0) S.flag[2] = 0;
            __syncthreads();
            if (lane == 0) atomicAdd(&S.flag[2], bad_local);
            __syncthreads();
            nbad = S.flag[2];
        }
        if (round == 2) kernel_on = 0;
        __syncthreads();
        if (n < 10) break;
    }
    if (t == 0) {            // Converter::toCvMat(SE3Quat)
        const se3q s = ld_est();
        const m33 R = qmat(s.r);
        float* o = A.out_pose12 + (size_t)b * 12;
        o[0] = (float)R.a00; o[1] = (float)R.a01; o[2] = (float)R.a02; o[3] = (float)R.a10; o[4] = (float)R.a11; o[5] = (float)R.a12;
        o[6] = (float)R.a20; o[7] = (float)R.a21; o[8] = (float)R.a22; o[9] = (float)s.t.x; o[10] = (float)s.t.y; o[11] = (float)s.t.z;
        double* inf = A.info + (size_t)b * 4;
        inf[0] = n - nbad; inf[1] = S.sc[7]; inf[2] = S.flag[1]; inf[3] = 0;
    }
}

} // namespace viorb

// ---------------------------------------------------------------------------------------------
// Host side: handle, launches, C ABI
// ---------------------------------------------------------------------------------------------
using namespace viorb;

struct viorb_frontend {
    viorb_frontend_config cfg;
    int max_batch = 0, cap = 0, device = 0, sort_n = 0;
    float wInv = 0, hInv = 0;
    uint32_t* d_cand = nullptr; int* d_cand_n = nullptr;
    float* d_pose_chi = nullptr;                              // k_pose_opt_vi_mp's per-edge chi2 scratch [max_batch][2][cap]
    uint32_t* d_lcand = nullptr; int* d_lcand_n = nullptr; int lcand_pcap = 0;
    unsigned char* d_search_work = nullptr; size_t search_work_bytes = 0;      // work arrays of the searches when a frame's keypoints do not fit LDS
    double *d_cam = nullptr, *d_gw = nullptr; float* d_inv_sigma2 = nullptr; float* d_scale = nullptr;
};

// Launch of the visual-inertial pose solve. VIORB_POSE_MP="P,WPP" selects an instantiation of k_pose_opt_vi_mp (problems per workgroup,
// wavefronts per problem); the default depends on the batch size (below).
#ifndef POSE_MP_DEFAULT_P
#define POSE_MP_DEFAULT_P 2
#define POSE_MP_DEFAULT_WPP 2
#endif
template <int P, int WPP> static int launch_pose_mp(const PoseOptArgs& A, int batch, hipStream_t st) {
    // VIORB_POSE_LDS_MIN (bytes, experiment switch): a larger LDS request than the solver needs limits the workgroups resident per CU
    static const size_t lds_min = [] { const char* e = getenv("VIORB_POSE_LDS_MIN"); return e ? (size_t)atol(e) : (size_t)0; }();
    const size_t lds = std::max(sizeof(PoseMpShared<P, WPP>), lds_min);
    if (lds > 64 * 1024) VIORB_HIP_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(k_pose_opt_vi_mp<P, WPP>), lds));
    hipLaunchKernelGGL((k_pose_opt_vi_mp<P, WPP>), dim3((batch + P - 1) / P), dim3(64 * P * WPP), lds, st, A, batch);
    return VIORB_OK;
}
static std::atomic<int> g_pose_shape_override{0};          // viorb_frontend_set_pose_shape: P * 16 + WPP, 0 = automatic
static int launch_pose_opt_vi(const PoseOptArgs& A, int batch, hipStream_t st) {
    static int cfg_env = -1, n_cu = 0;
    if (cfg_env < 0) {
        cfg_env = 0;
        if (const char* e = getenv("VIORB_POSE_MP")) { int P = 0, W = 0; sscanf(e, "%d,%d", &P, &W); cfg_env = 0x100 | (P * 16 + W); }
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n_cu = pr.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    // Throughput shape <2, 2> (two problems share a 256-thread workgroup's scalar phases: fewest instructions and fewest resident wavefronts per
    // problem) once there is a problem pair per CU; below that the chip is idle anyway and a solve's LATENCY is what a step waits for: four
    // wavefronts per problem, eight when even a problem per two CUs is not there (a single stream).
    int cfg = cfg_env & 0x100 ? (cfg_env & 0xff) : (batch >= 2 * n_cu ? POSE_MP_DEFAULT_P * 16 + POSE_MP_DEFAULT_WPP : (batch > n_cu / 2 ? 1 * 16 + 4 : 1 * 16 + 8));
    if (const int ov = g_pose_shape_override.load()) cfg = ov;
    switch (cfg) {
        case 1 * 16 + 4: return launch_pose_mp<1, 4>(A, batch, st);
        case 1 * 16 + 8: return launch_pose_mp<1, 8>(A, batch, st);
        case 1 * 16 + 2: return launch_pose_mp<1, 2>(A, batch, st);
        case 2 * 16 + 2: return launch_pose_mp<2, 2>(A, batch, st);
        case 2 * 16 + 3: return launch_pose_mp<2, 3>(A, batch, st);
        case 1 * 16 + 3: return launch_pose_mp<1, 3>(A, batch, st);
        case 2 * 16 + 4: return launch_pose_mp<2, 4>(A, batch, st);
        case 4 * 16 + 1: return launch_pose_mp<4, 1>(A, batch, st);
        case 4 * 16 + 2: return launch_pose_mp<4, 2>(A, batch, st);
        default: set_error("VIORB_POSE_MP: unsupported (P, WPP)"); return VIORB_ERR_UNSUPPORTED;
    }
}

extern "C" {

int viorb_frontend_create(const viorb_frontend_config* cfg, int max_batch, int cap, int device, viorb_frontend** out) {
    VIORB_REQUIRE(cfg && out, "null cfg/out");
    VIORB_REQUIRE(max_batch >= 1 && cap >= 1 && cap <= 65535, "max_batch >= 1, 1 <= cap <= 65535 (16-bit keypoint indices)");
    VIORB_REQUIRE(cfg->nlevels >= 1 && cfg->nlevels <= 16, "nlevels must be 1..16");
    VIORB_REQUIRE(cfg->max_x > cfg->min_x && cfg->max_y > cfg->min_y, "empty image bounds");
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    VIORB_HIP_TRY(hipSetDevice(device));
    viorb_frontend* h = new viorb_frontend();
    h->cfg = *cfg; h->max_batch = max_batch; h->cap = cap; h->device = device;
    if (h->cfg.gyr_meas_cov <= 0) h->cfg.gyr_meas_cov = 2.0e-3 * 2.0e-3 * 200;     // reference src/IMU/imudata.cpp:36-37
    if (h->cfg.acc_meas_cov <= 0) h->cfg.acc_meas_cov = 8.0e-3 * 8.0e-3 * 200;
    if (h->cfg.acc_bias_rw2 <= 0) h->cfg.acc_bias_rw2 = 5e-3 * 5e-3;               // :32
    // Frame.cc:184-185
    h->wInv = static_cast<float>(GRID_COLS) / static_cast<float>(cfg->max_x - cfg->min_x);
    h->hInv = static_cast<float>(GRID_ROWS) / static_cast<float>(cfg->max_y - cfg->min_y);
    int s = 64; while (s < cap) s <<= 1;
    h->sort_n = s;
    {   // the projection search's LDS plan limits cap to ~4900 — checked where it is launched: the other calls of the handle (grid, IMU
        // prediction, pose solves: the host drop-in of PoseOptimization builds its handle for the number of edges) have no such limit
        const size_t lds = search_lds_bytes(cap, search_slot_n(cap));
        if (lds > 64 * 1024 && lds <= 160 * 1024) VIORB_HIP_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(k_search_projection<false>), lds));
    }
    VIORB_HIP_TRY(hipMalloc(&h->d_cand, (size_t)max_batch * cap * CAND_CAP * sizeof(uint32_t)));
    VIORB_HIP_TRY(hipMalloc(&h->d_pose_chi, (size_t)max_batch * 2 * cap * sizeof(float)));
    VIORB_HIP_TRY(hipMalloc(&h->d_cand_n, (size_t)max_batch * cap * sizeof(int)));
    VIORB_HIP_TRY(hipMalloc(&h->d_cam, 16 * sizeof(double)));
    VIORB_HIP_TRY(hipMalloc(&h->d_gw, 3 * sizeof(double)));
    VIORB_HIP_TRY(hipMalloc(&h->d_inv_sigma2, 16 * sizeof(float)));
    VIORB_HIP_TRY(hipMemcpy(h->d_cam, cfg->cam, 16 * sizeof(double), hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(h->d_gw, cfg->gravity, 3 * sizeof(double), hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(h->d_inv_sigma2, cfg->inv_level_sigma2, 16 * sizeof(float), hipMemcpyHostToDevice));
    *out = h;
    return VIORB_OK;
}

int viorb_frontend_search_capacity(void) {                                 // largest cap the projection search's LDS plan holds
    int lo = 1, hi = 65535;
    while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (search_lds_bytes(mid, search_slot_n(mid)) <= 160 * 1024) lo = mid; else hi = mid - 1; }
    return lo;
}

int viorb_frontend_destroy(viorb_frontend* h) {
    if (!h) return VIORB_OK;
    (void)hipSetDevice(h->device);
    if (h->d_cand) (void)hipFree(h->d_cand);
    if (h->d_pose_chi) (void)hipFree(h->d_pose_chi);
    if (h->d_cand_n) (void)hipFree(h->d_cand_n);
    if (h->d_search_work) (void)hipFree(h->d_search_work);
    if (h->d_lcand) (void)hipFree(h->d_lcand);
    if (h->d_lcand_n) (void)hipFree(h->d_lcand_n);
    if (h->d_cam) (void)hipFree(h->d_cam);
    if (h->d_gw) (void)hipFree(h->d_gw);
    if (h->d_inv_sigma2) (void)hipFree(h->d_inv_sigma2);
    if (h->d_scale) (void)hipFree(h->d_scale);
    delete h;
    return VIORB_OK;
}

#define FE_CHECK_BATCH(h, batch)                                                              \
    VIORB_REQUIRE(h, "null handle");                                                          \
    VIORB_REQUIRE((batch) >= 1 && (batch) <= (h)->max_batch, "batch out of range");           \
    VIORB_HIP_TRY(hipSetDevice((h)->device))

int viorb_frontend_grid_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count, int batch,
                               int32_t* cell_start, int32_t* cell_idx, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(kps && count && cell_start && cell_idx, "null array");
    ProfScope ps("k_frame_grid", (hipStream_t)stream);
    hipLaunchKernelGGL(k_frame_grid, dim3(batch), dim3(256), (size_t)h->sort_n * 4, (hipStream_t)stream, kps, count, h->cap,
                       h->cfg.min_x, h->cfg.min_y, h->wInv, h->hInv, cell_start, cell_idx, h->sort_n);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

static UndistortArgs undistort_args(const float* intr4, const float* dist5) {
    UndistortArgs A;
    A.fx = intr4[0]; A.fy = intr4[1]; A.cx = intr4[2]; A.cy = intr4[3];
    for (int i = 0; i < 5; i++) A.k[i] = dist5[i];
    return A;
}

int viorb_frontend_undistort_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count, int batch, viorb_keypoint* kps_un, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(kps && count && kps_un, "null array");
    const float intr4[4] = {h->cfg.fx, h->cfg.fy, h->cfg.cx, h->cfg.cy};
    ProfScope ps("k_undistort", (hipStream_t)stream);
    hipLaunchKernelGGL(k_undistort, dim3((h->cap + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, kps, count, h->cap,
                       undistort_args(intr4, h->cfg.dist_coef), h->cfg.dist_coef[0] != 0.0f ? 1 : 0, kps_un);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_undistort_points(const float* xy, int n, const float* intr4, const float* dist5, float* xy_out) {
    VIORB_REQUIRE(xy && intr4 && dist5 && xy_out && n >= 0, "null argument");
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    if (n == 0) return VIORB_OK;
    std::vector<viorb_keypoint> rec((size_t)n);
    for (int i = 0; i < n; i++) { rec[i] = viorb_keypoint{}; rec[i].x = xy[2 * i]; rec[i].y = xy[2 * i + 1]; }
    viorb_keypoint* d = nullptr;
    VIORB_HIP_TRY(hipMalloc(&d, sizeof(viorb_keypoint) * (size_t)n));
    hipError_t e = hipMemcpy(d, rec.data(), sizeof(viorb_keypoint) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256, 1), dim3(256), 0, (hipStream_t)0, d, (const int*)nullptr, n, undistort_args(intr4, dist5),
                           dist5[0] != 0.0f ? 1 : 0, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(rec.data(), d, sizeof(viorb_keypoint) * (size_t)n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    VIORB_HIP_TRY(e);
    for (int i = 0; i < n; i++) { xy_out[2 * i] = rec[i].x; xy_out[2 * i + 1] = rec[i].y; }
    return VIORB_OK;
}

// Frame::ComputeImageBounds, reference src/Frame.cc:616-644
int viorb_image_bounds(int width, int height, const float* intr4, const float* dist5, float* b) {
    VIORB_REQUIRE(intr4 && dist5 && b && width > 0 && height > 0, "null argument / empty image");
    if (dist5[0] != 0.0f) {
        float mat[8] = {0.0f, 0.0f, (float)width, 0.0f, 0.0f, (float)height, (float)width, (float)height};
        int rc = viorb_undistort_points(mat, 4, intr4, dist5, mat);
        if (rc != VIORB_OK) return rc;
        b[0] = std::min(mat[0], mat[4]); b[1] = std::max(mat[2], mat[6]);
        b[2] = std::min(mat[1], mat[3]); b[3] = std::max(mat[5], mat[7]);
    } else {
        if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
        b[0] = 0.0f; b[1] = (float)width; b[2] = 0.0f; b[3] = (float)height;
    }
    return VIORB_OK;
}

int viorb_frontend_imu_predict_device(viorb_frontend* h, const double* imu, int n_imu, const double* t_last, const double* t_cur,
                                      const double* last_ns, int batch, double* preint, double* cur_ns, float* pose12, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(imu && t_last && t_cur && last_ns && preint && cur_ns && pose12 && n_imu >= 1, "null array / n_imu < 1");
    ProfScope ps("k_imu_predict", (hipStream_t)stream);
    hipLaunchKernelGGL(k_imu_predict, dim3(batch), dim3(128), 0, (hipStream_t)stream, imu, n_imu, t_last, t_cur, last_ns, h->d_gw,
                       h->d_cam, h->cfg.gyr_meas_cov, h->cfg.acc_meas_cov, preint, cur_ns, pose12);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

// the stereo arguments of the call in progress on this thread (set by viorb_frontend_search_projection_stereo_device only)
struct StereoSearchArgs { const float* cur_uright = nullptr; const float* last_pose12 = nullptr; float bf = 0, mb = 0; };
static thread_local StereoSearchArgs g_stereo_args;

int viorb_frontend_search_projection_retry_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc,
                                                  const int32_t* cur_count, const int32_t* cell_start, const int32_t* cell_idx,
                                                  const float* pose12, const viorb_keypoint* last_kps, const int32_t* last_count,
                                                  const uint8_t* last_flags, const float* last_Pw, const uint8_t* last_desc, float th,
                                                  int retry_below, int batch, int32_t* cur_match, int32_t* nmatches, int32_t* status,
                                                  void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(cur_kps && cur_desc && cur_count && cell_start && cell_idx && pose12 && last_kps && last_count && last_flags &&
                  last_Pw && last_desc && cur_match && nmatches && status, "null array");
    SearchArgs A;
    A.cur_kps = cur_kps; A.cur_desc = cur_desc; A.cur_count = cur_count; A.cell_start = cell_start; A.cell_idx = cell_idx;
    A.pose12 = pose12; A.last_kps = last_kps; A.last_count = last_count; A.last_flags = last_flags; A.last_Pw = last_Pw;
    A.last_desc = last_desc; A.cur_match = cur_match; A.nmatches = nmatches; A.status = status;
    A.cand = h->d_cand; A.cand_n = h->d_cand_n; A.cap = h->cap;
    A.minX = h->cfg.min_x; A.maxX = h->cfg.max_x; A.minY = h->cfg.min_y; A.maxY = h->cfg.max_y; A.wInv = h->wInv; A.hInv = h->hInv;
    A.fx = h->cfg.fx; A.fy = h->cfg.fy; A.cx = h->cfg.cx; A.cy = h->cfg.cy; A.th = th;
    for (int i = 0; i < 16; i++) A.scale[i] = h->cfg.scale_factors[i];
    A.check_ori = h->cfg.check_orientation;
    A.skip_if_at_least = retry_below > 0 ? retry_below : 0;
    A.cur_uright = g_stereo_args.cur_uright; A.last_pose12 = g_stereo_args.last_pose12; A.bf = g_stereo_args.bf; A.mb = g_stereo_args.mb;
    ProfScope ps("k_search_projection", (hipStream_t)stream);
    A.slot_n = search_slot_n(h->cap);
    A.work = nullptr; A.work_bytes = 0;
    if (search_lds_bytes(h->cap, A.slot_n) > 160 * 1024) {               // more keypoints per frame than LDS holds: the work arrays in global memory
        const size_t per = (search_lds_bytes(h->cap, 0) + 255) & ~(size_t)255;
        if (h->search_work_bytes < per * h->max_batch) {
            VIORB_HIP_TRY(hipDeviceSynchronize());               // the handle's scratch is shared by its searches: nothing may still walk the old allocation
            if (h->d_search_work) (void)hipFree(h->d_search_work);
            h->d_search_work = nullptr; h->search_work_bytes = 0;
            VIORB_HIP_TRY(hipMalloc(&h->d_search_work, per * h->max_batch));
            h->search_work_bytes = per * h->max_batch;
        }
        A.slot_n = 0; A.work = h->d_search_work; A.work_bytes = per;
        hipLaunchKernelGGL(k_search_projection<true>, dim3(batch), dim3(SEARCH_THREADS), 0, (hipStream_t)stream, A);
    } else
    hipLaunchKernelGGL(k_search_projection<false>, dim3(batch), dim3(SEARCH_THREADS), search_lds_bytes(h->cap, A.slot_n), (hipStream_t)stream, A);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_search_projection_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc,
                                            const int32_t* cur_count, const int32_t* cell_start, const int32_t* cell_idx,
                                            const float* pose12, const viorb_keypoint* last_kps, const int32_t* last_count,
                                            const uint8_t* last_flags, const float* last_Pw, const uint8_t* last_desc, float th,
                                            int batch, int32_t* cur_match, int32_t* nmatches, int32_t* status, void* stream) {
    return viorb_frontend_search_projection_retry_device(h, cur_kps, cur_desc, cur_count, cell_start, cell_idx, pose12, last_kps, last_count, last_flags,
                                                         last_Pw, last_desc, th, 0, batch, cur_match, nmatches, status, stream);
}

int viorb_frontend_search_projection_stereo_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc,
                                                   const int32_t* cur_count, const float* cur_uright, const int32_t* cell_start,
                                                   const int32_t* cell_idx, const float* pose12, const float* last_pose12,
                                                   const viorb_keypoint* last_kps, const int32_t* last_count, const uint8_t* last_flags,
                                                   const float* last_Pw, const uint8_t* last_desc, float th, float bf, float mb,
                                                   int retry_below, int batch, int32_t* cur_match, int32_t* nmatches, int32_t* status,
                                                   void* stream) {
    VIORB_REQUIRE(cur_uright && last_pose12, "null stereo array");
    g_stereo_args.cur_uright = cur_uright; g_stereo_args.last_pose12 = last_pose12; g_stereo_args.bf = bf; g_stereo_args.mb = mb;
    const int rc = viorb_frontend_search_projection_retry_device(h, cur_kps, cur_desc, cur_count, cell_start, cell_idx, pose12, last_kps, last_count,
                                                                 last_flags, last_Pw, last_desc, th, retry_below, batch, cur_match, nmatches, status, stream);
    g_stereo_args = StereoSearchArgs();
    return rc;
}

// the stereo arguments of the local-points search in progress on this thread (set by viorb_frontend_search_local_points_stereo_device only)
struct StereoLocalArgs { const float* cur_uright = nullptr; float bf = 0; float* frustum_xr = nullptr; };
static thread_local StereoLocalArgs g_stereo_local;

int viorb_frontend_search_local_points_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const int32_t* cur_count,
                                              const int32_t* cell_start, const int32_t* cell_idx, const float* pose12, const float* pts_f,
                                              const uint8_t* pts_flags, const uint8_t* pts_desc, const int32_t* pts_count, int pcap, float th,
                                              float nnratio, const uint8_t* cur_owner_obs, int batch, int32_t* match, int32_t* nmatches,
                                              float* frustum, int32_t* status, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(cur_kps && cur_desc && cur_count && cell_start && cell_idx && pose12 && pts_f && pts_flags && pts_desc && pts_count &&
                  cur_owner_obs && match && nmatches && status, "null array");
    VIORB_REQUIRE(pcap >= 1 && pcap <= 65535, "1 <= pcap <= 65535");
    const int lslot = local_search_lds_bytes(h->cap, pcap, LOCAL_SLOT) <= 160 * 1024 ? LOCAL_SLOT : 0;      // the slot cache only while it fits
    const size_t lds = local_search_lds_bytes(h->cap, pcap, lslot);
    const bool gw = lds > 160 * 1024;                                    // more keypoints / local points than LDS holds: work arrays in global memory
    if (h->lcand_pcap < pcap) {
        if (h->d_lcand) (void)hipFree(h->d_lcand);
        if (h->d_lcand_n) (void)hipFree(h->d_lcand_n);
        h->d_lcand = nullptr; h->d_lcand_n = nullptr; h->lcand_pcap = 0;
        VIORB_HIP_TRY(hipMalloc(&h->d_lcand, (size_t)h->max_batch * pcap * CAND_CAP * sizeof(uint32_t)));
        VIORB_HIP_TRY(hipMalloc(&h->d_lcand_n, (size_t)h->max_batch * pcap * sizeof(int)));
        h->lcand_pcap = pcap;
        if (lds > 64 * 1024 && !gw)
            VIORB_HIP_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(k_search_local_points<false>), lds));
    }
    LocalSearchArgs A;
    A.cur_kps = cur_kps; A.cur_desc = cur_desc; A.cur_count = cur_count; A.cell_start = cell_start; A.cell_idx = cell_idx; A.pose12 = pose12;
    A.pts_f = pts_f; A.pts_flags = pts_flags; A.pts_desc = pts_desc; A.pts_count = pts_count; A.cur_owner_obs = cur_owner_obs;
    A.match = match; A.nmatches = nmatches; A.status = status; A.frustum = frustum; A.cand = h->d_lcand; A.cand_n = h->d_lcand_n;
    A.cur_uright = g_stereo_local.cur_uright; A.bf = g_stereo_local.bf; A.frustum_xr = g_stereo_local.frustum_xr;
    A.cap = h->cap; A.pcap = pcap; A.slot_n = lslot;
    A.minX = h->cfg.min_x; A.maxX = h->cfg.max_x; A.minY = h->cfg.min_y; A.maxY = h->cfg.max_y; A.wInv = h->wInv; A.hInv = h->hInv;
    A.fx = h->cfg.fx; A.fy = h->cfg.fy; A.cx = h->cfg.cx; A.cy = h->cfg.cy; A.th = th; A.nnratio = nnratio;
    A.log_sf = (float)log((double)h->cfg.scale_factors[h->cfg.nlevels > 1 ? 1 : 0]);   // Frame::mfLogScaleFactor = log(mfScaleFactor), rounded once
    for (int i = 0; i < 16; i++) A.scale[i] = h->cfg.scale_factors[i];
    A.nlevels = h->cfg.nlevels;
    VIORB_HIP_TRY(hipMemsetAsync(status, 0, sizeof(int32_t) * batch, (hipStream_t)stream));
    ProfScope ps("k_search_local_points", (hipStream_t)stream);
    A.work = nullptr; A.work_bytes = 0;
    if (gw) {
        const size_t per = (lds + 255) & ~(size_t)255;
        if (h->search_work_bytes < per * h->max_batch) {
            VIORB_HIP_TRY(hipDeviceSynchronize());               // the handle's scratch is shared by its searches: nothing may still walk the old allocation
            if (h->d_search_work) (void)hipFree(h->d_search_work);
            h->d_search_work = nullptr; h->search_work_bytes = 0;
            VIORB_HIP_TRY(hipMalloc(&h->d_search_work, per * h->max_batch));
            h->search_work_bytes = per * h->max_batch;
        }
        A.work = h->d_search_work; A.work_bytes = per;
        hipLaunchKernelGGL(k_search_local_points<true>, dim3(batch), dim3(SEARCH_THREADS), 0, (hipStream_t)stream, A);
    } else
    hipLaunchKernelGGL(k_search_local_points<false>, dim3(batch), dim3(SEARCH_THREADS), lds, (hipStream_t)stream, A);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_search_local_points_stereo_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const int32_t* cur_count,
                                                     const float* cur_uright, float bf, const int32_t* cell_start, const int32_t* cell_idx,
                                                     const float* pose12, const float* pts_f, const uint8_t* pts_flags, const uint8_t* pts_desc,
                                                     const int32_t* pts_count, int pcap, float th, float nnratio, const uint8_t* cur_owner_obs,
                                                     int batch, int32_t* match, int32_t* nmatches, float* frustum, float* frustum_xr, int32_t* status,
                                                     void* stream) {
    VIORB_REQUIRE(cur_uright, "null stereo array");
    g_stereo_local.cur_uright = cur_uright; g_stereo_local.bf = bf; g_stereo_local.frustum_xr = frustum_xr;
    const int rc = viorb_frontend_search_local_points_device(h, cur_kps, cur_desc, cur_count, cell_start, cell_idx, pose12, pts_f, pts_flags, pts_desc, pts_count,
                                                             pcap, th, nnratio, cur_owner_obs, batch, match, nmatches, frustum, status, stream);
    g_stereo_local = StereoLocalArgs();
    return rc;
}

int viorb_frontend_build_observations_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count, const int32_t* match,
                                             const float* match_Pw, int batch, double* obs, int32_t* obs_index, int32_t* n_obs,
                                             void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(kps && count && match && match_Pw && obs && obs_index && n_obs, "null array");
    ProfScope ps("k_build_observations", (hipStream_t)stream);
    hipLaunchKernelGGL(k_build_observations, dim3(batch), dim3(256), 0, (hipStream_t)stream, kps, count, match, match_Pw,
                       h->d_inv_sigma2, h->cap, obs, obs_index, n_obs);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_build_observations2_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count, const int32_t* match_a,
                                              const float* Pw_a, const int32_t* match_b, const float* pts_b, int stride_b, int batch,
                                              double* obs, int32_t* obs_index, int32_t* n_obs, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(kps && count && match_a && Pw_a && match_b && pts_b && obs && obs_index && n_obs && stride_b >= 1, "null array");
    ProfScope ps("k_build_observations", (hipStream_t)stream);
    hipLaunchKernelGGL(k_build_observations2, dim3(batch), dim3(256), 0, (hipStream_t)stream, kps, count, match_a, Pw_a, match_b, pts_b, stride_b,
                       h->d_inv_sigma2, h->cap, obs, obs_index, n_obs);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_discard_outliers_device(viorb_frontend* h, int32_t* match, const int32_t* obs_index, const uint8_t* outlier,
                                           const int32_t* n_obs, const uint8_t* pt_flags, int batch, uint8_t* owner_obs, int32_t* n_map,
                                           void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(match && obs_index && outlier && n_obs && pt_flags && owner_obs && n_map, "null array");
    ProfScope ps("k_discard_outliers", (hipStream_t)stream);
    hipLaunchKernelGGL(k_discard_outliers, dim3(batch), dim3(256), 0, (hipStream_t)stream, match, obs_index, outlier, n_obs, pt_flags, h->cap,
                       owner_obs, n_map);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_pose_from_navstate_device(viorb_frontend* h, const double* ns, int batch, float* pose12, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(ns && pose12, "null array");
    hipLaunchKernelGGL(k_pose_from_navstate, dim3((batch + 63) / 64), dim3(64), 0, (hipStream_t)stream, ns, h->d_cam, batch, pose12);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_synth_local_points_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count, const double* pose12, const float* Pw,
                                    int batch, float* pts_f, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(kps && count && pose12 && Pw && pts_f, "null array");
    if (!h->d_scale) {
        VIORB_HIP_TRY(hipMalloc(&h->d_scale, 16 * sizeof(float)));
        VIORB_HIP_TRY(hipMemcpy(h->d_scale, h->cfg.scale_factors, 16 * sizeof(float), hipMemcpyHostToDevice));
    }
    ProfScope ps("k_synth_plane_points", (hipStream_t)stream);
    hipLaunchKernelGGL(k_synth_local_points, dim3((h->cap + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, kps, count, h->cap, pose12, Pw,
                       h->d_scale, h->cfg.nlevels, pts_f);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_roll_device(viorb_frontend* h, const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const int32_t* cur_count,
                               viorb_keypoint* last_kps, uint8_t* last_desc, int32_t* last_count, const float* last_pts_f,
                               const uint8_t* last_flags, float* loc_pts_f, uint8_t* loc_desc, uint8_t* loc_flags, int local_frames,
                               int shift_local, const double* ns_src, double* last_ns, double* prior_ns, const double* t_src, double* t_last,
                               const double* marg_src, double* marg_dst, int batch, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(cur_kps && cur_desc && cur_count && last_kps && last_desc && last_count && ns_src && last_ns && prior_ns && t_src && t_last, "null array");
    VIORB_REQUIRE(!loc_pts_f || (loc_desc && loc_flags && last_pts_f && last_flags && local_frames >= 1), "incomplete local-map tables");
    VIORB_REQUIRE((marg_src == nullptr) == (marg_dst == nullptr), "marg_src / marg_dst");
    RollArgs A;
    A.cur_kps = cur_kps; A.cur_desc = cur_desc; A.cur_count = cur_count; A.last_kps = last_kps; A.last_desc = last_desc; A.last_count = last_count;
    A.last_pts_f = last_pts_f; A.last_flags = last_flags; A.loc_pts_f = loc_pts_f; A.loc_desc = loc_desc; A.loc_flags = loc_flags;
    A.local_frames = local_frames; A.shift_local = shift_local;
    A.ns_src = ns_src; A.last_ns = last_ns; A.prior_ns = prior_ns; A.t_src = t_src; A.t_last = t_last; A.marg_src = marg_src; A.marg_dst = marg_dst;
    A.cap = h->cap;
    ProfScope ps("k_roll", (hipStream_t)stream);
    hipLaunchKernelGGL(k_roll, dim3((h->cap + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, A);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_fuse_device(viorb_frontend* h, const viorb_keypoint* kps, const uint8_t* desc, const float* uright, const int32_t* count,
                               const int32_t* cell_start, const int32_t* cell_idx, const float* pose12, const float* pts_f,
                               const uint8_t* pts_valid, const uint8_t* pts_desc, const int32_t* pts_count, int pcap, float th, float bf, int batch,
                               int32_t* best_idx, int32_t* nfused, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(kps && desc && uright && count && cell_start && cell_idx && pose12 && pts_f && pts_valid && pts_desc && pts_count && best_idx && nfused,
                  "null array");
    VIORB_REQUIRE(pcap >= 1, "pcap >= 1");
    FuseArgs A;
    A.kps = kps; A.desc = desc; A.uright = uright; A.count = count; A.cell_start = cell_start; A.cell_idx = cell_idx; A.pose12 = pose12;
    A.pts_f = pts_f; A.pts_valid = pts_valid; A.pts_desc = pts_desc; A.pts_count = pts_count; A.best_idx = best_idx; A.nfused = nfused;
    A.cap = h->cap; A.pcap = pcap; A.nlevels = h->cfg.nlevels;
    A.minX = h->cfg.min_x; A.maxX = h->cfg.max_x; A.minY = h->cfg.min_y; A.maxY = h->cfg.max_y; A.wInv = h->wInv; A.hInv = h->hInv;
    A.fx = h->cfg.fx; A.fy = h->cfg.fy; A.cx = h->cfg.cx; A.cy = h->cfg.cy; A.bf = bf; A.th = th;
    A.log_sf = (float)log((double)h->cfg.scale_factors[h->cfg.nlevels > 1 ? 1 : 0]);
    for (int i = 0; i < 16; i++) { A.scale[i] = h->cfg.scale_factors[i]; A.inv_sigma2[i] = h->cfg.inv_level_sigma2[i]; }
    ProfScope ps("k_fuse", (hipStream_t)stream);
    hipLaunchKernelGGL(k_fuse, dim3(batch), dim3(1024), 0, (hipStream_t)stream, A);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

// Shape of the VI pose solver for every later launch of this process: P problems per workgroup, WPP wavefronts per problem (one of 1,2 1,3 1,4
// 1,8 2,2 2,3 2,4 4,1 4,2); 0, 0 = by batch size (the default). The solver's results do not depend on it beyond the grouping of floating-point
// sums; the test-suite runs every parity test of the solver under the shapes the batch sizes of a deployment select.
int viorb_frontend_set_pose_shape(int problems_per_workgroup, int wavefronts_per_problem) {
    if (problems_per_workgroup == 0 && wavefronts_per_problem == 0) { g_pose_shape_override.store(0); return VIORB_OK; }
    const int c = problems_per_workgroup * 16 + wavefronts_per_problem;
    const int ok[] = {1 * 16 + 2, 1 * 16 + 3, 1 * 16 + 4, 1 * 16 + 8, 2 * 16 + 2, 2 * 16 + 3, 2 * 16 + 4, 4 * 16 + 1, 4 * 16 + 2};
    for (int v : ok) if (v == c) { g_pose_shape_override.store(c); return VIORB_OK; }
    set_error("viorb_frontend_set_pose_shape: unsupported (P, WPP) = (%d, %d)", problems_per_workgroup, wavefronts_per_problem);
    return VIORB_ERR_INVALID_ARG;
}

int viorb_frontend_pose_opt_device(viorb_frontend* h, int variant, int compute_marg, const double* cur_ns, const double* last_ns,
                                   const double* prior_ns, const double* marg_cov_inv, const double* preint, const double* obs_cur,
                                   const int32_t* n_cur, const double* obs_last, const int32_t* n_last, int batch, double* out_ns,
                                   double* out_last_ns, uint8_t* outlier_cur, uint8_t* outlier_last, double* marg_out, double* info,
                                   void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(variant == 0 || variant == 1, "variant must be 0 (KeyFrame) or 1 (Frame)");
    VIORB_REQUIRE(cur_ns && last_ns && preint && obs_cur && n_cur && out_ns && outlier_cur && info, "null array");
    VIORB_REQUIRE(!variant || (prior_ns && marg_cov_inv && obs_last && n_last && outlier_last), "variant 1 needs prior and last-frame arrays");
    VIORB_REQUIRE(!compute_marg || marg_out, "marg_out is NULL");
    PoseOptArgs A;
    A.variant = variant; A.compute_marg = compute_marg; A.cap = h->cap;
    A.cur_ns = cur_ns; A.last_ns = last_ns; A.prior_ns = prior_ns; A.marg_cov_inv = marg_cov_inv; A.preint = preint;
    A.gw = h->d_gw; A.cam = h->d_cam; A.obs_cur = obs_cur; A.obs_last = variant ? obs_last : nullptr; A.n_cur = n_cur; A.n_last = n_last;
    A.out_ns = out_ns; A.out_last_ns = out_last_ns; A.marg_out = marg_out; A.info = info;
    A.outlier_cur = outlier_cur; A.outlier_last = variant ? outlier_last : nullptr; A.acc_bias_rw2 = h->cfg.acc_bias_rw2; A.chi_store = h->d_pose_chi;
    A.variant_arr = nullptr; A.skip = nullptr;
    ProfScope ps("k_pose_opt_vi", (hipStream_t)stream);
    { const int rc = launch_pose_opt_vi(A, batch, (hipStream_t)stream); if (rc != VIORB_OK) return rc; }
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_pose_opt_select_device(viorb_frontend* h, const uint8_t* variant, const uint8_t* skip, int compute_marg,
                                          const double* cur_ns, const double* last_ns, const double* prior_ns, const double* marg_cov_inv,
                                          const double* preint, const double* obs_cur, const int32_t* n_cur, const double* obs_last,
                                          const int32_t* n_last, int batch, double* out_ns, double* out_last_ns, uint8_t* outlier_cur,
                                          uint8_t* outlier_last, double* marg_out, double* info, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(cur_ns && last_ns && preint && obs_cur && n_cur && out_ns && outlier_cur && info, "null array");
    VIORB_REQUIRE(prior_ns && marg_cov_inv && obs_last && n_last && outlier_last, "the Frame variant needs prior and last-frame arrays");
    VIORB_REQUIRE(!compute_marg || marg_out, "marg_out is NULL");
    PoseOptArgs A;
    A.variant = 1; A.compute_marg = compute_marg; A.cap = h->cap;
    A.cur_ns = cur_ns; A.last_ns = last_ns; A.prior_ns = prior_ns; A.marg_cov_inv = marg_cov_inv; A.preint = preint;
    A.gw = h->d_gw; A.cam = h->d_cam; A.obs_cur = obs_cur; A.obs_last = obs_last; A.n_cur = n_cur; A.n_last = n_last;
    A.out_ns = out_ns; A.out_last_ns = out_last_ns; A.marg_out = marg_out; A.info = info;
    A.outlier_cur = outlier_cur; A.outlier_last = outlier_last; A.acc_bias_rw2 = h->cfg.acc_bias_rw2; A.chi_store = h->d_pose_chi;
    A.variant_arr = variant; A.skip = skip;
    ProfScope ps("k_pose_opt_vi", (hipStream_t)stream);
    { const int rc = launch_pose_opt_vi(A, batch, (hipStream_t)stream); if (rc != VIORB_OK) return rc; }
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_self_index_device(viorb_frontend* h, const uint8_t* flags, const int32_t* count, int batch, int32_t* self_index, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(flags && count && self_index, "null array");
    hipLaunchKernelGGL(k_self_index, dim3((h->cap + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, flags, count, h->cap, self_index);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_synth_plane_points_device(viorb_frontend* h, const viorb_keypoint* kps, const int32_t* count, const double* pose12,
                                    double z0, int batch, float* Pw, uint8_t* flags, int32_t* self_index, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(kps && count && pose12 && Pw && flags, "null array");
    ProfScope ps("k_synth_plane_points", (hipStream_t)stream);
    hipLaunchKernelGGL(k_synth_plane_points, dim3((h->cap + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, kps, count, h->cap,
                       pose12, h->cfg.cam[0], h->cfg.cam[1], h->cfg.cam[2], h->cfg.cam[3], z0, Pw, flags, self_index);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_frontend_pose_opt_se3_device(viorb_frontend* h, const float* pose12, const double* obs7, const int32_t* n_obs, double bf, int batch,
                                       float* out_pose12, uint8_t* outlier, double* info, void* stream) {
    FE_CHECK_BATCH(h, batch);
    VIORB_REQUIRE(pose12 && obs7 && n_obs && out_pose12 && outlier && info, "null array");
    Se3Args A;
    A.pose12 = pose12; A.obs7 = obs7; A.n_obs = n_obs; A.cap = h->cap;
    A.fx = (double)h->cfg.fx; A.fy = (double)h->cfg.fy; A.cx = (double)h->cfg.cx; A.cy = (double)h->cfg.cy; A.bf = bf;   // e->fx = pFrame->fx (float -> double)
    A.out_pose12 = out_pose12; A.outlier = outlier; A.info = info;
    ProfScope ps("k_pose_opt_se3", (hipStream_t)stream);
    hipLaunchKernelGGL(k_pose_opt_se3, dim3(batch), dim3(256), 0, (hipStream_t)stream, A);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

// ---- host-buffer drop-ins ------------------------------------------------------------------------
int viorb_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    uint32_t x[8], y[8];
    memcpy(x, a, 32); memcpy(y, b, 32);
    return hamming256(x, y);
}

} // extern "C"

namespace {
// Scratch of the host-buffer drop-ins: a thread-local device arena that is bump-allocated per call and kept between calls (a Tracking
// thread calls these once or twice per frame; hipMalloc / hipFree per buffer cost 0.3-0.5 ms per call and synchronise the whole device).
struct HostArena {
    struct Block { void* p; size_t bytes; };
    std::vector<Block> blocks; size_t used = 0; int device = -1;
    uint8_t* pin = nullptr; size_t pin_bytes = 0;          // page-locked mirror of the newest block: a call's inputs go up in ONE copy, its outputs come down in one
    ~HostArena() { for (auto& b : blocks) (void)hipFree(b.p); if (pin) (void)hipHostFree(pin); }
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (!blocks.empty() && used + bytes <= blocks.back().bytes) { void* r = (char*)blocks.back().p + used; used += bytes; return r; }
        const size_t want = std::max<size_t>(bytes * 2, blocks.empty() ? (size_t)4 << 20 : blocks.back().bytes * 2);
        void* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) return nullptr;
        blocks.push_back({p, want}); used = bytes;
        return p;
    }
    void reset(int dev) {                 // start of a call: keep only the largest block
        if (dev != device) { for (auto& b : blocks) (void)hipFree(b.p); blocks.clear(); device = dev; }
        while (blocks.size() > 1) { (void)hipFree(blocks.front().p); blocks.erase(blocks.begin()); }
        used = 0;
        if (!blocks.empty() && pin_bytes < blocks.back().bytes) {
            if (pin) (void)hipHostFree(pin);
            pin = nullptr; pin_bytes = 0;
            if (hipHostMalloc(reinterpret_cast<void**>(&pin), blocks.back().bytes) == hipSuccess) pin_bytes = blocks.back().bytes; else pin = nullptr;
        }
    }
};
static thread_local HostArena g_host_arena;
int current_device();
// Per-call view of the arena. Inputs are staged in the page-locked mirror and go to the device in one asynchronous copy (flush); outputs
// are registered (down) and come back in one copy + one synchronisation (fetch). A call used to issue ~15 small pageable copies and
// five or six blocking downloads: 0.2 ms of the 0.5 ms a host-buffer pose solve cost. The mirror covers the block the arena held when
// the call began (base0: it stays allocated until the next call's reset()); a buffer taken after the arena grew during this very call
// lies in a newer block and falls back to direct copies. Whether a buffer is mirrored is decided by its ADDRESS alone, so that a buffer
// of the first block keeps going through the mirror after the arena has grown (a zero-staged buffer followed by a put() of the real
// data would otherwise be overwritten by flush()'s copy of the staged zeros).
struct DevBuf {
    struct Out { void* dst; const void* src; size_t bytes; };
    std::vector<Out> outs; size_t lo = (size_t)-1, hi = 0; void* base0 = nullptr; size_t base0_bytes = 0;
    DevBuf() {
        g_host_arena.reset(current_device());
        if (!g_host_arena.blocks.empty()) { base0 = g_host_arena.blocks.back().p; base0_bytes = std::min(g_host_arena.blocks.back().bytes, g_host_arena.pin_bytes); }
    }
    bool mirrored(const void* d, size_t bytes) const {
        return g_host_arena.pin && base0 && (const char*)d >= (const char*)base0 && (const char*)d + bytes <= (const char*)base0 + base0_bytes;
    }
    int put(void* d, const void* hsrc, size_t bytes) {       // host bytes -> the device buffer d (staged when d is mirrored)
        if (!bytes) return VIORB_OK;
        if (mirrored(d, bytes)) {
            const size_t off = (size_t)((char*)d - (char*)base0);
            if (hsrc) memcpy(g_host_arena.pin + off, hsrc, bytes); else memset(g_host_arena.pin + off, 0, bytes);
            lo = std::min(lo, off); hi = std::max(hi, off + bytes);
            return VIORB_OK;
        }
        const hipError_t e = hsrc ? hipMemcpyAsync(d, hsrc, bytes, hipMemcpyHostToDevice, nullptr) : hipMemsetAsync(d, 0, bytes, nullptr);
        if (e != hipSuccess) { set_error("H2D failed: %s", hipGetErrorString(e)); return VIORB_ERR_HIP; }
        return VIORB_OK;
    }
    template <class T> int up(T** d, const T* hsrc, size_t n) {
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        *d = (T*)g_host_arena.take(bytes);
        if (!*d) { set_error("hipMalloc failed for the host drop-in scratch"); return VIORB_ERR_HIP; }
        return n ? put(*d, hsrc, n * sizeof(T)) : VIORB_OK;
    }
    int flush() {                                             // before the first launch that reads the staged inputs
        if (hi > lo) {
            const hipError_t e = hipMemcpyAsync((char*)base0 + lo, g_host_arena.pin + lo, hi - lo, hipMemcpyHostToDevice, nullptr);
            if (e != hipSuccess) { set_error("H2D failed: %s", hipGetErrorString(e)); return VIORB_ERR_HIP; }
        }
        lo = (size_t)-1; hi = 0;
        return VIORB_OK;
    }
    void down(void* dst, const void* dsrc, size_t bytes) { if (dst && bytes) outs.push_back({dst, dsrc, bytes}); }
    int fetch() {                                             // after the last launch: one download of the span of the outputs, one synchronisation
        size_t a = (size_t)-1, b = 0;
        for (const Out& o : outs) if (mirrored(o.src, o.bytes)) { const size_t off = (size_t)((const char*)o.src - (const char*)base0); a = std::min(a, off); b = std::max(b, off + o.bytes); }
        hipError_t e = hipSuccess;
        if (b > a) e = hipMemcpyAsync(g_host_arena.pin + a, (const char*)base0 + a, b - a, hipMemcpyDeviceToHost, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) { set_error("D2H failed: %s", hipGetErrorString(e)); return VIORB_ERR_HIP; }
        for (const Out& o : outs) {
            if (mirrored(o.src, o.bytes)) memcpy(o.dst, g_host_arena.pin + ((const char*)o.src - (const char*)base0), o.bytes);
            else if ((e = hipMemcpy(o.dst, o.src, o.bytes, hipMemcpyDeviceToHost)) != hipSuccess) { set_error("D2H failed: %s", hipGetErrorString(e)); return VIORB_ERR_HIP; }
        }
        outs.clear();
        return VIORB_OK;
    }
};
#define FE_TRY(x) do { int _rc = (x); if (_rc != VIORB_OK) return _rc; } while (0)
// the host-buffer drop-ins run on the calling thread's current HIP device (hipSetDevice / torch.cuda.set_device), not on device 0
int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) d = 0; return d; }
// One front-end handle per calling thread, re-used across host drop-in calls: re-configured in place when only the camera / bounds /
// tables change, re-created when a call needs more keypoint capacity or runs on another device.
struct HostFrontend { viorb_frontend* h = nullptr; ~HostFrontend() { if (h) viorb_frontend_destroy(h); } };
static thread_local HostFrontend g_host_fe;
static int host_frontend(const viorb_frontend_config& c, int cap, viorb_frontend** out, bool for_search = false) {
    const int dev = current_device();
    viorb_frontend*& h = g_host_fe.h;
    // the searches' LDS plan is sized by the handle's pitch: a handle grown for a pose solve with more edges than the searches hold
    // keypoints (viorb_frontend_search_capacity) is replaced by one of the caller's size when a search comes
    const int lim = viorb_frontend_search_capacity();
    if (h && (h->cap < cap || h->device != dev || (for_search && h->cap > lim && cap <= lim))) { viorb_frontend_destroy(h); h = nullptr; }
    if (!h) {
        int want = 1024; while (want < cap) want *= 2;
        want = std::min(want, 32768);
        if (cap <= lim) want = std::min(want, lim);                   // growth in powers of two, but not past what the searches hold
        const int rc = viorb_frontend_create(&c, 1, std::max(cap, want), dev, &h);
        if (rc != VIORB_OK) { h = nullptr; return rc; }
    } else {
        viorb_frontend_config cc = c;
        if (cc.gyr_meas_cov <= 0) cc.gyr_meas_cov = 2.0e-3 * 2.0e-3 * 200;
        if (cc.acc_meas_cov <= 0) cc.acc_meas_cov = 8.0e-3 * 8.0e-3 * 200;
        if (cc.acc_bias_rw2 <= 0) cc.acc_bias_rw2 = 5e-3 * 5e-3;
        if (memcmp(&h->cfg, &cc, sizeof(cc)) == 0) { *out = h; return VIORB_OK; }      // same camera / tables as the previous call: nothing to upload
        h->cfg = cc;
        h->wInv = static_cast<float>(GRID_COLS) / static_cast<float>(cc.max_x - cc.min_x);
        h->hInv = static_cast<float>(GRID_ROWS) / static_cast<float>(cc.max_y - cc.min_y);
        VIORB_HIP_TRY(hipSetDevice(dev));
        VIORB_HIP_TRY(hipMemcpyAsync(h->d_cam, cc.cam, 16 * sizeof(double), hipMemcpyHostToDevice, nullptr));
        VIORB_HIP_TRY(hipMemcpyAsync(h->d_gw, cc.gravity, 3 * sizeof(double), hipMemcpyHostToDevice, nullptr));
        VIORB_HIP_TRY(hipMemcpyAsync(h->d_inv_sigma2, cc.inv_level_sigma2, 16 * sizeof(float), hipMemcpyHostToDevice, nullptr));
        if (h->d_scale) VIORB_HIP_TRY(hipMemcpyAsync(h->d_scale, cc.scale_factors, 16 * sizeof(float), hipMemcpyHostToDevice, nullptr));
    }
    *out = h;
    return VIORB_OK;
}
viorb_frontend_config default_cfg() {
    viorb_frontend_config c; memset(&c, 0, sizeof(c));
    c.min_x = 0; c.max_x = 752; c.min_y = 0; c.max_y = 480; c.nlevels = 8; c.check_orientation = 1;
    float s = 1.f; for (int i = 0; i < 16; i++) { c.scale_factors[i] = s; c.inv_level_sigma2[i] = 1.f / (s * s); s *= 1.2f; }
    return c;
}
} // namespace

extern "C" {

static int search_by_projection_frame_host(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int ncur,
                                           const float bounds4[4], const float pose12[12], const float* last_pose12, const float intr4[4],
                                           float bf, float mb, const float* scale_factors, int nlevels, const viorb_keypoint* last_kps, int nlast,
                                           const uint8_t* last_flags, const float* last_Pw, const uint8_t* last_desc, float th,
                                           int check_orientation, int32_t* cur_match, int* nmatches) {
    VIORB_REQUIRE(bounds4 && pose12 && intr4 && scale_factors && nmatches && ncur >= 0 && nlast >= 0, "null array");
    VIORB_REQUIRE(nlevels >= 1 && nlevels <= 16, "nlevels must be 1..16");
    VIORB_REQUIRE(ncur == 0 || cur_match, "cur_match is NULL");
    *nmatches = 0;
    for (int i = 0; i < ncur; i++) cur_match[i] = -1;
    if (ncur == 0 || nlast == 0) return VIORB_OK;
    VIORB_REQUIRE(cur_kps && cur_desc && last_kps && last_flags && last_Pw && last_desc, "null array");
    viorb_frontend_config c = default_cfg();
    c.min_x = bounds4[0]; c.max_x = bounds4[1]; c.min_y = bounds4[2]; c.max_y = bounds4[3];
    c.fx = intr4[0]; c.fy = intr4[1]; c.cx = intr4[2]; c.cy = intr4[3];
    c.nlevels = nlevels; c.check_orientation = check_orientation;
    for (int i = 0; i < 16; i++) c.scale_factors[i] = scale_factors[i < nlevels ? i : nlevels - 1];
    const int cap = std::max(ncur, nlast);
    viorb_frontend* h = nullptr;
    FE_TRY(host_frontend(c, cap, &h, true));
    const int hc = h->cap;                                 // staging arrays are pitched by the (cached, possibly larger) handle capacity
    DevBuf B; viorb_keypoint *d_ck, *d_lk; uint8_t *d_cd, *d_ld, *d_lf; float *d_lp, *d_pose; int *d_cc, *d_lc, *d_cs, *d_ci, *d_m, *d_nm, *d_st;
    FE_TRY(B.up(&d_ck, (const viorb_keypoint*)nullptr, (size_t)hc)); FE_TRY(B.up(&d_lk, (const viorb_keypoint*)nullptr, (size_t)hc));
    FE_TRY(B.up(&d_cd, (const uint8_t*)nullptr, (size_t)hc * 32)); FE_TRY(B.up(&d_ld, (const uint8_t*)nullptr, (size_t)hc * 32));
    FE_TRY(B.up(&d_lf, (const uint8_t*)nullptr, (size_t)hc)); FE_TRY(B.up(&d_lp, (const float*)nullptr, (size_t)hc * 3));
    FE_TRY(B.up(&d_pose, pose12, 12)); FE_TRY(B.up(&d_cc, &ncur, 1)); FE_TRY(B.up(&d_lc, &nlast, 1));
    FE_TRY(B.up(&d_cs, (const int*)nullptr, GRID_CELLS + 1)); FE_TRY(B.up(&d_ci, (const int*)nullptr, (size_t)hc));
    FE_TRY(B.up(&d_m, (const int*)nullptr, (size_t)hc)); FE_TRY(B.up(&d_nm, (const int*)nullptr, 1)); FE_TRY(B.up(&d_st, (const int*)nullptr, 1));
    FE_TRY(B.put(d_ck, cur_kps, sizeof(viorb_keypoint) * ncur)); FE_TRY(B.put(d_cd, cur_desc, (size_t)32 * ncur));
    FE_TRY(B.put(d_lk, last_kps, sizeof(viorb_keypoint) * nlast)); FE_TRY(B.put(d_ld, last_desc, (size_t)32 * nlast));
    FE_TRY(B.put(d_lf, last_flags, (size_t)nlast)); FE_TRY(B.put(d_lp, last_Pw, sizeof(float) * 3 * nlast));
    float *d_ur = nullptr, *d_lpose = nullptr;
    if (cur_uright) { FE_TRY(B.up(&d_ur, (const float*)nullptr, (size_t)hc)); FE_TRY(B.up(&d_lpose, last_pose12, 12)); FE_TRY(B.put(d_ur, cur_uright, sizeof(float) * ncur)); }
    FE_TRY(B.flush());
    FE_TRY(viorb_frontend_grid_device(h, d_ck, d_cc, 1, d_cs, d_ci, nullptr));
    if (cur_uright) {
        FE_TRY(viorb_frontend_search_projection_stereo_device(h, d_ck, d_cd, d_cc, d_ur, d_cs, d_ci, d_pose, d_lpose, d_lk, d_lc, d_lf, d_lp, d_ld, th, bf, mb,
                                                              0, 1, d_m, d_nm, d_st, nullptr));
    } else {
        FE_TRY(viorb_frontend_search_projection_device(h, d_ck, d_cd, d_cc, d_cs, d_ci, d_pose, d_lk, d_lc, d_lf, d_lp, d_ld, th, 1, d_m, d_nm, d_st, nullptr));
    }
    int st = 0;
    B.down(cur_match, d_m, sizeof(int) * ncur); B.down(nmatches, d_nm, sizeof(int)); B.down(&st, d_st, sizeof(int));
    FE_TRY(B.fetch());
    if (st != VIORB_OK) { set_error("SearchByProjection(Frame, Frame): device status %d", st); return st; }
    return VIORB_OK;
}

int viorb_search_by_projection_frame(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, int ncur, const float bounds4[4],
                                     const float pose12[12], const float intr4[4], const float* scale_factors, int nlevels,
                                     const viorb_keypoint* last_kps, int nlast, const uint8_t* last_flags, const float* last_Pw,
                                     const uint8_t* last_desc, float th, int check_orientation, int32_t* cur_match, int* nmatches) {
    return search_by_projection_frame_host(cur_kps, cur_desc, nullptr, ncur, bounds4, pose12, nullptr, intr4, 0.f, 0.f, scale_factors, nlevels, last_kps,
                                           nlast, last_flags, last_Pw, last_desc, th, check_orientation, cur_match, nmatches);
}

int viorb_search_by_projection_frame_stereo(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int ncur,
                                            const float bounds4[4], const float pose12[12], const float last_pose12[12], const float intr4[4],
                                            float bf, float mb, const float* scale_factors, int nlevels, const viorb_keypoint* last_kps, int nlast,
                                            const uint8_t* last_flags, const float* last_Pw, const uint8_t* last_desc, float th,
                                            int check_orientation, int32_t* cur_match, int* nmatches) {
    VIORB_REQUIRE(last_pose12 && (ncur == 0 || cur_uright), "null stereo array");
    return search_by_projection_frame_host(cur_kps, cur_desc, cur_uright, ncur, bounds4, pose12, last_pose12, intr4, bf, mb, scale_factors, nlevels,
                                           last_kps, nlast, last_flags, last_Pw, last_desc, th, check_orientation, cur_match, nmatches);
}

int viorb_fuse(const viorb_keypoint* kps, const uint8_t* desc, const float* uright, int n, const float bounds4[4], const float pose12[12],
               const float intr5[5], const float* scale_factors, const float* inv_level_sigma2, int nlevels, const float* pts_f,
               const uint8_t* pts_valid, const uint8_t* pts_desc, int npts, float th, int32_t* best_idx, int* nfused) {
    VIORB_REQUIRE(bounds4 && pose12 && intr5 && scale_factors && inv_level_sigma2 && nfused && n >= 0 && npts >= 0, "null array");
    VIORB_REQUIRE(nlevels >= 1 && nlevels <= 16, "nlevels must be 1..16");
    VIORB_REQUIRE(npts == 0 || best_idx, "best_idx is NULL");
    *nfused = 0;
    for (int i = 0; i < npts; i++) best_idx[i] = -1;
    if (n == 0 || npts == 0) return VIORB_OK;
    VIORB_REQUIRE(kps && desc && uright && pts_f && pts_valid && pts_desc && best_idx, "null array");
    viorb_frontend_config c = default_cfg();
    c.min_x = bounds4[0]; c.max_x = bounds4[1]; c.min_y = bounds4[2]; c.max_y = bounds4[3];
    c.fx = intr5[0]; c.fy = intr5[1]; c.cx = intr5[2]; c.cy = intr5[3];
    c.nlevels = nlevels;
    for (int i = 0; i < 16; i++) { c.scale_factors[i] = scale_factors[i < nlevels ? i : nlevels - 1]; c.inv_level_sigma2[i] = inv_level_sigma2[i < nlevels ? i : nlevels - 1]; }
    viorb_frontend* h = nullptr;
    FE_TRY(host_frontend(c, n, &h, true));
    const size_t hc = (size_t)h->cap;
    DevBuf B; viorb_keypoint* d_k; uint8_t *d_d, *d_pv, *d_pd; float *d_ur, *d_pose, *d_pf; int *d_c, *d_cs, *d_ci, *d_pc, *d_bi, *d_nf;
    FE_TRY(B.up(&d_k, (const viorb_keypoint*)nullptr, hc)); FE_TRY(B.up(&d_d, (const uint8_t*)nullptr, hc * 32)); FE_TRY(B.up(&d_ur, (const float*)nullptr, hc));
    FE_TRY(B.put(d_k, kps, sizeof(viorb_keypoint) * n)); FE_TRY(B.put(d_d, desc, (size_t)32 * n)); FE_TRY(B.put(d_ur, uright, sizeof(float) * n));
    FE_TRY(B.up(&d_pose, pose12, 12)); FE_TRY(B.up(&d_c, &n, 1)); FE_TRY(B.up(&d_cs, (const int*)nullptr, GRID_CELLS + 1)); FE_TRY(B.up(&d_ci, (const int*)nullptr, hc));
    FE_TRY(B.up(&d_pf, pts_f, (size_t)npts * 8)); FE_TRY(B.up(&d_pv, pts_valid, (size_t)npts)); FE_TRY(B.up(&d_pd, pts_desc, (size_t)npts * 32));
    FE_TRY(B.up(&d_pc, &npts, 1)); FE_TRY(B.up(&d_bi, (const int*)nullptr, (size_t)npts)); FE_TRY(B.up(&d_nf, (const int*)nullptr, 1));
    FE_TRY(B.flush());
    FE_TRY(viorb_frontend_grid_device(h, d_k, d_c, 1, d_cs, d_ci, nullptr));
    FE_TRY(viorb_frontend_fuse_device(h, d_k, d_d, d_ur, d_c, d_cs, d_ci, d_pose, d_pf, d_pv, d_pd, d_pc, npts, th, intr5[4], 1, d_bi, d_nf, nullptr));
    B.down(best_idx, d_bi, sizeof(int) * npts); B.down(nfused, d_nf, sizeof(int));
    FE_TRY(B.fetch());
    return VIORB_OK;
}

static int search_by_projection_points_host(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, float bf, int ncur,
                                            const float bounds4[4], const float pose12[12], const float intr4[4], const float* scale_factors, int nlevels,
                                            const float* pts_f, const uint8_t* pts_flags, const uint8_t* pts_desc, int npts, float th, float nnratio,
                                            const uint8_t* cur_owner_obs, int32_t* match, int* nmatches, float* frustum5, float* proj_xr) {
    VIORB_REQUIRE(bounds4 && pose12 && intr4 && scale_factors && nmatches && ncur >= 0 && npts >= 0, "null array");
    VIORB_REQUIRE(nlevels >= 1 && nlevels <= 16, "nlevels must be 1..16");
    VIORB_REQUIRE(ncur == 0 || match, "match is NULL");
    *nmatches = 0;
    for (int i = 0; i < ncur; i++) match[i] = -1;
    if (frustum5) memset(frustum5, 0, sizeof(float) * 5 * (size_t)npts);
    if (proj_xr) memset(proj_xr, 0, sizeof(float) * (size_t)npts);
    if (ncur == 0 || npts == 0) return VIORB_OK;
    VIORB_REQUIRE(cur_kps && cur_desc && pts_f && pts_flags && pts_desc && cur_owner_obs, "null array");
    viorb_frontend_config c = default_cfg();
    c.min_x = bounds4[0]; c.max_x = bounds4[1]; c.min_y = bounds4[2]; c.max_y = bounds4[3];
    c.fx = intr4[0]; c.fy = intr4[1]; c.cx = intr4[2]; c.cy = intr4[3];
    c.nlevels = nlevels;
    for (int i = 0; i < 16; i++) c.scale_factors[i] = scale_factors[i < nlevels ? i : nlevels - 1];
    viorb_frontend* h = nullptr;
    FE_TRY(host_frontend(c, ncur, &h, true));
    const size_t hc = (size_t)h->cap;
    DevBuf B; viorb_keypoint* d_k; uint8_t *d_d, *d_pfl, *d_pd, *d_own; float *d_pose, *d_pf, *d_fr = nullptr, *d_ur = nullptr, *d_xr = nullptr; int *d_c, *d_cs, *d_ci, *d_pc, *d_m, *d_nm, *d_st;
    FE_TRY(B.up(&d_k, (const viorb_keypoint*)nullptr, hc)); FE_TRY(B.up(&d_d, (const uint8_t*)nullptr, hc * 32)); FE_TRY(B.up(&d_own, (const uint8_t*)nullptr, hc));
    if (cur_uright) FE_TRY(B.up(&d_ur, (const float*)nullptr, hc));
    FE_TRY(B.put(d_k, cur_kps, sizeof(viorb_keypoint) * ncur)); FE_TRY(B.put(d_d, cur_desc, (size_t)32 * ncur)); FE_TRY(B.put(d_own, cur_owner_obs, (size_t)ncur));
    if (cur_uright) FE_TRY(B.put(d_ur, cur_uright, sizeof(float) * ncur));
    FE_TRY(B.up(&d_pose, pose12, 12)); FE_TRY(B.up(&d_c, &ncur, 1)); FE_TRY(B.up(&d_cs, (const int*)nullptr, GRID_CELLS + 1)); FE_TRY(B.up(&d_ci, (const int*)nullptr, hc));
    FE_TRY(B.up(&d_pf, pts_f, (size_t)npts * 8)); FE_TRY(B.up(&d_pfl, pts_flags, (size_t)npts)); FE_TRY(B.up(&d_pd, pts_desc, (size_t)npts * 32));
    FE_TRY(B.up(&d_pc, &npts, 1)); FE_TRY(B.up(&d_m, (const int*)nullptr, hc)); FE_TRY(B.up(&d_nm, (const int*)nullptr, 1)); FE_TRY(B.up(&d_st, (const int*)nullptr, 1));
    if (frustum5) FE_TRY(B.up(&d_fr, (const float*)nullptr, (size_t)npts * 5));
    if (proj_xr) FE_TRY(B.up(&d_xr, (const float*)nullptr, (size_t)npts));
    FE_TRY(B.flush());
    FE_TRY(viorb_frontend_grid_device(h, d_k, d_c, 1, d_cs, d_ci, nullptr));
    if (cur_uright)
        FE_TRY(viorb_frontend_search_local_points_stereo_device(h, d_k, d_d, d_c, d_ur, bf, d_cs, d_ci, d_pose, d_pf, d_pfl, d_pd, d_pc, npts, th, nnratio, d_own, 1, d_m,
                                                                d_nm, d_fr, d_xr, d_st, nullptr));
    else
    FE_TRY(viorb_frontend_search_local_points_device(h, d_k, d_d, d_c, d_cs, d_ci, d_pose, d_pf, d_pfl, d_pd, d_pc, npts, th, nnratio, d_own, 1, d_m, d_nm, d_fr, d_st,
                                                     nullptr));
    int st = 0;
    B.down(match, d_m, sizeof(int) * ncur); B.down(nmatches, d_nm, sizeof(int)); B.down(&st, d_st, sizeof(int));
    if (frustum5) B.down(frustum5, d_fr, sizeof(float) * 5 * (size_t)npts);
    if (proj_xr && d_xr) B.down(proj_xr, d_xr, sizeof(float) * (size_t)npts);
    FE_TRY(B.fetch());
    if (st != VIORB_OK) { set_error("SearchByProjection(Frame, MapPoints): device status %d", st); return st; }
    return VIORB_OK;
}

int viorb_search_by_projection_points(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, int ncur, const float bounds4[4], const float pose12[12],
                                      const float intr4[4], const float* scale_factors, int nlevels, const float* pts_f, const uint8_t* pts_flags,
                                      const uint8_t* pts_desc, int npts, float th, float nnratio, const uint8_t* cur_owner_obs, int32_t* match,
                                      int* nmatches, float* frustum5) {
    return search_by_projection_points_host(cur_kps, cur_desc, nullptr, 0.f, ncur, bounds4, pose12, intr4, scale_factors, nlevels, pts_f, pts_flags, pts_desc, npts, th,
                                            nnratio, cur_owner_obs, match, nmatches, frustum5, nullptr);
}

int viorb_search_by_projection_points_stereo(const viorb_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, float bf, int ncur,
                                             const float bounds4[4], const float pose12[12], const float intr4[4], const float* scale_factors, int nlevels,
                                             const float* pts_f, const uint8_t* pts_flags, const uint8_t* pts_desc, int npts, float th, float nnratio,
                                             const uint8_t* cur_owner_obs, int32_t* match, int* nmatches, float* frustum5, float* proj_xr) {
    VIORB_REQUIRE(ncur == 0 || cur_uright, "cur_uright is NULL");
    return search_by_projection_points_host(cur_kps, cur_desc, cur_uright, bf, ncur, bounds4, pose12, intr4, scale_factors, nlevels, pts_f, pts_flags, pts_desc, npts, th,
                                            nnratio, cur_owner_obs, match, nmatches, frustum5, proj_xr);
}

int viorb_preintegrate(const double* imu, int n_imu, const double bg[3], const double ba[3], double t_last, double t_cur, double* preint142) {
    VIORB_REQUIRE(imu && bg && ba && preint142 && n_imu >= 1, "null array / n_imu < 1");
    viorb_frontend_config c = default_cfg();
    for (int i = 0; i < 9; i += 4) c.cam[4 + i] = 1;
    viorb_frontend* h = nullptr;
    FE_TRY(host_frontend(c, 64, &h));
    double ns[22]; memset(ns, 0, sizeof(ns)); ns[9] = 1; for (int k = 0; k < 3; k++) { ns[10 + k] = bg[k]; ns[13 + k] = ba[k]; }
    DevBuf B; double *d_imu, *d_tl, *d_tc, *d_ns, *d_pre, *d_cur; float* d_pose;
    FE_TRY(B.up(&d_imu, imu, (size_t)n_imu * 7)); FE_TRY(B.up(&d_tl, &t_last, 1)); FE_TRY(B.up(&d_tc, &t_cur, 1));
    FE_TRY(B.up(&d_ns, ns, 22)); FE_TRY(B.up(&d_pre, (const double*)nullptr, 142)); FE_TRY(B.up(&d_cur, (const double*)nullptr, 22));
    FE_TRY(B.up(&d_pose, (const float*)nullptr, 12));
    FE_TRY(B.flush());
    FE_TRY(viorb_frontend_imu_predict_device(h, d_imu, n_imu, d_tl, d_tc, d_ns, 1, d_pre, d_cur, d_pose, nullptr));
    B.down(preint142, d_pre, 142 * sizeof(double));
    FE_TRY(B.fetch());
    return VIORB_OK;
}

int viorb_pose_opt_vi(int variant, int compute_marg, const double cur_ns[22], const double last_ns[22], const double prior_ns[22],
                      const double* marg_cov_inv144, const double preint[142], const double gw[3], const double cam[16],
                      const double* obs_cur, int n_cur, const double* obs_last, int n_last, double out_ns[22], double out_last_ns[22],
                      uint8_t* outlier_cur, uint8_t* outlier_last, double* marg_out144, double info[4]) {
    VIORB_REQUIRE(cur_ns && last_ns && preint && gw && cam && out_ns && info && n_cur >= 0 && n_last >= 0, "null array");
    VIORB_REQUIRE(n_cur == 0 || (obs_cur && outlier_cur), "obs_cur/outlier_cur NULL");
    VIORB_REQUIRE(variant == 0 || (prior_ns && marg_cov_inv144), "the Frame overload needs the prior NavState and its information");
    VIORB_REQUIRE(variant == 0 || n_last == 0 || (obs_last && outlier_last), "the Frame overload with n_last > 0 needs obs_last / outlier_last");
    viorb_frontend_config c = default_cfg();
    for (int i = 0; i < 16; i++) c.cam[i] = cam[i];
    for (int i = 0; i < 3; i++) c.gravity[i] = gw[i];
    const int cap = std::max(std::max(n_cur, n_last), 1);
    viorb_frontend* h = nullptr;
    FE_TRY(host_frontend(c, cap, &h));
    DevBuf B; double *d_cur, *d_last, *d_prior, *d_mci, *d_pre, *d_oc, *d_ol, *d_out, *d_outl, *d_marg, *d_info; int *d_nc, *d_nl; uint8_t *d_fc, *d_fl;
    double zero22[22] = {0}, zero144[144] = {0};
    FE_TRY(B.up(&d_cur, cur_ns, 22)); FE_TRY(B.up(&d_last, last_ns, 22)); FE_TRY(B.up(&d_prior, prior_ns ? prior_ns : zero22, 22));
    FE_TRY(B.up(&d_mci, marg_cov_inv144 ? marg_cov_inv144 : zero144, 144)); FE_TRY(B.up(&d_pre, preint, 142));
    FE_TRY(B.up(&d_oc, obs_cur, (size_t)n_cur * 6)); FE_TRY(B.up(&d_ol, obs_last, (size_t)n_last * 6));
    FE_TRY(B.up(&d_out, (const double*)nullptr, 22)); FE_TRY(B.up(&d_outl, (const double*)nullptr, 22));
    FE_TRY(B.up(&d_marg, (const double*)nullptr, 144)); FE_TRY(B.up(&d_info, (const double*)nullptr, 4));
    FE_TRY(B.up(&d_nc, &n_cur, 1)); FE_TRY(B.up(&d_nl, &n_last, 1));
    FE_TRY(B.up(&d_fc, (const uint8_t*)nullptr, cap)); FE_TRY(B.up(&d_fl, (const uint8_t*)nullptr, cap));
    FE_TRY(B.flush());
    FE_TRY(viorb_frontend_pose_opt_device(h, variant, compute_marg, d_cur, d_last, d_prior, d_mci, d_pre, d_oc, d_nc, d_ol, d_nl, 1,
                                          d_out, d_outl, d_fc, d_fl, d_marg, d_info, nullptr));
    B.down(out_ns, d_out, 22 * sizeof(double)); B.down(out_last_ns, d_outl, 22 * sizeof(double));
    if (n_cur) B.down(outlier_cur, d_fc, n_cur);
    if (n_last && outlier_last && variant) B.down(outlier_last, d_fl, n_last);
    if (compute_marg && marg_out144) B.down(marg_out144, d_marg, 144 * sizeof(double));
    B.down(info, d_info, 4 * sizeof(double));
    FE_TRY(B.fetch());
    return VIORB_OK;
}

int viorb_pose_opt_se3(const float pose12[12], const float intr5[5], const double* obs7, int n, float out_pose12[12], uint8_t* outlier,
                       double info[4]) {
    VIORB_REQUIRE(pose12 && intr5 && out_pose12 && info && n >= 0 && (n == 0 || (obs7 && outlier)), "null array");
    viorb_frontend_config c = default_cfg();
    c.fx = intr5[0]; c.fy = intr5[1]; c.cx = intr5[2]; c.cy = intr5[3];
    const int cap = std::max(n, 1);
    viorb_frontend* h = nullptr;
    FE_TRY(host_frontend(c, cap, &h));
    DevBuf B; float *d_p, *d_o; double *d_obs, *d_info; int* d_n; uint8_t* d_f;
    FE_TRY(B.up(&d_p, pose12, 12)); FE_TRY(B.up(&d_o, (const float*)nullptr, 12)); FE_TRY(B.up(&d_obs, obs7, (size_t)n * 7));
    FE_TRY(B.up(&d_info, (const double*)nullptr, 4)); FE_TRY(B.up(&d_n, &n, 1)); FE_TRY(B.up(&d_f, (const uint8_t*)nullptr, (size_t)cap));
    FE_TRY(B.flush());
    FE_TRY(viorb_frontend_pose_opt_se3_device(h, d_p, d_obs, d_n, (double)intr5[4], 1, d_o, d_f, d_info, nullptr));
    B.down(out_pose12, d_o, 12 * sizeof(float)); if (n) B.down(outlier, d_f, n);
    B.down(info, d_info, 4 * sizeof(double));
    FE_TRY(B.fetch());
    return VIORB_OK;
}

// ---- host-only test hooks: vio_core.h compiled for the host ---------------------------------------
void viorb_debug_pvr_edge(const double* i22, const double* j22, const double* b22, const double* preint142, const double* gw,
                          double* e9, double* J189) {
    pvr_edge(ld_pvr(i22), ld_pvr(j22), ld3(b22 + 16), ld3(b22 + 19), preint142, ld3(gw), e9, J189);
}
void viorb_debug_proj_edge(const double* ns22, const double* cam16, const double* obs6, double* e2, double* J12) {
    const pvr s = ld_pvr(ns22); const cam_t K = ld_cam(cam16);
    proj_edge(K, tr(qmat(s.q)), s.P, ld3(obs6), obs6[3], obs6[4], true, e2, J12, J12 + 6);
}
void viorb_debug_prior_edge(const double* pvr22, const double* bias22, const double* prior22, double* e12, double* J144) {
    prior_edge(ld_pvr(pvr22), ld3(bias22 + 13) + ld3(bias22 + 19), prior22, e12, J144);
}
void viorb_debug_update_ns(const double* ns22, const double* preint142, const double* gw, const double* cam16, double* out22, float* pose12) {
    const pvr r = update_ns(ld_pvr(ns22), ld3(preint142), ld3(preint142 + 3), ldm(preint142 + 6), preint142[141], ld3(gw));
    for (int k = 0; k < 22; k++) out22[k] = ns22[k];
    st_pvr(out22, r);
    pose_from_navstate_f32(r, cam16, pose12);
}
void viorb_debug_preint_step(double* small60, const double* omega, const double* acc, double dt) {
    preint_small M; M.dP = ld3(small60); M.dV = ld3(small60 + 3); M.dR = ldm(small60 + 6); M.JPg = ldm(small60 + 15); M.JPa = ldm(small60 + 24);
    M.JVg = ldm(small60 + 33); M.JVa = ldm(small60 + 42); M.JRg = ldm(small60 + 51); M.dt = 0;
    preint_step(M, ld3(omega), ld3(acc), dt);
    st3(small60, M.dP); st3(small60 + 3, M.dV); stm(small60 + 6, M.dR); stm(small60 + 15, M.JPg); stm(small60 + 24, M.JPa);
    stm(small60 + 33, M.JVg); stm(small60 + 42, M.JVa); stm(small60 + 51, M.JRg);
}

} // extern "C"
