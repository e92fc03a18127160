// viorb_amd/csrc/orb_extractor.hip — MI355X (gfx950) ORB extractor: the HIP replacement for
// ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc). Batched by construction: every kernel
// takes a batch of same-sized images (independent camera streams), all intermediate state stays in
// HBM, nothing returns to the host between stages.
//
// Stage -> kernel map (reference lines in brackets):
//   ComputePyramid [:1107-1132]            k_copy_level0 + k_resize_stream (one launch per level: strips of 32 output dword columns
//                                          streaming down the source rows, no LDS; OpenCV 11-bit fixed-point bilinear; LDS tile
//                                          forms k_resize2 / k_resize for geometries it cannot take)
//   per-cell FAST, two thresholds [:765-830] k_fast_cells: one wavefront per 30-px cell, cell tile in
//                                          LDS, score map in LDS, 3x3 NMS, ordered ballot compaction
//   DistributeOctTree [:481-763]           k_octree: one wavefront per (image, level); stable 4-way
//                                          segment partitions in LDS (see octree_arrays.h)
//   GaussianBlur 7x7 s=2 [:1085-1086]      k_blur: separable, strips streaming down the rows (v_dot4 row sums, v_dot2 column sums over
//                                          a register ring), no LDS; on the handle's second stream beside the quadtree
//   IC_Angle + rBRIEF [:77-147]            k_orient_describe: one wavefront per keypoint, wave-reduced
//                                          integer moments, 4 tests per lane, shuffle-packed bytes
// No 19-px border is stored around the levels: nothing on this path reads it.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <string>
#include <mutex>
#include <algorithm>
#include "viorb_common.h"
#include "orb_math.h"
#include "octree_arrays.h"

namespace viorb {

static thread_local char g_err[512] = "";
char* last_error_buf() { return g_err; }
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}

// ---- kernel profiler (process-wide, single-threaded use) ----------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int slot; };
struct Profiler {
    bool enabled = false;
    std::string only;                 // when not empty: only these kernels (comma-separated) are timed (an event pair costs ~8 us of stream time)
    std::vector<std::string> names;
    std::vector<ProfRec> recs;
    size_t used = 0;
} g_prof;
}
ProfScope::ProfScope(const char* name, hipStream_t s) : idx(-1), st(s) {
    if (!g_prof.enabled || !name) return;
    if (!g_prof.only.empty()) {                  // comma-separated list of kernel names
        const std::string key = "," + g_prof.only + ",", me = std::string(",") + name + ",";
        if (key.find(me) == std::string::npos) return;
    }
    if (g_prof.used >= g_prof.recs.size()) {
        if (g_prof.recs.size() >= 16384) return;
        ProfRec r; r.slot = -1;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        g_prof.recs.push_back(r);
    }
    int slot = -1;
    for (size_t i = 0; i < g_prof.names.size(); i++) if (g_prof.names[i] == name) slot = (int)i;
    if (slot < 0) { g_prof.names.push_back(name); slot = (int)g_prof.names.size() - 1; }
    idx = (int)g_prof.used++;
    g_prof.recs[idx].slot = slot;
    (void)hipEventRecord(g_prof.recs[idx].a, st);
}
bool prof_times_everything() { return g_prof.only.empty(); }
ProfScope::~ProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof.recs[idx].b, st); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the function object of the CURRENT device, so the cache is keyed by
// (device, kernel): a handle on a second device of the same process raises the limit there too. Lock-free fast path for the callers
// inside solver loops: a per-thread memo of the last (device, kernel, bytes) that succeeded.
hipError_t raise_dynamic_lds(const void* kernel, size_t bytes) {
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return rc;
    struct Memo { int dev; const void* k; size_t bytes; };
    static thread_local Memo memo[4] = {{-1, nullptr, 0}, {-1, nullptr, 0}, {-1, nullptr, 0}, {-1, nullptr, 0}};
    for (const Memo& m : memo) if (m.dev == dev && m.k == kernel && bytes <= m.bytes) return hipSuccess;
    static std::mutex mu;
    struct Seen { int dev; const void* k; size_t bytes; };
    static std::vector<Seen> seen;
    std::lock_guard<std::mutex> lk(mu);
    Seen* hit = nullptr;
    for (auto& e : seen) if (e.dev == dev && e.k == kernel) hit = &e;
    if (!hit || bytes > hit->bytes) {
        rc = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (rc != hipSuccess) return rc;
        if (hit) hit->bytes = bytes; else { seen.push_back({dev, kernel, bytes}); hit = &seen.back(); }
    }
    static thread_local int next = 0;
    memo[next] = {dev, kernel, hit->bytes}; next = (next + 1) & 3;
    return hipSuccess;
}

static const int8_t kPatternHost[1024] = {
#include "orb_pattern.inc"
};

enum { MINB = 16, PATCH = 31, HALF_PATCH = 15, MAX_LEVELS = 16 };

// ---------------------------------------------------------------------------------------------
// Geometry shared by host and device
// ---------------------------------------------------------------------------------------------
struct LevelDev {
    int w, h, stride;          // level size in px, row pitch in bytes (multiple of 64)
    uint32_t plane_off;        // byte offset of the level inside one image's plane block
    int cell_base, ncells;     // FAST cells of this level in the cell table
    int quota;                 // mnFeaturesPerLevel
    int oct_w, oct_h;          // maxBorder - minBorder
    int n_ini;                 // quadtree roots
    float hx;                  // root width
    float scale;               // mvScaleFactor[l]
    float kp_size;             // (float)(int)(31 * scale)
    int xtab_off, ytab_off;    // resize tables (level >= 1): index of first entry
    int kp_off;                // offset of this level inside the per-image level-keypoint buffer
    int oct_tier_cap;          // candidates the level's own first quadtree launch holds (0: the common first launch), see launch_all
};
struct CellDesc {              // one FAST cell = sub-image [x0,x0+cw) x [y0,y0+ch) of its level
    int32_t level, x0, y0, cw, ch, shx, shy, pitch;   // pitch = bytes per row of the cell's LDS tile (multiple of 4: cw + alignment slack + one spare dword). 32-bit fields: the wave-uniform descriptor then arrives by scalar loads (16-bit fields took a vector-memory round trip)
    uint32_t src_off;          // byte offset, inside one image's plane block, of the aligned dword holding pixel (x0, y0)
    int32_t stride;            // row pitch of the level (so the kernel needs no second, dependent table look-up)
};

static inline int host_cv_round(double v) { return (int)nearbyint(v); }
static inline int host_cv_floor(double v) { int i = host_cv_round(v); float d = (float)(v - i); return i - (d < 0); }
static inline int host_cv_ceil(double v) { int i = host_cv_round(v); float d = (float)(i - v); return i + (d < 0); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// Kernels
// ---------------------------------------------------------------------------------------------

// XCD-aware placement. Workgroups are dealt round-robin over the 8 XCDs (block b and block b + 8 share one) and every XCD has its
// own 4 MB L2, so with the plain (tile, image) grid neighbouring tiles of an image sat on different L2s and every 128-byte line
// that two tiles share (a 70-byte blur row, the 6-px overlap of FAST cells, a keypoint's windows) was fetched from HBM once per XCD:
// FETCH_SIZE read 2.1x (blur), 1.9x (orientation + descriptors) and 1.45x (FAST) the bytes the kernel needs. With a 1-D grid of
// nper * 8 * ceil(batch / 8) blocks, block L works on item (L / 8) % nper of image ((L / 8) / nper) * 8 + L % 8: all work of image b
// runs on XCD b % 8, in consecutive blocks, and an image's 1.1 MB level set stays in that L2 while its tiles are in flight.
// A speed measure only (the mapping of blocks to XCDs is not a contract); batch < 8 uses the plain order.
struct XcdPlace { int nper, batch, xcd, img0; };            // images img0 .. batch - 1 (a launch may take a sub-range of the batch)
__device__ __forceinline__ bool xcd_place(const XcdPlace P, int& img, int& item) {
    const int L = blockIdx.x;
    if (P.xcd) { const int i = L >> 3, g = i / P.nper; item = i - g * P.nper; img = P.img0 + g * 8 + (L & 7); }
    else { img = L / P.nper; item = L - img * P.nper; img += P.img0; }
    return img < P.batch;
}
static inline XcdPlace make_place(int nper, int batch, int img0 = 0, int img1 = -1) {
    XcdPlace P; P.nper = nper; P.batch = img1 < 0 ? batch : std::min(batch, img1); P.xcd = batch >= 8; P.img0 = img0; return P;
}
static inline unsigned place_blocks(const XcdPlace& P) { const int n = P.batch - P.img0; return (unsigned)P.nper * (unsigned)(P.xcd ? 8 * ((n + 7) / 8) : n); }

// Level 0 = the input image re-pitched into the plane buffer.
__global__ void k_copy_level0(const uint8_t* __restrict__ src, int w, int h, int sstride, size_t spitch,
                              uint8_t* __restrict__ planes, size_t frame_bytes, int dstride, int* __restrict__ status, XcdPlace PL, int gx) {
    int b, item;
    if (!xcd_place(PL, b, item)) return;
    const int y = item / gx, bx = item - y * gx;
    if (item == 0 && threadIdx.x == 0) status[b] = 0;      // per-image status word of this extraction (one launch less than a memset)
    const int x = (bx * blockDim.x + threadIdx.x) * 16;
    if (x >= dstride) return;
    const uint8_t* s = src + (size_t)b * spitch + (size_t)y * sstride;
    uint8_t* d = planes + (size_t)b * frame_bytes + (size_t)y * dstride;
    uint32_t v[4];
    if (x + 16 <= w && ((((uintptr_t)(s + x)) & 15) == 0)) {
        const uint4 q = *reinterpret_cast<const uint4*>(s + x);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int xx = x + 4 * k + j;
                uint32_t px = xx < w ? s[xx] : 0;
                acc |= px << (8 * j);
            }
            v[k] = acc;
        }
    }
    *reinterpret_cast<uint4*>(d + x) = make_uint4(v[0], v[1], v[2], v[3]);
}

// cv::resize(prev, cur, INTER_LINEAR) on 8U, OpenCV 2.4 fixed point (see oracle/cvprim.cpp for the
// derivation): H = S[sx0]*a0 + S[sx1]*a1 (11-bit coefficients), out = (((b0*(H0>>4))>>16) +
// ((b1*(H1>>4))>>16) + 2) >> 2. Tables are built on the host with OpenCV's double/float recipe.
// Block = 64x16 output tile, 256 threads x 4 px; the source rectangle is staged in LDS as dwords.
#define RS_TW 64
#define RS_TH 16
__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ planes, size_t frame_bytes,
                                                const LevelDev* __restrict__ lv, int level,
                                                const int2* __restrict__ xtab_all,
                                                const int2* __restrict__ ytab_all, int lds_pitch_dw,
                                                int lds_rows, XcdPlace PL, int gx) {
    extern __shared__ uint32_t s_tile[];
    int b, item;
    if (!xcd_place(PL, b, item)) return;
    const int tby = item / gx, tbx = item - tby * gx;
    const LevelDev L = lv[level], P = lv[level - 1];
    const int2* xtab = xtab_all + L.xtab_off;
    const int2* ytab = ytab_all + L.ytab_off;
    const uint8_t* src = planes + (size_t)b * frame_bytes + P.plane_off;
    uint8_t* dst = planes + (size_t)b * frame_bytes + L.plane_off;
    const int tx0 = tbx * RS_TW, ty0 = tby * RS_TH;
    const int tx1 = min(tx0 + RS_TW, L.w) - 1, ty1 = min(ty0 + RS_TH, L.h) - 1;
    const int ys0 = ytab[ty0].x & 0xffff, ys1 = ytab[ty1].x >> 16;
    const int xs0 = (xtab[tx0].x & 0xffff) & ~3, xs1 = xtab[tx1].x >> 16;
    const int ndw = ((xs1 - xs0) >> 2) + 1, nrows = ys1 - ys0 + 1;
    // host guarantees ndw <= lds_pitch_dw and nrows <= lds_rows
    {
        // two trips cover the default geometry (scale 1.2: 21 x 21 dwords); both loads are in flight before the first store
        const int total = nrows * lds_pitch_dw;
        uint32_t v[2]; int at[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = threadIdx.x + 256 * k, ic = min(i, total - 1);
            const int r = ic / lds_pitch_dw, c = ic - r * lds_pitch_dw;
            at[k] = (i < total && c < ndw) ? i : -1;
            v[k] = *reinterpret_cast<const uint32_t*>(src + (size_t)(ys0 + r) * P.stride + xs0 + 4 * min(c, ndw - 1));
        }
#pragma unroll
        for (int k = 0; k < 2; k++) if (at[k] >= 0) s_tile[at[k]] = v[k];
        for (int i = threadIdx.x + 512; i < total; i += 256) {
            const int r = i / lds_pitch_dw, c = i - r * lds_pitch_dw;
            if (c < ndw)
                s_tile[i] = *reinterpret_cast<const uint32_t*>(src + (size_t)(ys0 + r) * P.stride + xs0 + 4 * c);
        }
    }
    __syncthreads();
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(s_tile);
    const int r = threadIdx.x >> 4, xg = threadIdx.x & 15;
    const int dy = ty0 + r;
    if (dy >= L.h) return;
    const int2 yt = ytab[dy];
    const uint8_t* S0 = t8 + ((yt.x & 0xffff) - ys0) * (lds_pitch_dw * 4) - xs0;
    const uint8_t* S1 = t8 + ((yt.x >> 16) - ys0) * (lds_pitch_dw * 4) - xs0;
    const int b0 = yt.y & 0xffff, b1 = yt.y >> 16;
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int dx = min(tx0 + xg * 4 + j, L.w - 1);
        const int2 xt = xtab[dx];
        const int sx0 = xt.x & 0xffff, sx1 = xt.x >> 16, a0 = xt.y & 0xffff, a1 = xt.y >> 16;
        const int h0 = S0[sx0] * a0 + S0[sx1] * a1;
        const int h1 = S1[sx0] * a0 + S1[sx1] * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        packed |= (uint32_t)(v & 0xff) << (8 * j);
    }
    const int dxw = tx0 + xg * 4;
    if (dxw < L.stride) *reinterpret_cast<uint32_t*>(dst + (size_t)dy * L.stride + dxw) = packed;
}

// Second form of the same resize, used for every level whose geometry allows it (the default): 64x32 output tile (8 px per thread), the
// tile's slices of the coefficient tables and its source rectangle requested together right after ONE scalar tile-descriptor load (the
// first form needs four dependent round trips before its first pixel: level record, table ends, rectangle, per-pixel table entry), the
// arithmetic fed from LDS only. With copy_dst (level 1 only) the staged rectangle — which is the input image — is also written to the
// level-0 plane, so the separate re-pitching copy (one full read and write of the image, one launch) disappears; tiles overlap by a
// column / row or two and write identical bytes there.
#define RS2_TW 64
#define RS2_TH 32
#define RS2_TRIPS 4               // <= 1024 staged dwords per tile (host-checked)
struct Resize2Args {
    const uint8_t* src; size_t src_pitch; int src_stride;     // per-image source: base + b * src_pitch, rows src_stride apart
    uint8_t* dst; size_t dst_pitch; int dst_stride, dst_w, dst_h;
    uint8_t* copy_dst; size_t copy_pitch; int copy_stride;    // level-0 plane (NULL: no copy)
    const int2* xtab; const int2* ytab; const int4* tiles;    // tiles[t] = {xs0 (multiple of 4), ndw, ys0, nrows}
    int gx, lds_pitch_dw; int* status;
};
__global__ __launch_bounds__(256) void k_resize2(Resize2Args A, XcdPlace PL) {
    extern __shared__ uint32_t s_tile[];
    int b, item;
    if (!xcd_place(PL, b, item)) return;
    const int tby = item / A.gx, tbx = item - tby * A.gx;
    const int4 T = A.tiles[item];
    const int xs0 = T.x, ndw = T.y, ys0 = T.z, nrows = T.w;
    const int tx0 = tbx * RS2_TW, ty0 = tby * RS2_TH;
    int2* s_xt = reinterpret_cast<int2*>(s_tile + A.lds_pitch_dw * 48);      // [64], behind the <= 48 staged rows
    int2* s_yt = s_xt + RS2_TW;                                              // [32]
    const uint8_t* src = A.src + (size_t)b * A.src_pitch;
    if (A.status && item == 0 && threadIdx.x == 0) A.status[b] = 0;          // per-image status word of this extraction
    // every load of the tile is issued before the first store
    int2 tab = make_int2(0, 0);
    if (threadIdx.x < RS2_TW) tab = A.xtab[min(tx0 + (int)threadIdx.x, A.dst_w - 1)];
    else if (threadIdx.x < RS2_TW + RS2_TH) tab = A.ytab[min(ty0 + (int)threadIdx.x - RS2_TW, A.dst_h - 1)];
    const int total = nrows * A.lds_pitch_dw;
    uint32_t v[RS2_TRIPS]; int at[RS2_TRIPS], gr[RS2_TRIPS], gc[RS2_TRIPS];
#pragma unroll
    for (int k = 0; k < RS2_TRIPS; k++) {
        const int i = threadIdx.x + 256 * k, ic = min(i, total - 1);
        const int r = ic / A.lds_pitch_dw, c = ic - r * A.lds_pitch_dw;
        at[k] = (i < total && c < ndw) ? i : -1;
        gr[k] = ys0 + r; gc[k] = xs0 + 4 * min(c, ndw - 1);
        v[k] = *reinterpret_cast<const uint32_t*>(src + (size_t)gr[k] * A.src_stride + gc[k]);
    }
    if (threadIdx.x < RS2_TW) s_xt[threadIdx.x] = tab; else if (threadIdx.x < RS2_TW + RS2_TH) s_yt[threadIdx.x - RS2_TW] = tab;
#pragma unroll
    for (int k = 0; k < RS2_TRIPS; k++) if (at[k] >= 0) s_tile[at[k]] = v[k];
    if (A.copy_dst) {
        uint8_t* cd = A.copy_dst + (size_t)b * A.copy_pitch;
#pragma unroll
        for (int k = 0; k < RS2_TRIPS; k++) if (at[k] >= 0) *reinterpret_cast<uint32_t*>(cd + (size_t)gr[k] * A.copy_stride + gc[k]) = v[k];
    }
    __syncthreads();
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(s_tile);
    const int xg = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    int2 xt[4];
#pragma unroll
    for (int j = 0; j < 4; j++) xt[j] = s_xt[xg * 4 + j];
    uint8_t* dst = A.dst + (size_t)b * A.dst_pitch;
    const int pitch = A.lds_pitch_dw * 4;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int r = r0 + 16 * half, dy = ty0 + r;
        if (dy >= A.dst_h) continue;
        const int2 yt = s_yt[r];
        const uint8_t* S0 = t8 + ((yt.x & 0xffff) - ys0) * pitch - xs0;
        const uint8_t* S1 = t8 + ((yt.x >> 16) - ys0) * pitch - xs0;
        const int b0 = yt.y & 0xffff, b1 = yt.y >> 16;
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int sx0 = xt[j].x & 0xffff, sx1 = xt[j].x >> 16, a0 = xt[j].y & 0xffff, a1 = xt[j].y >> 16;
            const int h0 = S0[sx0] * a0 + S0[sx1] * a1;
            const int h1 = S1[sx0] * a0 + S1[sx1] * a1;
            const int val = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(val & 0xff) << (8 * j);
        }
        const int dxw = tx0 + xg * 4;
        if (dxw < A.dst_stride) *reinterpret_cast<uint32_t*>(dst + (size_t)dy * A.dst_stride + dxw) = packed;
    }
}

typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t k, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, a), __builtin_bit_cast(ushort2v, k), c, false);
}
// Third form of the resize, the default whenever the level's tables allow it (downscaling, every source row is the lower row of at
// most one output row, the four outputs of a dword take their eight source bytes from an 8-byte window): no LDS, no tiles. The resize
// is bound by vector issue like the rest of the extractor, so it is built for instructions per pixel:
//   * a half-wavefront owns 32 output dword columns and streams down the SOURCE rows of a band of output rows (one unaligned 8-byte
//     load per lane and row, RSS_PF rows in flight); the two halves of a wavefront take two bands of the same strip;
//   * per source row the horizontal interpolation of the lane's four outputs is one v_perm (the two source bytes into 16-bit fields, the
//     selector is fixed per column) + one v_dot2_u32_u16 with (a0, a1) each, kept as (sum >> 4) << 9; each source row is interpolated
//     ONCE (the tile forms did it for both source rows of every output row);
//   * an output row is emitted at its lower source row from the previous and the current row sums:
//     ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) as two v_mul_hi_u32_u24 of pre-shifted operands, + 2, >> 2, packed.
//     Which output row a source row emits, and its (b0, b1), is a per-level table (etab) each lane loads once for its band and
//     the row loop reads with v_readlane.
// ~16 vector instructions per output pixel (tile forms: ~44). Same integer recipe as above, bit for bit.
#define RSS_LANES 32
#define RSS_PF 6
#define RSS_MAX_SRC_ROWS 60               // source rows of one band: one etab entry per lane, and the row loop (a multiple of RSS_PF) must stay below lane 64
struct ResizeStreamArgs {
    const uint8_t* planes; size_t frame_bytes;
    uint32_t src_off, dst_off;
    int src_stride, src_h, dst_stride, dst_w, dst_h;
    const int2* xtab; const int2* ytab; const uint2* etab; const int4* items;   // items[i] = {first dword column of the strip, first row of band A, rows per band, 0}
};
struct __attribute__((packed, aligned(1))) rss_u8x8 { uint32_t lo, hi; };
__global__ __launch_bounds__(64) void k_resize_stream(ResizeStreamArgs A, XcdPlace PL) {
    int b, item;
    if (!xcd_place(PL, b, item)) return;
    const int4 t = A.items[item];
    const uint8_t* src = A.planes + (size_t)b * A.frame_bytes + A.src_off;
    uint8_t* dst = const_cast<uint8_t*>(A.planes) + (size_t)b * A.frame_bytes + A.dst_off;
    const int lane = threadIdx.x, hl = lane & (RSS_LANES - 1);
    const bool upper = lane >= RSS_LANES;
    const int Rb = t.z, y0A = t.y, y0B = t.y + Rb;
    const int rowsA = min(Rb, A.dst_h - y0A), rowsB = max(min(Rb, A.dst_h - y0B), 0);
    // source rows of the two bands: first = upper row of the first output row, last = lower row of the last output row
    const int rsA = A.ytab[y0A].x & 0xffff, reA = A.ytab[y0A + rowsA - 1].x >> 16;
    const int rsB = rowsB > 0 ? A.ytab[min(y0B, A.dst_h - 1)].x & 0xffff : 0, reB = rowsB > 0 ? A.ytab[min(y0B + rowsB - 1, A.dst_h - 1)].x >> 16 : -1;
    const int ns = max(reA - rsA, reB - rsB) + 1;                                // rows the loop walks (<= RSS_MAX_SRC_ROWS, host-checked)
    // this lane's four output columns: byte selectors inside the 8-byte window that starts at the first column's left source pixel, weights
    const int d = t.x + hl, sdw = A.dst_stride >> 2;
    uint32_t sel[4], aw[4]; int sxf = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int2 e = A.xtab[min(4 * d + j, A.dst_w - 1)];
        if (j == 0) sxf = e.x & 0xffff;
        sel[j] = (uint32_t)((e.x & 0xffff) - sxf) | ((uint32_t)((e.x >> 16) - sxf) << 16) | 0x0c000c00u;
        aw[j] = (uint32_t)e.y;
    }
    // emit entries of both bands, one source row per lane, packed as band-relative row (7 bits, 127 = none) | b0 << 7 | b1 << 19 | same-row flag << 31
    auto pack_entry = [&](int r0, int r1, int yb0, int nrows) {
        const int r = r0 + lane;
        const uint2 e = A.etab[min(r, A.src_h - 1)];
        const int y = (int)(e.x & 0x7fff) - yb0;
        const bool ok = r <= r1 && (e.x & 0x7fff) != 0x7fff && y >= 0 && y < nrows;
        return ok ? ((uint32_t)y | ((e.y & 0xfffu) << 7) | (((e.y >> 16) & 0xfffu) << 19) | ((e.x >> 15) << 31)) : 127u;
    };
    const uint32_t EA = pack_entry(rsA, reA, y0A, rowsA), EB = pack_entry(rsB, reB, y0B, rowsB);
    const uint8_t* pc = src + sxf;
    uint8_t* pd = dst + 4u * (uint32_t)min(d, sdw - 1);
    const bool col_ok = d < sdw;
    // loads are unconditional (row index clamped) and the loop runs to a multiple of RSS_PF: see k_blur
    uint32_t lo[RSS_PF], hi[RSS_PF];
    auto request = [&](int i, uint32_t& l, uint32_t& h) {
        const uint32_t offA = __builtin_amdgcn_readfirstlane(min(rsA + i, A.src_h - 1) * A.src_stride), offB = __builtin_amdgcn_readfirstlane(min(rsB + i, A.src_h - 1) * A.src_stride);
        const rss_u8x8 v = *reinterpret_cast<const rss_u8x8*>(pc + (upper ? offB : offA));
        l = v.lo; h = v.hi;
    };
#pragma unroll
    for (int u = 0; u < RSS_PF; u++) request(u, lo[u], hi[u]);
    uint32_t hp[4] = {0, 0, 0, 0};
    for (int i0 = 0; i0 < ns; i0 += RSS_PF) {
#pragma unroll
        for (int u = 0; u < RSS_PF; u++) {
            const int i = i0 + u;
            uint32_t hc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t two = __builtin_amdgcn_perm(hi[u], lo[u], sel[j]);             // (left source byte, right source byte) as 16-bit fields
                hc[j] = (dot2_u16(two, aw[j], 0u) & 0x7fff0u) << 5;                         // ((S[sx0] * a0 + S[sx1] * a1) >> 4) << 9
            }
            request(i + RSS_PF, lo[u], hi[u]);
            // this row's emit entries of both bands (scalar); everything that differs between the halves is one select of two scalars
            const uint32_t eA = __builtin_amdgcn_readlane(EA, i & 63), eB = __builtin_amdgcn_readlane(EB, i & 63);
            const uint32_t yrA = eA & 127u, yrB = eB & 127u;
            const bool emit = upper ? yrB < (uint32_t)rowsB : yrA < (uint32_t)rowsA;
            if (emit && col_ok) {
                const uint32_t b0 = upper ? (eB & (0xfffu << 7)) : (eA & (0xfffu << 7)), b1 = upper ? ((eB >> 12) & (0xfffu << 7)) : ((eA >> 12) & (0xfffu << 7));
                const uint32_t yoA = __builtin_amdgcn_readfirstlane((y0A + (int)yrA) * A.dst_stride), yoB = __builtin_amdgcn_readfirstlane((y0B + (int)yrB) * A.dst_stride);
                const uint32_t yo = upper ? yoB : yoA;
                uint32_t h0[4] = {hp[0], hp[1], hp[2], hp[3]};
                if ((int)(eA | eB) < 0) {                                                   // bottom clamp (last output rows only): both source rows are this one
                    const bool same = upper ? (int)eB < 0 : (int)eA < 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) h0[j] = same ? hc[j] : hp[j];
                }
                uint32_t v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t t0 = (uint32_t)(((unsigned long long)h0[j] * b0) >> 32), t1 = (uint32_t)(((unsigned long long)hc[j] * b1) >> 32);
                    v[j] = (t0 + t1 + 2u) >> 2;
                }
                const uint32_t w01 = v[0] | (v[1] << 8), w23 = v[2] | (v[3] << 8);
                *reinterpret_cast<uint32_t*>(pd + yo) = w01 | (w23 << 16);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) hp[j] = hc[j];
        }
    }
}

// FAST-9-16 corner strength of the pixel at p (byte pointer into an LDS tile with `pitch` bytes per
// row): max over the 16 arcs of 9 contiguous ring pixels of min(v - ring) and of min(ring - v),
// minus 1 == cv::cornerScore<16>; the pixel is a FAST corner at threshold t iff strength >= t.
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ uint32_t pk_max16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) - __builtin_bit_cast(short2v, b));
}
__device__ __forceinline__ uint32_t swap16(uint32_t a) { return __builtin_amdgcn_alignbit(a, a, 16); }
// The ring differences d[k] = v - ring[k] fit 16 bits, and the 16 arcs come in opposite pairs: register k holds (d[k], d[k+8]), so
// every min / max of the arc recurrence is one packed 16-bit instruction for two arcs (integer min/max issue at quarter rate on
// CDNA4, which is what bounds this kernel), with a half-swap wherever an index wraps past 15.
__device__ __forceinline__ int fast_strength(const uint8_t* p, int pitch) {
    const uint32_t v = p[0], vv = v | (v << 16);
    uint32_t P[8];
    P[0] = pk_sub16(vv, (uint32_t)p[3 * pitch] | ((uint32_t)p[-3 * pitch] << 16));
    P[1] = pk_sub16(vv, (uint32_t)p[3 * pitch + 1] | ((uint32_t)p[-3 * pitch - 1] << 16));
    P[2] = pk_sub16(vv, (uint32_t)p[2 * pitch + 2] | ((uint32_t)p[-2 * pitch - 2] << 16));
    P[3] = pk_sub16(vv, (uint32_t)p[pitch + 3] | ((uint32_t)p[-pitch - 3] << 16));
    P[4] = pk_sub16(vv, (uint32_t)p[3] | ((uint32_t)p[-3] << 16));
    P[5] = pk_sub16(vv, (uint32_t)p[-pitch + 3] | ((uint32_t)p[pitch - 3] << 16));
    P[6] = pk_sub16(vv, (uint32_t)p[-2 * pitch + 2] | ((uint32_t)p[2 * pitch - 2] << 16));
    P[7] = pk_sub16(vv, (uint32_t)p[-3 * pitch + 1] | ((uint32_t)p[3 * pitch - 1] << 16));
    uint32_t S[8], lo2[8], hi2[8], lo4[8], hi4[8];
#pragma unroll
    for (int k = 0; k < 8; k++) S[k] = swap16(P[k]);                       // (d[k+8], d[k])
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint32_t nx = k < 7 ? P[k + 1] : S[0]; lo2[k] = pk_min16(P[k], nx); hi2[k] = pk_max16(P[k], nx); }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t nl = k < 6 ? lo2[k + 2] : swap16(lo2[k - 6]), nh = k < 6 ? hi2[k + 2] : swap16(hi2[k - 6]);
        lo4[k] = pk_min16(lo2[k], nl); hi4[k] = pk_max16(hi2[k], nh);
    }
    uint32_t A = 0x80008000u, B = 0x7fff7fffu;                               // (-32768, -32768), (32767, 32767)
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t nl = k < 4 ? lo4[k + 4] : swap16(lo4[k - 4]), nh = k < 4 ? hi4[k + 4] : swap16(hi4[k - 4]);
        A = pk_max16(A, pk_min16(pk_min16(lo4[k], nl), S[k]));
        B = pk_min16(B, pk_max16(pk_max16(hi4[k], nh), S[k]));
    }
    const int a = max((int)(short)(A & 0xffff), (int)(short)(A >> 16)), bq = min((int)(short)(B & 0xffff), (int)(short)(B >> 16));
    return max(a, -bq) - 1;
}

// One wavefront per FAST cell (reference :789-828): load the (<= wCell+6)x(hCell+6) cell image into
// LDS, then three ordered passes, each compacting its survivors with wave ballots so the next pass runs
// on dense lanes and every list stays in row-major (cv::FAST output) order:
//   1. necessary test on the 4 compass ring pixels (any 9-arc holds two ADJACENT compass pixels, so a
//      corner at threshold t needs an adjacent compass pair both darker than v-t or both brighter than v+t);
//   2. exact corner strength (cv::cornerScore<16>) of the survivors -> score map in LDS, list of corners;
//   3. cv::FAST's strict 3x3 NMS at iniThFAST over the corner list and, only if that leaves the cell
//      empty, again at minThFAST. Survivors go to the cell's slot as packed (x | y<<12 | score<<24) with
//      the reference's j*wCell / i*hCell shift applied.
// Inclusive prefix sum over the 64 lanes with DPP row shifts and row broadcasts (the gfx9 sequence LLVM's atomic optimiser emits):
// four shifted adds scan each row of 16, row_bcast:15 carries row 0 / 2 into row 1 / 3, row_bcast:31 carries the lower half into the upper.
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}
__constant__ uint32_t c_rcp16[64];            // ceil(65536 / d), d = 1 .. 63 (c_rcp16[0] unused)
#define FAST_FETCH_TRIPS 7
#define FAST_LAUNCHES 8
#define FAST_CELLS_PER_WAVE 2         // measured per 256 images: 1: 0.578 ms, 2: 0.553, 4: 0.565, 8: 0.582
__global__ __launch_bounds__(64) void k_fast_cells(const uint8_t* __restrict__ planes, size_t frame_bytes,
                                                   const LevelDev* __restrict__ lv,
                                                   const CellDesc* __restrict__ cells, int ini_th, int min_th,
                                                   uint32_t* __restrict__ slots, int slot_cap,
                                                   int* __restrict__ cell_cnt, int ncells_total,
                                                   int tile_bytes, int score_bytes, int list_cap, XcdPlace PL) {
    extern __shared__ uint32_t s_mem[];
    int img_b, item;
    if (!xcd_place(PL, img_b, item)) return;
    uint8_t* tile = reinterpret_cast<uint8_t*>(s_mem);
    uint8_t* sc = tile + tile_bytes;
    uint16_t* surv = reinterpret_cast<uint16_t*>(sc + score_bytes);     // [list_cap] (r << 8 | q) of pass-1 survivors
    uint16_t* corn = surv + list_cap;                                    // [list_cap] corners (strength >= minTh)
    const int lane = threadIdx.x;
    // A wave takes FAST_CELLS_PER_WAVE consecutive cells. The tile of the next cell is requested (into registers) as soon as the current
    // one has been handed to LDS, so its L2 round trip runs under the current cell's three passes instead of in front of them.
    const int cell0 = item * FAST_CELLS_PER_WAVE;
    const int ncell = min((int)FAST_CELLS_PER_WAVE, ncells_total - cell0);
    uint32_t v[FAST_FETCH_TRIPS];
    const uint8_t* img = planes + (size_t)img_b * frame_bytes;
    // Dword lane + 64 k of the cell's tile image (cd.pitch / 4 dwords per row), unconditionally: columns right of the cell's last dword are real
    // pixels of the same row (a cell ends >= 16 px before the level's right edge), rows below the cell are clamped to its last row, and the LDS
    // tile region holds all 64 * FAST_FETCH_TRIPS dwords (host), so neither the loads nor the stores need a per-dword validity test. The pitch
    // is the cell's own (the widest cells of a pyramid, on its small levels, would cost the 30-px cells of the large levels a fifth more dwords
    // and an eighth trip); lane / pitch_dw by a 16-bit reciprocal (exact for lane < 64).
    auto request_tile = [&](const CellDesc& cd) {
        const int pdw = cd.pitch >> 2;
        const int rcp = (int)c_rcp16[min(pdw, 63)];                      // ceil(65536 / pdw): a scalar table load instead of an integer division
        int r = __mul24(lane, rcp) >> 16, q = lane - __mul24(r, pdw);          // 24-bit multiplies: full rate (v_mul_lo_u32 is quarter rate)
        const int sr = (64 * rcp) >> 16, sq = 64 - sr * pdw;
        const uint8_t* src = img + cd.src_off;
#pragma unroll
        for (int k = 0; k < FAST_FETCH_TRIPS; k++) {
            v[k] = *reinterpret_cast<const uint32_t*>(src + (uint32_t)(__mul24(min(r, cd.ch - 1), cd.stride) + 4 * q));
            r += sr; q += sq;
            if (q >= pdw) { q -= pdw; r++; }
        }
    };
    CellDesc c = cells[cell0];
    request_tile(c);
    for (int kk = 0; kk < ncell; kk++) {
    const int cell = cell0 + kk;
    const int tile_pitch = c.pitch, pitch_dw = tile_pitch >> 2;
    const int x0a = c.x0 & ~3, xoff = c.x0 - x0a;
    const int ndw = ((c.x0 + c.cw - 1 - x0a) >> 2) + 1;
    {
#pragma unroll
        for (int k = 0; k < FAST_FETCH_TRIPS; k++) s_mem[lane + 64 * k] = v[k];
        const int ntile = c.ch * pitch_dw;
        if (ntile > 64 * FAST_FETCH_TRIPS) {                                      // larger tiles than the default geometry
            const uint8_t* src = img + c.src_off;
            for (int i = lane + 64 * FAST_FETCH_TRIPS; i < ntile; i += 64) {
                const int r = i / pitch_dw, q = i - r * pitch_dw;
                if (q < ndw) s_mem[i] = *reinterpret_cast<const uint32_t*>(src + r * c.stride + 4 * q);
            }
        }
    }
    const bool has_next = kk + 1 < ncell;
    const CellDesc cn = cells[has_next ? cell + 1 : cell];
    const int dw = c.cw - 6, dh = c.ch - 6;          // interior (detection) region
    const int sp = dw + 2;                            // score map pitch, 1-px zero frame
    for (int i = lane; i < (score_bytes >> 2); i += 64) reinterpret_cast<uint32_t*>(sc)[i] = 0;
    __syncthreads();
    if (has_next) request_tile(cn);
    uint32_t* my_slots = slots + ((size_t)img_b * ncells_total + cell) * slot_cap;
    int total = 0;
    if (dw > 0 && dh > 0) {
        const int npx = dw * dh;
        const unsigned long long lt = (1ull << lane) - 1ull;
        // The reference detects at iniThFAST and only when a cell comes back empty again at minThFAST (:797-807). Same here: the pre-test,
        // the exact strengths and the NMS run at iniThFAST first — the weak threshold lets three to four times as many pixels through
        // the pre-test — and the whole cell is redone at minThFAST only if nothing survived (the score map keeps what it already has).
        for (int attempt = 0; attempt < 2 && total == 0; attempt++) {
            const int th = attempt ? min_th : ini_th;
            // ---- pass 1
            int nsurv = 0;
            {
                // A lane takes a PAIR of adjacent aligned dwords = 8 horizontally adjacent pixels of a row (lanes run over (row, pair) in
                // row-major order, so lane order x byte order is cv::FAST's keypoint order): eight dword LDS reads (two centres, left,
                // right, two three rows up, two three rows down) instead of forty byte reads, and the address / stepping / prefix-sum /
                // store overhead is paid once per eight pixels (it was ~45 % of a four-pixel trip). The kernel is VALU-issue bound, so
                // instructions per pixel is the lever.
                const int g0 = (xoff + 3) >> 2;                                  // tile dword holding the first interior pixel
                const int G = ((xoff + 3 + dw - 1) >> 2) - g0 + 1;               // dwords per interior row
                const int G2 = (G + 1) >> 1;                                     // dword pairs per interior row
                const int rcpG = (int)c_rcp16[min(G2, 63)];                      // ceil(65536 / G2); n / G2 = (n * rcpG) >> 16 exactly for n <= 64
                const int step_r = (64 * rcpG) >> 16, step_g = 64 - step_r * G2;
                int r = __mul24(lane, rcpG) >> 16, gp = lane - __mul24(r, G2);
                const int ntrip = (dh * G2 + 63) >> 6;
                const uint32_t M = 0x00ff00ffu, Hb = 0x80008000u;
                const uint32_t K1 = Hb + (uint32_t)th * 0x00010001u, K2 = Hb - (uint32_t)(th + 1) * 0x00010001u;
                const int sh0 = (xoff + 3) - 4 * g0;                             // first interior pixel's byte in its dword
                const int hi_last = min((G & 1) ? 4 : 8, dw - (8 * (G2 - 1) - sh0));
                const uint32_t mask_first = (0xffu << sh0) & 0xffu, mask_last = (1u << hi_last) - 1u;
                for (int trip = 0; trip < ntrip; trip++) {
                    const int rc = min(r, dh - 1);                                // lanes past the last row read a valid address and are masked
                    const int ga = 2 * gp, gb = min(ga + 1, G - 1);               // second dword of the last (odd) pair: clamped, masked below
                    const uint32_t* row = reinterpret_cast<const uint32_t*>(tile + __mul24(rc + 3, tile_pitch)) + g0;
                    const uint32_t* rowU = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(row) - 3 * tile_pitch);
                    const uint32_t* rowD = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(row) + 3 * tile_pitch);
                    const uint32_t Lf = row[ga - 1], C0 = row[ga], C1 = row[gb], Rt = row[gb + 1];
                    const uint32_t U0 = rowU[ga], U1 = rowU[gb], D0 = rowD[ga], D1 = rowD[gb];
                    // pixels 3 to the left / right of each dword's four (the second dword's left neighbour is the first, unless it was clamped)
                    const uint32_t Cl1 = gb == ga ? Lf : C0;
                    const uint32_t W0 = __builtin_amdgcn_alignbyte(C0, Lf, 1), E0 = __builtin_amdgcn_alignbyte(gb == ga ? Rt : C1, C0, 3);
                    const uint32_t W1 = __builtin_amdgcn_alignbyte(C1, Cl1, 1), E1 = __builtin_amdgcn_alignbyte(Rt, C1, 3);
                    const int q0 = 4 * (g0 + ga) - (xoff + 3);                   // interior column of byte 0 (may be < 0 in the first dword)
                    // Four pixels at once per dword, bytes widened to 16-bit fields (even bytes in one register, odd bytes in another): with
                    // c1 = 0x8000 + t - v and c2 = 0x8000 - t - 1 - v per field, bit 15 of (p + c1) says "p is NOT darker than v - t" and
                    // bit 15 of (p + c2) says "p is brighter than v + t"; no field can carry into its neighbour. Some adjacent pair of the four
                    // compass pixels (0, 4, 8, 12) is dark-dark iff (dark0 | dark8) & (dark4 | dark12), same for bright: plain and/or/add
                    // at full rate instead of four extract + sixteen min/max per pixel.
                    // The four (dword, half) results carry pixel k's verdict in bit 15 (field 0) and bit 31 (field 1); shift + and-or gathers them
                    // into X = bits {0, 1, 4, 5} (fields 0 of the even / odd bytes of dword 0 / 1) and {16, 17, 20, 21} (fields 1), one fold
                    // puts pixel k on bit k.
                    uint32_t X = 0;
    #pragma unroll
                    for (int dwi = 0; dwi < 2; dwi++) {
                        const uint32_t C = dwi ? C1 : C0, U = dwi ? U1 : U0, D = dwi ? D1 : D0, W = dwi ? W1 : W0, E = dwi ? E1 : E0;
    #pragma unroll
                        for (int half = 0; half < 2; half++) {
                            // even bytes: one and; odd bytes: one v_perm_b32 (bytes 1 and 3 into the low byte of each 16-bit field)
                            auto field = [&](uint32_t x) { return half ? __builtin_amdgcn_perm(x, x, 0x0c030c01u) : (x & M); };
                            const uint32_t v2 = field(C), c1 = K1 - v2, c2 = K2 - v2;
                            const uint32_t p0 = field(D), p4 = field(E), p8 = field(U), p12 = field(W);
                            const uint32_t nd = ((p0 + c1) & (p8 + c1)) | ((p4 + c1) & (p12 + c1));      // bit 15: no adjacent dark pair
                            const uint32_t br = ((p0 + c2) | (p8 + c2)) & ((p4 + c2) | (p12 + c2));      // bit 15: an adjacent bright pair
                            const uint32_t cand = ~nd | br;
                            X |= (cand >> (15 - half - 4 * dwi)) & (0x00010001u << (half + 4 * dwi));
                        }
                    }
                    uint32_t bits = (X | (X >> 14)) & 0xffu;                             // fields 1 are pixels 2, 3 (6, 7) of their dword
                    // columns outside the interior: only the first pair of a row (pixels left of it) and the last (right of it, and the clamped
                    // second dword of an odd row end) have any; their masks are per-cell constants
                    bits &= (gp == 0 ? mask_first : 0xffu) & (gp == G2 - 1 ? mask_last : 0xffu);
                    if (r >= dh) bits = 0;
                    // ordered compaction: position = survivors in lower lanes + survivors in lower bytes of this lane, from a DPP prefix sum
                    // of the per-lane counts
                    const int cnt = __popc(bits), incl = wave_inclusive_scan(cnt);
                    const int trip_total = __builtin_amdgcn_readlane(incl, 63);
                    if (trip_total) {                                            // wave-uniform: nothing to store on a flat stretch
                        int pos = nsurv + incl - cnt;
                        const uint32_t rq0 = ((uint32_t)r << 8) + (uint32_t)q0;
                        uint32_t bb = bits;
                        while (bb) {                                             // survivors are sparse: one or two trips for the whole wave
                            const int k = __builtin_ctz(bb);
                            surv[pos++] = (uint16_t)(rq0 + k);
                            bb &= bb - 1;
                        }
                        nsurv += trip_total;
                    }
                    r += step_r; gp += step_g;
                    if (gp >= G2) { gp -= G2; r++; }
                }
            }
            __syncthreads();
            // ---- pass 2
            int ncorn = 0;
            for (int base = 0; base < nsurv; base += 64) {
                bool is_c = false;
                uint16_t rq = 0;
                if (base + lane < nsurv) {
                    rq = surv[base + lane];
                    const int r = rq >> 8, q = rq & 0xff;
                    const int s = fast_strength(tile + __mul24(r + 3, tile_pitch) + xoff + q + 3, tile_pitch);
                    if (s >= th) { is_c = true; sc[__mul24(r + 1, sp) + q + 1] = (uint8_t)s; }
                }
                const unsigned long long m = __ballot(is_c);
                if (is_c) corn[ncorn + __popcll(m & lt)] = rq;
                ncorn += __popcll(m);
            }
            __syncthreads();
            // ---- pass 3
            {
                for (int base = 0; base < ncorn; base += 64) {
                    bool keep = false;
                    int r = 0, q = 0, s = 0;
                    if (base + lane < ncorn) {
                        const uint16_t rq = corn[base + lane];
                        r = rq >> 8; q = rq & 0xff;
                        const uint8_t* z = sc + __mul24(r + 1, sp) + q + 1;
                        s = z[0];
                        // all eight neighbours are read before any is tested: one LDS round trip instead of a short-circuit chain of eight
                        int nb[8];
                        nb[0] = z[-sp - 1]; nb[1] = z[-sp]; nb[2] = z[-sp + 1]; nb[3] = z[-1]; nb[4] = z[1]; nb[5] = z[sp - 1]; nb[6] = z[sp]; nb[7] = z[sp + 1];
                        int mx = 0;
    #pragma unroll
                        for (int k = 0; k < 8; k++) mx = max(mx, nb[k]);           // the map only holds corners at the current threshold
                        keep = s > mx;
                    }
                    const unsigned long long m = __ballot(keep);
                    if (keep) {
                        const int pos = total + __popcll(m & lt);
                        if (pos < slot_cap)
                            my_slots[pos] = (uint32_t)(q + 3 + c.shx) | ((uint32_t)(r + 3 + c.shy) << 12) | ((uint32_t)s << 24);
                    }
                    total += __popcll(m);
                }
            }
            __syncthreads();
        }
    }
    if (lane == 0) cell_cnt[(size_t)img_b * ncells_total + cell] = min(total, slot_cap);
    __syncthreads();                                  // every LDS read of this cell is done before the next tile is written
    c = cn;
    }
}

// ---------------------------------------------------------------------------------------------
// k_fast_cells3 (round 3): the same three passes with a third fewer vector instructions per cell. The kernel is bound by vector issue
// (SQ_INSTS_VALU per cell, profiles/r03_*), so every change removes instructions, none touches a result:
//   * compile-time LDS geometry (tile pitch 52 B, score-map pitch 44 B): every row / neighbour offset of the three passes is an
//     instruction immediate. Cells wider than 13 dwords or taller than 44 rows take the generic kernel above (host check).
//   * tile fetch as (4 rows x 16 dwords) per trip: lane -> (lane / 16, lane % 16) once per cell, a trip's four rows are the same per-lane
//     offset from a wave-uniform base (scalar adds), LDS stores at immediate offsets — 12 vector instructions per cell instead of ~70.
//   * pre-test by packed min / max: "some adjacent pair of the compass pixels (0, 4, 8, 12) is darker than v - t" is
//     max(min(p0, p8), min(p4, p12)) < v - t, brighter: min(max(p0, p8), max(p4, p12)) > v + t — 6 packed min / max + 2 adds for two pixels
//     instead of 8 adds + 6 logic operations.
//   * polarity-split strength: the pre-test says WHICH polarity can be a corner; a survivor entry carries it, and pass 2 evaluates only
//     that polarity of cv::cornerScore — max over the 16 arcs of the min of s (v - ring), s = +1 dark / -1 bright — half the min / max
//     tree. Exact: a pixel whose bright pre-test failed at t has no bright arc at t, so its bright score is below t and cannot be the
//     maximum of a corner's score nor make it a corner; a pixel cannot be a corner in both polarities (two 9-arcs do not fit in 16).
//     A pixel passing both pre-tests gets two consecutive entries (8 % of the survivors on the bench images).
// ---------------------------------------------------------------------------------------------
#define F3_TPD 13                     // LDS tile pitch, dwords
#define F3_TP (4 * F3_TPD)            // ... bytes
#define F3_TRIPS 11                   // 4 rows per trip: tiles of up to 44 rows
#define F3_SP 44                      // score-map pitch, bytes (interior width + 2 <= 44)
#define F3_SROWS 40                   // score-map rows (interior height + 2 <= 40)
__device__ __forceinline__ uint32_t pk_mul16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) * __builtin_bit_cast(short2v, b));
}
// cv::cornerScore<16> of ONE polarity: sgn = 0x00010001 (dark: d = v - ring) or 0xffffffff (bright: d = ring - v); returns
// max over the 16 arcs of 9 of min(d) - 1. p points at the centre pixel inside the LDS tile (pitch F3_TP).
__device__ __forceinline__ int fast_strength_pol(const uint8_t* p, uint32_t sgn) {
    const uint32_t v = p[0], vv = v | (v << 16);
    uint32_t P[8];
    P[0] = pk_sub16(vv, (uint32_t)p[3 * F3_TP] | ((uint32_t)p[-3 * F3_TP] << 16));
    P[1] = pk_sub16(vv, (uint32_t)p[3 * F3_TP + 1] | ((uint32_t)p[-3 * F3_TP - 1] << 16));
    P[2] = pk_sub16(vv, (uint32_t)p[2 * F3_TP + 2] | ((uint32_t)p[-2 * F3_TP - 2] << 16));
    P[3] = pk_sub16(vv, (uint32_t)p[F3_TP + 3] | ((uint32_t)p[-F3_TP - 3] << 16));
    P[4] = pk_sub16(vv, (uint32_t)p[3] | ((uint32_t)p[-3] << 16));
    P[5] = pk_sub16(vv, (uint32_t)p[-F3_TP + 3] | ((uint32_t)p[F3_TP - 3] << 16));
    P[6] = pk_sub16(vv, (uint32_t)p[-2 * F3_TP + 2] | ((uint32_t)p[2 * F3_TP - 2] << 16));
    P[7] = pk_sub16(vv, (uint32_t)p[-3 * F3_TP + 1] | ((uint32_t)p[3 * F3_TP - 1] << 16));
    uint32_t S[8], lo2[8], lo4[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { P[k] = pk_mul16(P[k], sgn); S[k] = swap16(P[k]); }          // (d[k], d[k+8]) and (d[k+8], d[k])
#pragma unroll
    for (int k = 0; k < 8; k++) lo2[k] = pk_min16(P[k], k < 7 ? P[k + 1] : S[0]);
#pragma unroll
    for (int k = 0; k < 8; k++) lo4[k] = pk_min16(lo2[k], k < 6 ? lo2[k + 2] : swap16(lo2[k - 6]));
    uint32_t A = 0x80008000u;
#pragma unroll
    for (int k = 0; k < 8; k++) A = pk_max16(A, pk_min16(pk_min16(lo4[k], k < 4 ? lo4[k + 4] : swap16(lo4[k - 4])), S[k]));
    return max((int)(short)(A & 0xffff), (int)(short)(A >> 16)) - 1;
}
// bit k of an 8-bit mask -> bit 2k
__host__ __device__ __forceinline__ uint32_t spread8(uint32_t x) {
    x = (x | (x << 4)) & 0x0f0fu; x = (x | (x << 2)) & 0x3333u; x = (x | (x << 1)) & 0x5555u;
    return x;
}
// LDS per wavefront (the kernel is latency-bound at the occupancy its LDS footprint allows — r03 counters: vector pipes 33 % busy, LDS 27 %,
// ~11 resident waves per CU at 11.4 KB each — so the footprint is what is tuned): tile | score map | survivor buffer | corner list.
//   * survivors are consumed as they come: after every pass-1 trip the full groups of 64 go through pass 2 and the remainder (< 64) moves to
//     the front, so the buffer holds at most 63 + 64 * 16 entries however many pixels survive;
//   * the corner list holds F3_CORN_CAP entries; a cell with more corners than that (more than a third of its pixels) takes a dense NMS
//     pass over the score map instead — same output, no list.
#define F3_SURV_CAP (63 + 64 * 16 + 1)
#define F3_CORN_CAP 512
// Phase cycle counts for development (-DVIORB_FAST_TIMING): the kernel is bound by vector issue, so a phase's share of the s_memtime ticks is its
// share of the instructions
#ifdef VIORB_FAST_TIMING
#define FT_DECL unsigned long long ft_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ft_t0 = __builtin_amdgcn_s_memtime(); int ft_cells = 0, ft_surv = 0, ft_corn = 0, ft_att2 = 0
#define FT_LAP(k) do { const unsigned long long ft_now = __builtin_amdgcn_s_memtime(); ft_acc[k] += ft_now - ft_t0; ft_t0 = ft_now; } while (0)
#else
#define FT_DECL
#define FT_LAP(k)
#endif
__global__ __launch_bounds__(64) void k_fast_cells3(const uint8_t* __restrict__ planes, size_t frame_bytes,
                                                    const CellDesc* __restrict__ cells, int ini_th, int min_th,
                                                    uint32_t* __restrict__ slots, int slot_cap,
                                                    int* __restrict__ cell_cnt, int ncells_total, int tile_bytes, int score_bytes, XcdPlace PL) {
    extern __shared__ uint32_t s_mem[];
    int img_b, item;
    if (!xcd_place(PL, img_b, item)) return;
    uint8_t* tile = reinterpret_cast<uint8_t*>(s_mem);                   // [rows][F3_TP]
    uint8_t* sc = tile + tile_bytes;                                      // [rows][F3_SP]
    uint16_t* surv = reinterpret_cast<uint16_t*>(sc + score_bytes);      // [F3_SURV_CAP] (r << 7 | q << 1 | polarity) of pass-1 survivors
    uint16_t* corn = surv + F3_SURV_CAP;                                 // [F3_CORN_CAP] corners (r << 7 | q << 1)
    const int lane = threadIdx.x;
    const int cell0 = item * FAST_CELLS_PER_WAVE;
    const int ncell = min((int)FAST_CELLS_PER_WAVE, ncells_total - cell0);
    const uint8_t* img = planes + (size_t)img_b * frame_bytes;
    // tile fetch: trip k holds rows 4k .. 4k + 3, 16 dwords of each (lanes whose dword lies beyond the tile pitch stay idle)
    const int lrow = lane >> 4, lq = lane & 15;
    const bool fetch_lane = lq < F3_TPD;
    uint32_t v[F3_TRIPS];
    auto request_tile = [&](const CellDesc& cd) {
        const uint8_t* src = img + cd.src_off;
        const int ntrip = (cd.ch + 3) >> 2;
        const uint32_t voff = (uint32_t)(__mul24(lrow, cd.stride) + 4 * lq);
        // the last trip may reach below the cell: its rows are clamped to the cell's last row (the tile rows past ch are never read)
        const uint32_t voff_last = (uint32_t)(__mul24(min(4 * (ntrip - 1) + lrow, cd.ch - 1) - 4 * (ntrip - 1), cd.stride) + 4 * lq);
        if (fetch_lane) {
#pragma unroll
            for (int k = 0; k < F3_TRIPS; k++)
                if (k < ntrip) {                                          // wave-uniform
                    const uint8_t* base = src + (size_t)(4 * k) * (size_t)cd.stride;
                    v[k] = *reinterpret_cast<const uint32_t*>(base + (k == ntrip - 1 ? voff_last : voff));
                }
        }
    };
    CellDesc c = cells[cell0];
    request_tile(c);
    const int lds_lane = __mul24(lrow, F3_TP) + 4 * lq;
    const unsigned long long lt = (1ull << lane) - 1ull;
    FT_DECL;
    for (int kk = 0; kk < ncell; kk++) {
    FT_LAP(7);
    const int cell = cell0 + kk;
    const int x0a = c.x0 & ~3, xoff = c.x0 - x0a;
    {
        const int ntrip = (c.ch + 3) >> 2;
        if (fetch_lane) {
#pragma unroll
            for (int k = 0; k < F3_TRIPS; k++)
                if (k < ntrip) *reinterpret_cast<uint32_t*>(tile + lds_lane + 4 * k * F3_TP) = v[k];
        }
    }
    const bool has_next = kk + 1 < ncell;
    const CellDesc cn = cells[has_next ? cell + 1 : cell];
    const int dw = c.cw - 6, dh = c.ch - 6;          // interior (detection) region
    {   // score map: rows 0 .. dh + 1, 16 bytes per lane and trip
        const int nq = ((dh + 2) * F3_SP + 15) >> 4;
        for (int i = lane; i < nq; i += 64) reinterpret_cast<uint4*>(sc)[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if (has_next) request_tile(cn);
    FT_LAP(0);
    uint32_t* my_slots = slots + ((size_t)img_b * ncells_total + cell) * slot_cap;
    int total = 0;
    if (dw > 0 && dh > 0) {
        const uint8_t* tile_c = tile + 3 * F3_TP + xoff + 3;             // interior pixel (0, 0)
        for (int attempt = 0; attempt < 2 && total == 0; attempt++) {
            const int th = attempt ? min_th : ini_th;
            // ---- pass 1 (pre-test on 8 pixels per lane, see k_fast_cells; verdicts as an interleaved 16-bit mask, bit 2k = pixel k can be a
            // dark corner, bit 2k + 1 = a bright one) with pass 2 (the strength of each entry's polarity; corners (>= th) to the score map
            // and, in order, to the corner list) run on every full group of 64 survivors as soon as it exists
            int pending = 0, ncorn = 0;
            {
                const int g0 = (xoff + 3) >> 2;
                const int G = ((xoff + 3 + dw - 1) >> 2) - g0 + 1;
                const int G2 = (G + 1) >> 1;
                const int rcpG = (int)c_rcp16[min(G2, 63)];
                const int step_r = (64 * rcpG) >> 16, step_g = 64 - step_r * G2;
                int r = __mul24(lane, rcpG) >> 16, gp = lane - __mul24(r, G2);
                const int ntrip = (dh * G2 + 63) >> 6;
                const uint32_t M = 0x00ff00ffu, Hb = 0x80008000u;
                const uint32_t K1 = Hb - (uint32_t)(th + 1) * 0x00010001u;       // + v - A: bit 15 set <=> A < v - t  (an adjacent dark pair)
                const uint32_t K2 = Hb - (uint32_t)(th + 1) * 0x00010001u;       // - v + B: bit 15 set <=> B > v + t  (an adjacent bright pair)
                const int sh0 = (xoff + 3) - 4 * g0;
                const int hi_last = min((G & 1) ? 4 : 8, dw - (8 * (G2 - 1) - sh0));
                const uint32_t m8_first = (0xffu << sh0) & 0xffu, m8_last = (1u << hi_last) - 1u;
                const uint32_t mask_first = spread8(m8_first) * 3u, mask_last = spread8(m8_last) * 3u;
                FT_LAP(1);
#ifdef VIORB_FAST_TIMING
                if (attempt) ft_att2++;
#endif
                for (int trip = 0; trip < ntrip; trip++) {
                    const int rc = min(r, dh - 1);
                    const int ga = 2 * gp, gb = min(ga + 1, G - 1);
                    const uint32_t* rowU = reinterpret_cast<const uint32_t*>(tile + __mul24(rc, F3_TP)) + g0;      // ring row 3 above the centre row
                    const uint32_t Lf = rowU[3 * F3_TPD + ga - 1], C0 = rowU[3 * F3_TPD + ga], C1 = rowU[3 * F3_TPD + gb], Rt = rowU[3 * F3_TPD + gb + 1];
                    const uint32_t U0 = rowU[ga], U1 = rowU[gb], D0 = rowU[6 * F3_TPD + ga], D1 = rowU[6 * F3_TPD + gb];
                    const uint32_t Cl1 = gb == ga ? Lf : C0;
                    const uint32_t W0 = __builtin_amdgcn_alignbyte(C0, Lf, 1), E0 = __builtin_amdgcn_alignbyte(gb == ga ? Rt : C1, C0, 3);
                    const uint32_t W1 = __builtin_amdgcn_alignbyte(C1, Cl1, 1), E1 = __builtin_amdgcn_alignbyte(Rt, C1, 3);
                    const int q0 = 4 * (g0 + ga) - (xoff + 3);
                    uint32_t X = 0;
    #pragma unroll
                    for (int dwi = 0; dwi < 2; dwi++) {
                        const uint32_t C = dwi ? C1 : C0, U = dwi ? U1 : U0, D = dwi ? D1 : D0, W = dwi ? W1 : W0, E = dwi ? E1 : E0;
    #pragma unroll
                        for (int half = 0; half < 2; half++) {
                            auto field = [&](uint32_t x) { return half ? __builtin_amdgcn_perm(x, x, 0x0c030c01u) : (x & M); };
                            const uint32_t v2 = field(C);
                            const uint32_t p0 = field(D), p4 = field(E), p8 = field(U), p12 = field(W);
                            const uint32_t A = pk_max16(pk_min16(p0, p8), pk_min16(p4, p12));
                            const uint32_t B = pk_min16(pk_max16(p0, p8), pk_max16(p4, p12));
                            const uint32_t dk = (K1 + v2) - A, br = (K2 - v2) + B;                 // bits 15 / 31: the verdicts of the two pixels
                            // pixel 4 dwi + 2 field + half -> bits 8 dwi + 4 field + 2 half (dark) and + 1 (bright); field 1 arrives 16 bits up, folded below
                            X |= (dk >> (15 - 8 * dwi - 2 * half)) & (0x00010001u << (8 * dwi + 2 * half));
                            X |= (br >> (14 - 8 * dwi - 2 * half)) & (0x00010001u << (8 * dwi + 2 * half + 1));
                        }
                    }
                    uint32_t bits = (X | (X >> 12)) & 0xffffu;
                    bits &= (gp == 0 ? mask_first : 0xffffu) & (gp == G2 - 1 ? mask_last : 0xffffu);
                    if (r >= dh) bits = 0;
                    const int cnt = __popc(bits), incl = wave_inclusive_scan(cnt);
                    const int trip_total = __builtin_amdgcn_readlane(incl, 63);
                    if (trip_total) {
                        int pos = pending + incl - cnt;
                        const uint32_t rq0 = ((uint32_t)r << 7) + (uint32_t)(2 * q0);
                        uint32_t bb = bits;
                        while (bb) {
                            const int k = __builtin_ctz(bb);
                            surv[pos++] = (uint16_t)(rq0 + k);
                            bb &= bb - 1;
                        }
                        pending += trip_total;
                    }
                    r += step_r; gp += step_g;
                    if (gp >= G2) { gp -= G2; r++; }
                    // ---- pass 2 on the full groups (after the last trip: on whatever is left)
                    const int need = trip + 1 < ntrip ? 64 : 1;
                    FT_LAP(2);
#ifdef VIORB_FAST_TIMING
                    ft_surv += trip_total;
#endif
                    if (pending >= need) {
                        __syncthreads();
                        int done = 0;
                        while (pending - done >= need) {
                            const int n = min(64, pending - done);
                            bool is_c = false;
                            uint32_t e = 0;
                            if (lane < n) {
                                e = surv[done + lane];
                                const int er = e >> 7, eq = (e >> 1) & 63;
                                const int s = fast_strength_pol(tile_c + __mul24(er, F3_TP) + eq, (e & 1) ? 0xffffffffu : 0x00010001u);
                                if (s >= th) { is_c = true; sc[__mul24(er + 1, F3_SP) + eq + 1] = (uint8_t)s; }
                            }
                            const unsigned long long m = __ballot(is_c);
                            if (is_c) { const int cp = ncorn + __popcll(m & lt); if (cp < F3_CORN_CAP) corn[cp] = (uint16_t)(e & ~1u); }
                            ncorn += __popcll(m);
                            done += n;
                        }
                        const int left = pending - done;                          // < 64: to the front of the buffer
                        if (left > 0) {
                            const uint16_t t = lane < left ? surv[done + lane] : (uint16_t)0;
                            __syncthreads();
                            if (lane < left) surv[lane] = t;
                        }
                        pending = left;
                        __syncthreads();
                        FT_LAP(3);
                    }
                }
            }
            __syncthreads();
            FT_LAP(4);
#ifdef VIORB_FAST_TIMING
            ft_corn += ncorn;
#endif
            // ---- pass 3: strict 3x3 NMS, over the corner list — or, for a cell with more corners than the list holds, over every pixel of the
            // score map in the same (row-major) order
            const bool dense = ncorn > F3_CORN_CAP;
            const int n3 = dense ? dh * dw : ncorn;
            for (int base = 0; base < n3; base += 64) {
                bool keep = false;
                int r = 0, q = 0, s = 0;
                if (base + lane < n3) {
                    if (dense) { r = (base + lane) / dw; q = (base + lane) - r * dw; }
                    else { const uint32_t e = corn[base + lane]; r = e >> 7; q = (e >> 1) & 63; }
                    const uint8_t* z = sc + __mul24(r + 1, F3_SP) + q + 1;
                    s = z[0];
                    int nb[8];
                    nb[0] = z[-F3_SP - 1]; nb[1] = z[-F3_SP]; nb[2] = z[-F3_SP + 1]; nb[3] = z[-1]; nb[4] = z[1]; nb[5] = z[F3_SP - 1]; nb[6] = z[F3_SP]; nb[7] = z[F3_SP + 1];
                    int mx = 0;
    #pragma unroll
                    for (int k = 0; k < 8; k++) mx = max(mx, nb[k]);
                    keep = s > mx;                                                 // a pixel that is no corner has s == 0
                }
                const unsigned long long m = __ballot(keep);
                if (keep) {
                    const int pos = total + __popcll(m & lt);
                    if (pos < slot_cap)
                        my_slots[pos] = (uint32_t)(q + 3 + c.shx) | ((uint32_t)(r + 3 + c.shy) << 12) | ((uint32_t)s << 24);
                }
                total += __popcll(m);
            }
            __syncthreads();
            FT_LAP(5);
        }
    }
    if (lane == 0) cell_cnt[(size_t)img_b * ncells_total + cell] = min(total, slot_cap);
    __syncthreads();
    c = cn;
    FT_LAP(6);
#ifdef VIORB_FAST_TIMING
    ft_cells++;
#endif
    }
#ifdef VIORB_FAST_TIMING
    if (lane == 0 && (blockIdx.x % 997) == 3)
        printf("fast3 block=%d cells=%d att2=%d surv=%d corn=%d | tile+clear=%llu setup=%llu pass1=%llu pass2=%llu tail=%llu nms=%llu end=%llu loop=%llu\n", (int)blockIdx.x, ft_cells, ft_att2, ft_surv, ft_corn,
               ft_acc[0], ft_acc[1], ft_acc[2], ft_acc[3], ft_acc[4], ft_acc[5], ft_acc[6], ft_acc[7]);
#endif
}

// ---------------------------------------------------------------------------------------------
// Quadtree distribution: one wavefront per (level, image). Mirrors distribute_octree_arrays()
// (octree_arrays.h) step for step; every control decision is wave-uniform.
// ---------------------------------------------------------------------------------------------
// single-wavefront workgroups: a workgroup barrier is one s_barrier and also orders LDS traffic
#define WAVE_SYNC() __syncthreads()
#define OCT_NCAP_SMALL 4096
// Two instantiations of the same code: BIG = false keeps candidates (keys + two u16 permutation buffers) in LDS — up to oct_ncap of
// them — and BIG = true takes a level with MORE candidates than that (an image of noise: the reference has no limit, src/ORBextractor.cc:779
// is only a reserve) with u32 permutations in a global scratch slot, 32-bit node ranges and 64-bit sort keys; nodes and the sort buffer
// stay in LDS either way. Same decisions, same output.
struct OctNodeBig { int16_t x0, x1, y0, y1; uint32_t begin, count; uint16_t flags, pad; };
template <bool BIG> struct OctT;
template <> struct OctT<false> { typedef uint16_t Perm; typedef OctNode Node; typedef uint32_t SortKey; enum { KEY_SHIFT = 16 }; };
template <> struct OctT<true> { typedef uint32_t Perm; typedef OctNodeBig Node; typedef unsigned long long SortKey; enum { KEY_SHIFT = 32 }; };
template <bool BIG> struct OctLdsT {
    uint32_t* keys; typename OctT<BIG>::Perm* perm0; typename OctT<BIG>::Perm* perm1; typename OctT<BIG>::Node* nd; typename OctT<BIG>::SortKey* sortb;
};

__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// DivideNode. Every expansion owns FOUR consecutive node slots [slot, slot + 4): child c goes to slot + c, an empty child is written
// as a tombstone, and the stable compaction that closes a round removes tombstones and dead parents alike — so the surviving
// order is exactly the reference's "push the non-empty children to the list front in n1..n4 order". Fixed slots make expansions
// independent of each other: a round's small nodes are expanded one per LANE (64 at a time), only big nodes take the whole wave.
#define OCT_LANE_LIMIT 64                     // points a lane-local expansion handles (child codes live in two 64-bit registers)
struct OctExpandResult { int off, nexp; };    // non-empty children, children with more than one point

template <bool BIG> __device__ __forceinline__ void oct_write_children(const OctLdsT<BIG>& S, const typename OctT<BIG>::Node& p, int i, int slot, int mx, int my, int sb,
                                                   const int tc[4], const int sc4[4]) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        typename OctT<BIG>::Node o;
        o.x0 = (c & 1) ? (int16_t)mx : p.x0; o.x1 = (c & 1) ? p.x1 : (int16_t)mx;
        o.y0 = (c & 2) ? (int16_t)my : p.y0; o.y1 = (c & 2) ? p.y1 : (int16_t)my;
        o.begin = (decltype(o.begin))sc4[c]; o.count = (decltype(o.count))tc[c];
        o.flags = tc[c] > 0 ? (uint16_t)((sb ^ 1) | (tc[c] == 1 ? OCT_NOMORE : 0)) : (uint16_t)OCT_DEAD; o.pad = 0;
        S.nd[slot + c] = o;
    }
    S.nd[i].flags = p.flags | OCT_DEAD;
}

// whole wave on one node: stable 4-way partition of its key segment with ballots. A single wavefront per tree has nothing to hide
// LDS latency behind but its own instruction stream, so four 64-point chunks are in flight at a time: their four index loads, then
// their four key loads, then the sixteen ballots. Nodes of <= 256 points (all but the first rounds) keep indices and child codes
// in registers between the counting and the scatter pass.
#define OCT_CODE(key) (((int)((key) & 0xfff) < mx ? 0 : 1) | ((int)(((key) >> 12) & 0xfff) < my ? 0 : 2))
template <bool BIG> __device__ __forceinline__ OctExpandResult oct_expand_wave(const OctLdsT<BIG>& S, int i, int slot, int lane) {
    typedef typename OctT<BIG>::Perm Perm;
    const typename OctT<BIG>::Node p = S.nd[i];
    const int mx = p.x0 + ((p.x1 - p.x0 + 1) >> 1), my = p.y0 + ((p.y1 - p.y0 + 1) >> 1);
    const int sb = p.flags & OCT_BUF;
    const Perm* src = (sb ? S.perm1 : S.perm0) + (int)p.begin;
    Perm* dst = sb ? S.perm0 : S.perm1;
    const int cnt = (int)p.count, beg = (int)p.begin;
    const unsigned long long lt = lanemask_lt(lane);
    int t[4] = {0, 0, 0, 0};
    Perm id[4]; int c[4];
    auto load4 = [&](int o) {                                    // chunks o, o + 64, o + 128, o + 192 (a chunk past the end costs one clamped load)
#pragma unroll
        for (int u = 0; u < 4; u++) id[u] = src[min(o + 64 * u + lane, cnt - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t key = S.keys[id[u]]; c[u] = o + 64 * u + lane < cnt ? OCT_CODE(key) : -1; }
    };
    int w0, w1, w2, w3;                                          // running write positions of the four children
    auto scatter4 = [&]() {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned long long m0 = __ballot(c[u] == 0), m1 = __ballot(c[u] == 1), m2 = __ballot(c[u] == 2), m3 = __ballot(c[u] == 3);
            const unsigned long long mm = c[u] == 0 ? m0 : (c[u] == 1 ? m1 : (c[u] == 2 ? m2 : m3));
            int at = w3;
            at = c[u] == 2 ? w2 : at; at = c[u] == 1 ? w1 : at; at = c[u] == 0 ? w0 : at;
            if (c[u] >= 0) dst[at + __popcll(mm & lt)] = id[u];
            w0 += __popcll(m0); w1 += __popcll(m1); w2 += __popcll(m2); w3 += __popcll(m3);
        }
    };
    auto count4 = [&]() {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            t[0] += __popcll(__ballot(c[u] == 0)); t[1] += __popcll(__ballot(c[u] == 1));
            t[2] += __popcll(__ballot(c[u] == 2)); t[3] += __popcll(__ballot(c[u] == 3));
        }
    };
    if (cnt <= 256) {
        load4(0); count4();
        w0 = beg; w1 = w0 + t[0]; w2 = w1 + t[1]; w3 = w2 + t[2];
        scatter4();
    } else {
        for (int o = 0; o < cnt; o += 256) { load4(o); count4(); }
        w0 = beg; w1 = w0 + t[0]; w2 = w1 + t[1]; w3 = w2 + t[2];
        for (int o = 0; o < cnt; o += 256) { load4(o); scatter4(); }
    }
    const int sc4[4] = {beg, beg + t[0], beg + t[0] + t[1], beg + t[0] + t[1] + t[2]};
    if (lane == 0) oct_write_children(S, p, i, slot, mx, my, sb, t, sc4);
    OctExpandResult r; r.off = (t[0] > 0) + (t[1] > 0) + (t[2] > 0) + (t[3] > 0); r.nexp = (t[0] > 1) + (t[1] > 1) + (t[2] > 1) + (t[3] > 1);
    return r;
}

// one lane on one node (count <= OCT_LANE_LIMIT): counting pass (child codes kept in registers), then the stable scatter; four
// points in flight per trip for the same reason as above
template <bool BIG> __device__ __forceinline__ void oct_lane_count(const OctLdsT<BIG>& S, const typename OctT<BIG>::Node& p, int mx, int my, unsigned long long& lo, unsigned long long& hi, int tc[4]) {
    const typename OctT<BIG>::Perm* src = ((p.flags & OCT_BUF) ? S.perm1 : S.perm0) + (int)p.begin;
    lo = 0; hi = 0; tc[0] = tc[1] = tc[2] = tc[3] = 0;
    const int cnt = (int)p.count;
    for (int k = 0; k < cnt; k += 4) {
        typename OctT<BIG>::Perm id[4]; uint32_t key[4];
#pragma unroll
        for (int u = 0; u < 4; u++) id[u] = src[min(k + u, cnt - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) key[u] = S.keys[id[u]];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool in = k + u < cnt;
            const int cx = (int)(key[u] & 0xfff) < mx ? 0 : 1, cy = (int)((key[u] >> 12) & 0xfff) < my ? 0 : 1;
            lo |= (unsigned long long)(in ? cx : 0) << ((k + u) & 63); hi |= (unsigned long long)(in ? cy : 0) << ((k + u) & 63);
            const int c = in ? (cx | (cy << 1)) : -1;
            tc[0] += c == 0; tc[1] += c == 1; tc[2] += c == 2; tc[3] += c == 3;
        }
    }
}
template <bool BIG> __device__ __forceinline__ void oct_lane_scatter(const OctLdsT<BIG>& S, const typename OctT<BIG>::Node& p, int i, int slot, int mx, int my, unsigned long long lo,
                                                 unsigned long long hi, const int tc[4]) {
    const int sb = p.flags & OCT_BUF;
    const typename OctT<BIG>::Perm* src = (sb ? S.perm1 : S.perm0) + (int)p.begin;
    typename OctT<BIG>::Perm* dst = sb ? S.perm0 : S.perm1;
    const int sc4[4] = {(int)p.begin, (int)p.begin + tc[0], (int)p.begin + tc[0] + tc[1], (int)p.begin + tc[0] + tc[1] + tc[2]};
    int w0 = sc4[0], w1 = sc4[1], w2 = sc4[2], w3 = sc4[3];
    const int cnt = (int)p.count;
    for (int k = 0; k < cnt; k += 4) {
        typename OctT<BIG>::Perm id[4];
#pragma unroll
        for (int u = 0; u < 4; u++) id[u] = src[min(k + u, cnt - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (k + u < cnt) {
                const int c = (int)((lo >> (k + u)) & 1) | ((int)((hi >> (k + u)) & 1) << 1);
                const int pos = c == 0 ? w0 : (c == 1 ? w1 : (c == 2 ? w2 : w3));
                dst[pos] = id[u];
                w0 += c == 0; w1 += c == 1; w2 += c == 2; w3 += c == 3;
            }
        }
    }
    oct_write_children(S, p, i, slot, mx, my, sb, tc, sc4);
}

// Stable in-place removal of dead nodes; returns the new node count; first_new = new index of `upto`.
template <bool BIG> __device__ __forceinline__ int oct_compact(const OctLdsT<BIG>& S, int nn, int upto, int lane, int& first_new) {
    int w = 0;
    first_new = -1;
    for (int base = 0; base < nn; base += 64) {
        const int i = base + lane;
        typedef typename OctT<BIG>::Node Node;
        enum { NW = sizeof(Node) / 4, FW = offsetof(Node, flags) / 4, FS = 8 * (offsetof(Node, flags) % 4) };
        uint32_t wd[NW];                                         // the node as words: a struct copy across the barrier went through scratch
        const uint32_t* from = reinterpret_cast<const uint32_t*>(S.nd + min(i, nn - 1));
#pragma unroll
        for (int k = 0; k < NW; k++) wd[k] = from[k];
        const bool alive = (i < nn) && !((wd[FW] >> FS) & OCT_DEAD);
        const unsigned long long m = __ballot(alive);
        if (upto >= base && upto < base + 64) first_new = w + __popcll(m & lanemask_lt(upto - base));
        WAVE_SYNC();
        if (alive) {
            uint32_t* to = reinterpret_cast<uint32_t*>(S.nd + w + __popcll(m & lanemask_lt(lane)));
#pragma unroll
            for (int k = 0; k < NW; k++) to[k] = wd[k];
        }
        w += __popcll(m);
        WAVE_SYNC();
    }
    if (first_new < 0) first_new = w;
    return w;
}

// Ascending bitonic sort of m (power of two) u32 keys in LDS by one wavefront.
template <class K> __device__ __forceinline__ void oct_sort(K* a, int m, int lane) {
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (m >> 1); t += 64) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const K x = a[lo], y = a[hi];
                if ((x > y) == up) { a[lo] = y; a[hi] = x; }
            }
            WAVE_SYNC();
        }
}

template <bool BIG>
__global__ __launch_bounds__(64) void k_octree(const LevelDev* __restrict__ lv, const uint32_t* __restrict__ slots,
                                               int slot_cap, const int* __restrict__ cell_cnt, int ncells_total,
                                               uint32_t* __restrict__ lvl_kp, int kp_pitch, int* __restrict__ lvl_cnt,
                                               int* __restrict__ lvl_ncand, int nlevels, int* __restrict__ status,
                                               int ncap, int nodecap, int sortcap, int n_above, int n_upto,
                                               int level0, int use_tier_cap, int defer_nodes,
                                               uint32_t* __restrict__ big_scratch, int* __restrict__ big_next, int big_slots, int* __restrict__ lvl_tot,
                                               uint8_t* __restrict__ node_scratch = nullptr, size_t node_bytes = 0) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_oct[];
    typedef typename OctT<BIG>::Perm Perm; typedef typename OctT<BIG>::Node Node; typedef typename OctT<BIG>::SortKey SortKey;
    OctLdsT<BIG> S;
    if (BIG) {                                                          // nodes + sort keys in LDS; keys / permutations in a scratch slot claimed below
        S.sortb = reinterpret_cast<SortKey*>(s_oct);
        S.nd = reinterpret_cast<Node*>(S.sortb + sortcap);
        S.keys = nullptr; S.perm0 = S.perm1 = nullptr;
    } else {
        S.keys = s_oct;
        S.perm0 = reinterpret_cast<Perm*>(s_oct + ncap);
        S.perm1 = S.perm0 + ncap;
        S.nd = reinterpret_cast<Node*>(S.perm1 + ncap);
        S.sortb = reinterpret_cast<SortKey*>(S.nd + nodecap);
    }
    // grid = (image, level): with the level in x and 8 levels, the round-robin deal of workgroups to the 8 XCDs would send every
    // level-0 quadtree (the longest) to XCD 0 and every level-7 one to XCD 7; image-major order spreads each level over all XCDs
    // and starts the long ones first
    // A launch may cover the levels from level0 on only (gridDim.y of them): the higher levels have their own first launches with smaller LDS
    // plans (fewer candidates, smaller quota: more workgroups per CU), and the full-capacity launch takes what exceeded a level's plan there
    // (use_tier_cap: count above the level's oct_tier_cap instead of n_above)
    const int lane = threadIdx.x, level = blockIdx.y + level0, b = blockIdx.x;
    const LevelDev L = lv[level];
    const int N = L.quota;
    if (use_tier_cap && L.oct_tier_cap > 0) n_above = L.oct_tier_cap;
    // ---- 0. this launch only takes the (image, level) pairs with n_above < candidates <= n_upto: the common case runs
    //         with a smaller LDS footprint (3 workgroups per CU), a second launch with the full capacity takes the rest
    const int* cc = cell_cnt + (size_t)b * ncells_total + L.cell_base;
    {
        int tot = 0;
        if (n_above < 0) {                                               // the first launch counts the level's candidates, the later ones read the count
            for (int c0 = 0; c0 < L.ncells; c0 += 512) {                 // eight independent loads in flight per lane
                int part[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { const int ci = c0 + 64 * j + lane; part[j] = cc[min(ci, L.ncells - 1)]; if (ci >= L.ncells) part[j] = 0; }
#pragma unroll
                for (int j = 0; j < 8; j++) tot += part[j];
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
            if (lane == 0) lvl_tot[b * nlevels + level] = tot;
        } else tot = lvl_tot[b * nlevels + level];
        // bit 30: a first launch ran out of NODE slots on this level (its node list is sized for the common case, defer_nodes) and left it to the
        // full-capacity launch, whatever its candidate count
        const bool deferred = (tot & 0x40000000) != 0;
        tot &= 0x3fffffff;
        if (!(deferred && !BIG && !defer_nodes) && (tot <= n_above || tot > n_upto)) return;
        if (BIG) {
            // one scratch slot of 3 * ncap words per over-size (image, level); when more levels than slots overflow in one call the rest
            // report VIORB_ERR_CAPACITY (big_slots = min(batch * levels, 64): every level of a small batch of pure-noise images fits)
            int slot = 0;
            if (lane == 0) slot = atomicAdd(big_next, 1);
            slot = __shfl(slot, 0);
            if (slot >= big_slots) { if (lane == 0) { lvl_cnt[b * nlevels + level] = 0; lvl_ncand[b * nlevels + level] = tot; status[b] = VIORB_ERR_CAPACITY; } return; }
            uint32_t* base = big_scratch + (size_t)slot * 3 * (size_t)ncap;
            S.keys = base; S.perm0 = reinterpret_cast<Perm*>(base + ncap); S.perm1 = reinterpret_cast<Perm*>(base + 2 * (size_t)ncap);
            if (node_scratch) {                                          // a quota whose node list does not fit LDS: nodes + sort keys in the slot as well
                S.sortb = reinterpret_cast<SortKey*>(node_scratch + (size_t)slot * node_bytes);
                S.nd = reinterpret_cast<Node*>(S.sortb + sortcap);
            }
        }
    }
    // ---- 1. gather candidates in the reference's push order (cell-major, row-major inside a cell): one cell per lane, eight of its
    //         slots requested before the first is stored (the slot loop used to be one L2 round trip per slot)
    int n = 0;
    bool overflow = false;
    {
        const uint32_t* sl = slots + ((size_t)b * ncells_total + L.cell_base) * slot_cap;
        for (int base = 0; base < L.ncells; base += 64) {
            const int ci = base + lane;
            const int cnt = ci < L.ncells ? cc[ci] : 0;
            // inclusive wave scan of cnt
            const int incl = wave_inclusive_scan(cnt);
            const int start = n + incl - cnt;
            const uint32_t* mine = sl + (size_t)min(ci, L.ncells - 1) * slot_cap;
            for (int k0 = 0; __any(k0 < cnt); k0 += 8) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = mine[min(k0 + u, slot_cap - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int pos = start + k0 + u;
                    if (k0 + u < cnt && pos < ncap) S.keys[pos] = v[u];
                }
            }
            n += __shfl(incl, 63);
        }
        if (n > ncap) { overflow = true; n = ncap; }
    }
    if (lane == 0) lvl_ncand[b * nlevels + level] = n;
    WAVE_SYNC();
    uint32_t* out = lvl_kp + (size_t)b * kp_pitch + L.kp_off;
    if (n == 0 || L.n_ini < 1 || N < 1) {
        if (lane == 0) { lvl_cnt[b * nlevels + level] = 0; if (overflow) status[b] = VIORB_ERR_CAPACITY; }
        return;
    }
    // ---- 2. roots: stable counting sort by root index; array holds the roots back-to-front
    int nn = 0, live = 0;
    {
        int begin = 0;
        for (int r = 0; r < L.n_ini; r++) {
            int cnt = 0;
            for (int o = 0; o < n; o += 64) {
                const int idx = o + lane;
                bool mine = false;
                if (idx < n) mine = ((int)((float)(S.keys[idx] & 0xfff) / L.hx)) == r;
                const unsigned long long m = __ballot(mine);
                if (mine) S.perm0[begin + cnt + __popcll(m & lanemask_lt(lane))] = (Perm)idx;
                cnt += __popcll(m);
            }
            if (cnt > 0) {
                if (lane == 0) {
                    Node o;
                    o.x0 = (int16_t)(int)(L.hx * (float)r); o.x1 = (int16_t)(int)(L.hx * (float)(r + 1));
                    o.y0 = 0; o.y1 = (int16_t)L.oct_h;
                    o.begin = (decltype(o.begin))begin; o.count = (decltype(o.count))cnt;
                    o.flags = (uint16_t)(cnt == 1 ? OCT_NOMORE : 0); o.pad = 0;
                    S.nd[nn] = o;
                }
                nn++;
            }
            begin += cnt;
        }
        WAVE_SYNC();
        // reverse the root array (list front = root 0 must be the LAST array element)
        for (int i = lane; i < (nn >> 1); i += 64) {
            const Node a = S.nd[i], z = S.nd[nn - 1 - i];
            S.nd[i] = z; S.nd[nn - 1 - i] = a;
        }
        WAVE_SYNC();
        live = nn;
    }
    // ---- 3. subdivision rounds
    bool finish = false;
    int guard = 0;
    while (!finish && guard++ < 64) {
        const int prev = live, nn0 = nn;
        // every expandable node of the list, front to back (array downwards): the e-th one owns slots nn0 + 4e .. nn0 + 4e + 3
        int ne = 0;
        for (int base = 0; base < nn0; base += 64) {
            const int i = nn0 - 1 - (base + lane);
            ne += __popcll(__ballot(i >= 0 && !(S.nd[max(i, 0)].flags & (OCT_NOMORE | OCT_DEAD))));
        }
        if (nn0 + 4 * ne > nodecap) { overflow = true; break; }
        int n_to_expand = 0, e_base = 0;
        for (int base = 0; base < nn0; base += 64) {
            const int i = nn0 - 1 - (base + lane);
            Node p; p.flags = OCT_DEAD; p.count = 0;
            if (i >= 0) p = S.nd[i];
            const bool ex = i >= 0 && !(p.flags & (OCT_NOMORE | OCT_DEAD));
            const unsigned long long m = __ballot(ex);
            const int e = e_base + __popcll(m & lanemask_lt(lane));
            const bool small = ex && p.count <= OCT_LANE_LIMIT;
            int off = 0, nexp = 0;
            if (small) {
                const int mx = p.x0 + ((p.x1 - p.x0 + 1) >> 1), my = p.y0 + ((p.y1 - p.y0 + 1) >> 1);
                unsigned long long lo, hi; int tc[4];
                oct_lane_count(S, p, mx, my, lo, hi, tc);
                oct_lane_scatter(S, p, i, nn0 + 4 * e, mx, my, lo, hi, tc);
                off = (tc[0] > 0) + (tc[1] > 0) + (tc[2] > 0) + (tc[3] > 0); nexp = (tc[0] > 1) + (tc[1] > 1) + (tc[2] > 1) + (tc[3] > 1);
            }
            unsigned long long mb = __ballot(ex && !small);
            int big_off = 0, big_nexp = 0;
            while (mb) {                                          // big nodes: the whole wave on each, in list order
                const int l = __ffsll((long long)mb) - 1;
                mb &= mb - 1;
                const OctExpandResult r = oct_expand_wave(S, nn0 - 1 - (base + l), nn0 + 4 * __shfl(e, l), lane);
                big_off += r.off - 1; big_nexp += r.nexp;
            }
            int d_live = small ? off - 1 : 0, d_exp = nexp;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { d_live += __shfl_xor(d_live, d); d_exp += __shfl_xor(d_exp, d); }
            live += d_live + big_off; n_to_expand += d_exp + big_nexp;
            e_base += __popcll(m);
        }
        nn = nn0 + 4 * ne;
        WAVE_SYNC();
        int first_new;
        nn = oct_compact(S, nn, nn0, lane, first_new);
        if (live >= N || live == prev) {
            finish = true;
        } else if (live + n_to_expand * 3 > N) {
            int guard2 = 0;
            while (!finish && guard2++ < 4096) {
                const int prev2 = live;
                // build (count<<16 | index) list of the nodes created in the previous round
                int m = 0;
                for (int base = first_new; base < nn; base += 64) {
                    const int i = base + lane;
                    bool want = false; SortKey key = 0;
                    if (i < nn) { const Node o = S.nd[i]; want = o.count > 1; key = ((SortKey)o.count << OctT<BIG>::KEY_SHIFT) | (SortKey)i; }
                    const unsigned long long bm = __ballot(want);
                    if (want) { const int pos = m + __popcll(bm & lanemask_lt(lane)); if (pos < sortcap) S.sortb[pos] = key; }
                    m += __popcll(bm);
                }
                if (m > sortcap) { overflow = true; m = sortcap; }
                int mp = 1; while (mp < m) mp <<= 1;
                for (int i = m + lane; i < mp; i += 64) S.sortb[i] = 0;      // pad sorts to the front
                WAVE_SYNC();
                oct_sort(S.sortb, mp, lane);
                // largest first (sortb from the top); candidate q owns slots nn1 + 4q ..; stop after the expansion that reaches N
                const int nn1 = nn;
                int processed = 0;
                bool reached = false;
                for (int base = 0; base < m && !reached && !finish; base += 64) {
                    const int q = base + lane;
                    const bool valid = q < m;
                    const int i = valid ? (int)(S.sortb[mp - 1 - q] & 0xffff) : 0;
                    Node p; p.flags = OCT_DEAD; p.count = 0;
                    if (valid) p = S.nd[i];
                    if (__any(valid && p.count > OCT_LANE_LIMIT)) {           // rare: big nodes this late -> one at a time on the whole wave
                        for (int l = 0; l < 64 && base + l < m; l++) {
                            if (nn1 + 4 * (processed + 1) > nodecap) { overflow = true; finish = true; break; }
                            const OctExpandResult r = oct_expand_wave(S, (int)(S.sortb[mp - 1 - (base + l)] & 0xffff), nn1 + 4 * processed, lane);
                            processed++; live += r.off - 1;
                            if (live >= N) { reached = true; break; }
                        }
                        continue;
                    }
                    const int mx = p.x0 + ((p.x1 - p.x0 + 1) >> 1), my = p.y0 + ((p.y1 - p.y0 + 1) >> 1);
                    unsigned long long lo = 0, hi = 0; int tc[4] = {0, 0, 0, 0};
                    if (valid) oct_lane_count(S, p, mx, my, lo, hi, tc);
                    const int inc = valid ? (tc[0] > 0) + (tc[1] > 0) + (tc[2] > 0) + (tc[3] > 0) - 1 : 0;
                    const int incl = wave_inclusive_scan(inc);
                    const unsigned long long mh = __ballot(valid && live + incl >= N);
                    const int nvalid = min(64, m - base);
                    int cut = mh ? __ffsll((long long)mh) - 1 : nvalid - 1;   // last candidate of this chunk that is expanded
                    // node capacity: candidate q needs slots up to nn1 + 4 (processed + lane + 1)
                    const int fit = (nodecap - nn1) / 4 - processed;          // candidates of this chunk that still fit
                    if (fit < cut + 1) { overflow = true; finish = true; cut = fit - 1; }
                    if (valid && lane <= cut) oct_lane_scatter(S, p, i, nn1 + 4 * (processed + lane), mx, my, lo, hi, tc);
                    if (cut >= 0) { live += __shfl(incl, cut); processed += cut + 1; }
                    if (mh) reached = true;
                }
                nn = nn1 + 4 * processed;
                WAVE_SYNC();
                nn = oct_compact(S, nn, nn1, lane, first_new);
                if (live >= N || live == prev2) finish = true;
            }
        }
    }
    if (defer_nodes && overflow) {                                      // nothing of this level has been written yet: the full-capacity launch does it all
        if (lane == 0) lvl_tot[b * nlevels + level] |= 0x40000000;
        return;
    }
    // ---- 4. best response per node (first maximum), output in list order = array descending
    for (int base = 0; base < nn; base += 64) {
        const int i = base + lane;
        if (i < nn) {
            const Node o = S.nd[i];
            const Perm* pm = (o.flags & OCT_BUF) ? S.perm1 : S.perm0;
            uint32_t best = S.keys[pm[o.begin]];
            for (int k = 1; k < (int)o.count; k++) {
                const uint32_t key = S.keys[pm[o.begin + k]];
                if ((key >> 24) > (best >> 24)) best = key;
            }
            // level coordinates: + minBorder (reference :841-842)
            out[nn - 1 - i] = (best & 0xff000000u) | ((((best >> 12) & 0xfff) + MINB) << 12) | ((best & 0xfff) + MINB);
        }
    }
    if (lane == 0) {
        lvl_cnt[b * nlevels + level] = nn;
        if (overflow) status[b] = VIORB_ERR_CAPACITY;
    }
}

// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8U, OpenCV 2.4 integer path:
// taps {18,34,49,55,49,34,18} (x256, sum 257) in both directions, (sum + 2^15) >> 16, saturated. Everything is exact integer
// arithmetic, so neither the order of the two separable passes nor the grouping of the products changes a bit of the result.
// The kernel is bound by vector issue, not by HBM (its 2 P of traffic would take 0.1 ms), so it is built for instructions per pixel:
// no LDS, no tiles — a half-wavefront (32 lanes) owns a strip of 32 dword columns (128 px) and streams down a band of rows, one
// coalesced dword load per lane and row (BS_PF rows in flight); the two halves of a wavefront take two bands of the same strip.
//   H  the row's left / right neighbour dwords come from the adjacent lanes (DPP wave shifts; the two end lanes of a half load
//      theirs), six v_alignbyte build the byte windows and 8 v_dot4_u32_u8 give the four 7-tap row sums (<= 255 * 257, 16 bits);
//   V  a row sum is packed with the previous row's into a 16-bit pair (one v_lshl_or per pixel); an output row is
//      3 v_dot2_u32_u16 over the pairs (o, o+1), (o+2, o+3), (o+4, o+5) + one v_mad_u32_u24 for row o+6: the last six pairs of
//      every pixel stay in registers (a ring, the row loop is unrolled by six).
// ~13 vector instructions per pixel (the 64x64 LDS tile form of this kernel, V by v_pk_mad_u16 then H by v_dot2, needed ~30 of
// which half were staging and index arithmetic). Image borders: rows by reflecting the row index (scalar), columns by byte
// permutes of the edge lanes' dwords.
#define BS_LANES 32
#define BS_PF 6
__constant__ int c_gauss[7];
__global__ __launch_bounds__(64) void k_blur(const uint8_t* __restrict__ planes, uint8_t* __restrict__ blur,
                                             size_t frame_bytes, const LevelDev* __restrict__ lv,
                                             const int4* __restrict__ items, XcdPlace PL) {
    int b, item;
    if (!xcd_place(PL, b, item)) return;
    const int4 t = items[item];                       // level, first dword column of the strip, first row of band A, rows per band
    const LevelDev L = lv[t.x];
    const uint8_t* src = planes + (size_t)b * frame_bytes + L.plane_off;
    uint8_t* dst = blur + (size_t)b * frame_bytes + L.plane_off;
    const int lane = threadIdx.x, hl = lane & (BS_LANES - 1);
    const bool upper = lane >= BS_LANES;
    const int Rb = t.w, y0A = t.z, y0B = t.z + Rb;
    const int rows = upper ? min(Rb, L.h - y0B) : min(Rb, L.h - y0A);          // band B may be short or empty
    const int sdw = L.stride >> 2;
    const int d = t.y + hl;                                                     // this lane's dword column
    const int dlast = (L.w - 1) >> 2, m = (L.w - 1) & 3;                        // dword and byte of the last pixel of a row
    const int dc = min(d, sdw - 1);
    const bool end_lane = hl == 0 || hl == BS_LANES - 1;
    const int dn = hl == 0 ? max(dc - 1, 0) : min(dc + 1, sdw - 1);            // neighbour column an end lane loads itself
    // byte selectors of the right border (BORDER_REFLECT_101: pixel w - 1 + k = pixel w - 1 - k) over the stream [left dword, dword]:
    // bytes of the last dword beyond byte m, and the dword after it
    const uint32_t selC = m == 0 ? 0x01020304u : m == 1 ? 0x03040504u : m == 2 ? 0x05060504u : 0x07060504u;
    const uint32_t selR = m == 0 ? 0x0c0c0c0cu : m == 1 ? 0x0c000102u : m == 2 ? 0x01020304u : 0x03040506u;
    const bool has_edge = t.y == 0 || t.y + BS_LANES > dlast - 1;                // wave-uniform
    const bool is_d0 = d == 0, is_dl = d == dlast, is_dl1 = d == dlast - 1;
    const uint32_t k0 = (uint32_t)c_gauss[0], k1 = (uint32_t)c_gauss[1], k2 = (uint32_t)c_gauss[2], k3 = (uint32_t)c_gauss[3];
    const uint32_t Ka = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), Kb = k2 | (k1 << 8) | (k0 << 16);   // taps -3..0, +1..+3
    const uint32_t K01 = k0 | (k1 << 16), K23 = k2 | (k3 << 16), K45 = k2 | (k1 << 16);
    const int nrow = Rb + 6;                                                    // input rows per band
    // Loads are unconditional (row index clamped, every lane also loads a neighbour dword although only the end lanes use it) and the
    // row loop runs to a multiple of six without an exit: any branch around a load or inside the unrolled body made the compiler copy the
    // register ring and wait for each load where it was issued.
    const uint8_t* pc = src + 4u * (uint32_t)dc;
    const uint8_t* pn = src + 4u * (uint32_t)(end_lane ? dn : dc);
    uint32_t cb[BS_PF], nb[BS_PF];
    auto request = [&](int i, uint32_t& c, uint32_t& n) {
        const int ii = min(i, nrow - 1);
        int ya = y0A + ii - 3, yb = y0B + ii - 3;                               // both bands' input row, reflected at the image border
        ya = ya < 0 ? -ya : ya; ya = ya >= L.h ? 2 * L.h - 2 - ya : ya; ya = min(max(ya, 0), L.h - 1);
        yb = yb < 0 ? -yb : yb; yb = yb >= L.h ? 2 * L.h - 2 - yb : yb; yb = min(max(yb, 0), L.h - 1);
        const uint32_t offA = __builtin_amdgcn_readfirstlane(ya * L.stride), offB = __builtin_amdgcn_readfirstlane(yb * L.stride);
        const uint32_t off = upper ? offB : offA;
        c = *reinterpret_cast<const uint32_t*>(pc + off);
        n = *reinterpret_cast<const uint32_t*>(pn + off);
    };
#pragma unroll
    for (int u = 0; u < BS_PF; u++) request(u, cb[u], nb[u]);
    uint32_t P[6][4], hp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 6; u++) { P[u][0] = P[u][1] = P[u][2] = P[u][3] = 0; }
    uint8_t* pd = dst + 4u * (uint32_t)min(d, sdw - 1);
    const bool col_ok = d < sdw;
    for (int i0 = 0; i0 < nrow; i0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int i = i0 + u;
            uint32_t c = cb[u];
            uint32_t Lr = __builtin_amdgcn_update_dpp(c, c, 0x138, 0xf, 0xf, true);        // wave_shr:1 — lane n reads lane n - 1
            uint32_t Rr = __builtin_amdgcn_update_dpp(c, c, 0x130, 0xf, 0xf, true);        // wave_shl:1 — lane n reads lane n + 1
            Lr = hl == 0 ? nb[u] : Lr; Rr = hl == BS_LANES - 1 ? nb[u] : Rr;
            request(i + BS_PF, cb[u], nb[u]);
            if (has_edge) {
                const uint32_t Lf = __builtin_amdgcn_perm(Rr, c, 0x01020304u);               // pixels -4 .. -1 = pixels 4 .. 1
                const uint32_t Cf = __builtin_amdgcn_perm(c, Lr, selC), Rf = __builtin_amdgcn_perm(c, Lr, selR);
                const uint32_t Rf1 = __builtin_amdgcn_perm(Rr, c, selC);                     // the dword before the last: its right neighbour is the patched last dword
                Lr = is_d0 ? Lf : Lr;
                Rr = is_dl ? Rf : (is_dl1 ? Rf1 : Rr);
                c = is_dl ? Cf : c;
            }
            // H: row sums of the lane's four pixels
            uint32_t h[4];
            h[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, Lr, 1), Ka, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rr, c, 1), Kb, 0u, false), false);
            h[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, Lr, 2), Ka, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rr, c, 2), Kb, 0u, false), false);
            h[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(c, Lr, 3), Ka, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Rr, c, 3), Kb, 0u, false), false);
            h[3] = __builtin_amdgcn_udot4(c, Ka, __builtin_amdgcn_udot4(Rr, Kb, 0u, false), false);
            // V: rows o .. o + 6 = i - 6 .. i; the pair of rows (r - 1, r) lives in P[r % 6]
            uint32_t a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                a[j] = dot2_u16(P[(u + 1) % 6][j], K01, 1u << 15);
                a[j] = dot2_u16(P[(u + 3) % 6][j], K23, a[j]);
                a[j] = dot2_u16(P[(u + 5) % 6][j], K45, a[j]);
                a[j] = __umul24(h[j], k0) + a[j];
                a[j] = min(a[j], 0x00ffffffu);                                               // (sum + 2^15) >> 16 saturated to 255 = byte 2
                P[u][j] = hp[j] | (h[j] << 16);
                hp[j] = h[j];
            }
            const int o = i - 6;
            const uint32_t yoA = __builtin_amdgcn_readfirstlane((y0A + o) * L.stride), yoB = __builtin_amdgcn_readfirstlane((y0B + o) * L.stride);
            if (o >= 0 && o < rows && col_ok) {
                const uint32_t lo = __builtin_amdgcn_perm(a[1], a[0], 0x0c0c0602u), hi = __builtin_amdgcn_perm(a[3], a[2], 0x0c0c0602u);
                *reinterpret_cast<uint32_t*>(pd + (upper ? yoB : yoA)) = lo | (hi << 16);
            }
        }
    }
}

// One wavefront per keypoint: IC_Angle (reference :77-104) on the un-blurred level, then the
// steered 256-bit BRIEF (reference :107-147) on the blurred level, then the cv::KeyPoint record
// (reference :837-847, :1095-1101).
__constant__ __attribute__((aligned(16))) uint32_t c_pattern[256];   // x0 | y0<<8 | x1<<16 | y1<<24 (int8 each)
__constant__ int c_umax[16];
__constant__ uint32_t c_orient_mask[256];              // [row 0..31][dword 0..7]: 0xff per byte inside the circle (row 31: 0)
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
#define DESC_WIN_HALF 18
#define DESC_WIN_ROWS 37
#define DESC_WIN_DW 10
#define DESC_WIN_TRIPS 6           // ceil(37 * 10 / 64)
__global__ __launch_bounds__(256) void k_orient_describe(const uint8_t* __restrict__ planes,
                                                         const uint8_t* __restrict__ blur, size_t frame_bytes,
                                                         const LevelDev* __restrict__ lv, int nlevels,
                                                         const uint32_t* __restrict__ lvl_kp, int kp_pitch,
                                                         const int* __restrict__ lvl_cnt,
                                                         viorb_keypoint* __restrict__ out_kp,
                                                         uint8_t* __restrict__ out_desc, int out_cap,
                                                         int* __restrict__ out_cnt, int* __restrict__ status, XcdPlace PL) {
    const int lane = threadIdx.x & 63;
    int b, item;
    if (!xcd_place(PL, b, item)) return;
    const int k = item * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);             // keypoint slot inside the image (wave-uniform: the level search below stays on the scalar unit)
    const int* cnt = lvl_cnt + b * nlevels;
    int level = -1, local = 0, total = 0;
    for (int l = 0; l < nlevels; l++) {
        const int c = cnt[l];
        if (level < 0 && k < total + c) { level = l; local = k - total; }
        total += c;
    }
    // out_cap = viorb_extractor_max_keypoints_for(image size): per level quota + 2, or 4 x roots where the unchecked first round keeps more
    // (592x158 with 172 features over 8 levels of 1.1 keeps 24 on a level whose quota is 15); the pitch of every per-keypoint array
    // downstream. Exceeding it would be a bug: reported, never truncated silently
    if (k == 0 && lane == 0) { out_cnt[b] = min(total, out_cap); if (total > out_cap) status[b] = VIORB_ERR_CAPACITY; }
    if (level < 0 || k >= out_cap) return;
    const LevelDev L = lv[level];
    const uint32_t key = lvl_kp[(size_t)b * kp_pitch + L.kp_off + local];
    const int cx = (int)(key & 0xfff), cy = (int)((key >> 12) & 0xfff), score = (int)(key >> 24);
    // ---- orientation: integer moments over the circular patch. Lane = (row r8, dword j) of an 8-row x 32-byte slab, four
    // slabs cover the 31 rows; the circle is a constant byte-mask table, so the four (unaligned) dword loads are
    // independent and branch-free; sum I and sum (u+15) I come from two v_dot4_u32_u8 per dword.
    const uint8_t* img = planes + (size_t)b * frame_bytes + L.plane_off + (size_t)cy * L.stride + cx;
    // The 512 descriptor samples of a keypoint lie within +-18 px (the largest pattern radius is 18.38): the 37 x 40-byte window of
    // the blurred level is requested now, as 370 row-contiguous dwords, and goes through LDS — eight scattered byte loads per lane
    // touched ~30 cache lines per instruction and made the texture-address unit this kernel's limit.
    __shared__ uint32_t s_win[4][DESC_WIN_ROWS * DESC_WIN_DW];
    uint32_t* win = s_win[threadIdx.x >> 6];
    const uint8_t* bim = blur + (size_t)b * frame_bytes + L.plane_off + (size_t)cy * L.stride + cx;
    uint32_t wv[DESC_WIN_TRIPS];
#pragma unroll
    for (int it = 0; it < DESC_WIN_TRIPS; it++) {
        const int i = min(lane + 64 * it, DESC_WIN_ROWS * DESC_WIN_DW - 1);
        const int wr = i / DESC_WIN_DW, wc = i - wr * DESC_WIN_DW;
        wv[it] = *reinterpret_cast<const u32_unaligned*>(bim + __mul24(wr - DESC_WIN_HALF, L.stride) - DESC_WIN_HALF + 4 * wc);      // 24-bit multiplies: v_mul_lo_u32 is quarter rate
    }
    int m10 = 0, m01 = 0;
    {
        const int j = lane & 7, r8 = lane >> 3;
        const uint32_t wu = 0x03020100u + 0x04040404u * (uint32_t)j;       // u + 15 of the four bytes
        const uint8_t* col = img - HALF_PATCH + 4 * j;
        uint32_t px[4]; int vv[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = min(it * 8 + r8, PATCH - 1);                   // slab 3 has 7 rows; the 8th is masked off
            vv[it] = row - HALF_PATCH;
            px[it] = *reinterpret_cast<const u32_unaligned*>(col + __mul24(vv[it], L.stride));
        }
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const uint32_t q = px[it] & c_orient_mask[it * 64 + lane];
            const int s1 = (int)__builtin_amdgcn_udot4(q, 0x01010101u, 0u, false);
            const int su = (int)__builtin_amdgcn_udot4(q, wu, 0u, false);
            m10 += su - __mul24(HALF_PATCH, s1);
            m01 += __mul24(vv[it], s1);
        }
    }
    // wave sums by DPP (the inclusive scan's last lane holds the total) instead of six ds_bpermute butterflies per moment
    m10 = __builtin_amdgcn_readlane(wave_inclusive_scan(m10), 63); m01 = __builtin_amdgcn_readlane(wave_inclusive_scan(m01), 63);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // ---- descriptor
    const float factor_pi = (float)(3.14159265358979323846 / 180.f);
    float sn, cs;
    sincos_f32(angle * factor_pi, &sn, &cs);
    const float a = cs, bb = sn;
#pragma unroll
    for (int it = 0; it < DESC_WIN_TRIPS; it++) {
        const int i = lane + 64 * it;
        if (i < DESC_WIN_ROWS * DESC_WIN_DW) win[i] = wv[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint8_t* w8 = reinterpret_cast<const uint8_t*>(win) + DESC_WIN_HALF * (DESC_WIN_DW * 4) + DESC_WIN_HALF;
    uint32_t nib = 0;
    const uint4 pq = reinterpret_cast<const uint4*>(c_pattern)[lane];
    const uint32_t pqa[4] = {pq.x, pq.y, pq.z, pq.w};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t q = pqa[t];
        const float x0 = (float)(int8_t)(q & 0xff), y0 = (float)(int8_t)((q >> 8) & 0xff);
        const float x1 = (float)(int8_t)((q >> 16) & 0xff), y1 = (float)(int8_t)(q >> 24);
        const int r0 = round_half_even(x0 * bb + y0 * a), c0 = round_half_even(x0 * a - y0 * bb);
        const int r1 = round_half_even(x1 * bb + y1 * a), c1 = round_half_even(x1 * a - y1 * bb);
        const int t0 = w8[__mul24(r0, DESC_WIN_DW * 4) + c0], t1 = w8[__mul24(r1, DESC_WIN_DW * 4) + c1];
        nib |= (uint32_t)(t0 < t1) << t;
    }
    // lane i holds bits 4i..4i+3: bytes from lane pairs, dwords from 8-lane groups
    uint32_t byte = nib | (__shfl_down(nib, 1) << 4);
    uint32_t word = byte | (__shfl_down(byte, 2) << 8) | (__shfl_down(byte, 4) << 16) | (__shfl_down(byte, 6) << 24);
    const size_t o = (size_t)b * out_cap + k;
    if ((lane & 7) == 0) reinterpret_cast<uint32_t*>(out_desc + o * 32)[lane >> 3] = word;
    if (lane == 0) {
        viorb_keypoint kp;
        kp.x = level ? (float)cx * L.scale : (float)cx;
        kp.y = level ? (float)cy * L.scale : (float)cy;
        kp.size = L.kp_size; kp.angle = angle; kp.response = (float)score;
        kp.octave = level; kp.class_id = -1;
        out_kp[o] = kp;
    }
}

// ---------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (reference src/Frame.cc:646-820), one workgroup per rectified stereo pair.
//   1. right keypoints sorted by y (bitonic, u64 keys in LDS) so that the reference's row table
//      vRowIndices[(int)vL] becomes a y-window scan; the best candidate is the minimum of
//      (Hamming << 16 | iR), i.e. the first minimum in the reference's iR push order;
//   2. 11x11 SAD over 11 shifts on the un-blurred pyramid level of the left keypoint (integer arithmetic: the
//      centre-subtracted float patches of the reference hold small integers), parabola fit, disparity -> depth;
//   3. median-based rejection: sort (SAD << 16 | iL), drop SAD >= 1.5 * 1.4 * median.
// ---------------------------------------------------------------------------------------------
struct StereoArgs {
    const viorb_keypoint *kl, *kr; const uint8_t *dl, *dr; const int *nl, *nr;
    const uint8_t *planesL, *planesR; size_t frame_bytes; const LevelDev* lv;
    int cap, sort_n, nlevels; float bf, fx;
    float scale[16], inv_scale[16];
    float* uright; float* depth; int* nmatched;
    unsigned char* work; size_t work_bytes;      // k_stereo_match<true>: the sort / SAD arrays in global memory (more features than LDS holds)
};
__host__ __device__ inline size_t stereo_lds_bytes(int cap, int sort_n) { return (size_t)sort_n * (8 + 4 + 4 + 4) + (size_t)cap * 4 + 64; }

template <bool GW>
__global__ __launch_bounds__(1024) void k_stereo_match(StereoArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_st_lds[];
    unsigned char* s_st = GW ? A.work + (size_t)blockIdx.x * A.work_bytes : s_st_lds;
    const int p = blockIdx.x, t = threadIdx.x, cap = A.cap, sn = A.sort_n;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(s_st);            // [sn] (ybits << 32 | iR)
    float* ry = reinterpret_cast<float*>(keys + sn);                                    // [sn] y of the sorted right keypoints
    int* rid = reinterpret_cast<int*>(ry + sn);                                         // [sn] their indices
    uint32_t* skeys = reinterpret_cast<uint32_t*>(rid + sn);                            // [sn] (sad << 16 | iL)
    int* sad = reinterpret_cast<int*>(skeys + sn);                                      // [cap]
    __shared__ int s_cnt, s_removed;
    const int N = min(A.nl[p], cap), Nr = min(A.nr[p], cap);
    const viorb_keypoint* kl = A.kl + (size_t)p * cap; const viorb_keypoint* kr = A.kr + (size_t)p * cap;
    const uint8_t* dl = A.dl + (size_t)p * cap * 32; const uint8_t* dr = A.dr + (size_t)p * cap * 32;
    float* uR = A.uright + (size_t)p * cap; float* dep = A.depth + (size_t)p * cap;
    for (int i = t; i < sn; i += blockDim.x)
        keys[i] = i < Nr ? (((unsigned long long)__float_as_uint(kr[i].y) << 32) | (unsigned)i) : ~0ull;    // y > 0: float bits order like the values
    if (t == 0) { s_cnt = 0; s_removed = 0; }
    __syncthreads();
    for (int k = 2; k <= sn; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = t; q < (sn >> 1); q += blockDim.x) {
                const int lo = ((q & ~(j - 1)) << 1) | (q & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long x = keys[lo], y = keys[hi];
                if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();
        }
    for (int i = t; i < sn; i += blockDim.x) { ry[i] = i < Nr ? __uint_as_float((unsigned)(keys[i] >> 32)) : 3.0e38f; rid[i] = (int)(keys[i] & 0xffffffffu); }
    __syncthreads();
    const float mb = A.bf / A.fx, minD = 0.f, maxD = A.bf / mb;
    const float rmax = 2.0f * A.scale[A.nlevels - 1];
    const int nRows = A.lv[0].h;
    const int thOrbDist = (100 + 50) / 2;
    for (int iL = t; iL < cap; iL += blockDim.x) {
        float out_u = -1.f, out_d = -1.f; int out_sad = -1;
        if (iL < N) {
            const viorb_keypoint kpL = kl[iL];
            const int levelL = kpL.octave; const float vL = kpL.y, uL = kpL.x;
            const int rowi = (int)vL;
            const float minU = uL - maxD, maxU = uL - minD;
            if (rowi >= 0 && rowi < nRows && !(maxU < 0)) {
                const float ylo = (float)rowi - rmax - 1.0f, yhi = (float)rowi + rmax + 1.0f;
                int lo = 0, hi = Nr;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (ry[mid] < ylo) lo = mid + 1; else hi = mid; }
                const uint4* pl = reinterpret_cast<const uint4*>(dl + (size_t)iL * 32);
                const uint4 da = pl[0], db = pl[1];
                uint32_t best = 0xffffffffu;
                for (int k = lo; k < Nr && ry[k] <= yhi; k++) {
                    const int iR = rid[k];
                    const viorb_keypoint kpR = kr[iR];
                    const float r = 2.0f * A.scale[kpR.octave];
                    const int maxr = (int)ceilf(kpR.y + r), minr = (int)floorf(kpR.y - r);
                    if (rowi < minr || rowi > maxr) continue;
                    if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
                    if (!(kpR.x >= minU && kpR.x <= maxU)) continue;
                    const uint4* pr = reinterpret_cast<const uint4*>(dr + (size_t)iR * 32);
                    const uint4 ea = pr[0], eb = pr[1];
                    const int dist = __popc(da.x ^ ea.x) + __popc(da.y ^ ea.y) + __popc(da.z ^ ea.z) + __popc(da.w ^ ea.w) +
                                     __popc(db.x ^ eb.x) + __popc(db.y ^ eb.y) + __popc(db.z ^ eb.z) + __popc(db.w ^ eb.w);
                    if (dist < 100) best = min(best, ((uint32_t)dist << 16) | (uint32_t)iR);
                }
                if (best != 0xffffffffu && (int)(best >> 16) < thOrbDist) {
                    const int bestIdxR = (int)(best & 0xffff);
                    const float uR0 = kr[bestIdxR].x;
                    const float sfac = A.inv_scale[levelL];
                    const int cu = (int)roundf(kpL.x * sfac), cv = (int)roundf(kpL.y * sfac), cr = (int)roundf(uR0 * sfac);
                    const LevelDev Lv = A.lv[levelL];
                    const int w = 5, L = 5;
                    const float iniu = (float)cr + L - w, endu = (float)cr + L + w + 1;
                    const bool inside = !(iniu < 0 || endu >= Lv.w) && cv - w >= 0 && cv + w < Lv.h && cu - w >= 0 && cu + w < Lv.w &&
                                        cr - L - w >= 0 && cr + L + w < Lv.w;
                    if (inside) {
                        const uint8_t* IL = A.planesL + (size_t)p * A.frame_bytes + Lv.plane_off;
                        const uint8_t* IR = A.planesR + (size_t)p * A.frame_bytes + Lv.plane_off;
                        const int cL = IL[(size_t)cv * Lv.stride + cu];
                        int cRv[11], acc[11];
#pragma unroll
                        for (int sft = 0; sft < 11; sft++) { cRv[sft] = IR[(size_t)cv * Lv.stride + cr + sft - L]; acc[sft] = 0; }
                        for (int dy = -w; dy <= w; dy++) {
                            const uint8_t* rl = IL + (size_t)(cv + dy) * Lv.stride + cu - w;
                            const uint8_t* rr = IR + (size_t)(cv + dy) * Lv.stride + cr - L - w;
                            int lrow[11], rrow[21];
#pragma unroll
                            for (int x = 0; x < 11; x++) lrow[x] = (int)rl[x] - cL;
#pragma unroll
                            for (int x = 0; x < 21; x++) rrow[x] = rr[x];
#pragma unroll
                            for (int sft = 0; sft < 11; sft++)
#pragma unroll
                                for (int x = 0; x < 11; x++) acc[sft] += abs(lrow[x] - (rrow[sft + x] - cRv[sft]));
                        }
                        int bestSad = 0x7fffffff, bestinc = 0;
#pragma unroll
                        for (int sft = 0; sft < 11; sft++) if (acc[sft] < bestSad) { bestSad = acc[sft]; bestinc = sft - L; }
                        if (!(bestinc == -L || bestinc == L)) {
                            float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                            for (int sft = 1; sft < 10; sft++) if (sft - L == bestinc) { d1 = (float)acc[sft - 1]; d2 = (float)acc[sft]; d3 = (float)acc[sft + 1]; }
                            const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
                            if (!(deltaR < -1 || deltaR > 1)) {
                                float bestuR = A.scale[levelL] * ((float)cr + (float)bestinc + deltaR);
                                float disparity = uL - bestuR;
                                if (disparity >= minD && disparity < maxD) {
                                    if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }   // "bestuR = uL-0.01": a double subtraction rounded once (Frame.cc:797)
                                    out_d = A.bf / disparity; out_u = bestuR; out_sad = bestSad;
                                }
                            }
                        }
                    }
                }
            }
        }
        uR[iL] = out_u; dep[iL] = out_d; sad[iL] = out_sad;
    }
    __syncthreads();
    // ---- median rejection (SAD <= 121 * 510 fits 16 bits)
    for (int i = t; i < sn; i += blockDim.x) skeys[i] = (i < cap && sad[i] >= 0) ? (((uint32_t)sad[i] << 16) | (uint32_t)i) : 0xffffffffu;
    for (int i = t; i < cap; i += blockDim.x) if (sad[i] >= 0) atomicAdd(&s_cnt, 1);
    __syncthreads();
    for (int k = 2; k <= sn; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = t; q < (sn >> 1); q += blockDim.x) {
                const int lo = ((q & ~(j - 1)) << 1) | (q & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const uint32_t x = skeys[lo], y = skeys[hi];
                if ((x > y) == up) { skeys[lo] = y; skeys[hi] = x; }
            }
            __syncthreads();
        }
    const int cnt = s_cnt;
    if (cnt > 0) {
        const float median = (float)(skeys[cnt / 2] >> 16);
        const float thDist = 1.5f * 1.4f * median;
        for (int i = t; i < cnt; i += blockDim.x) {
            const uint32_t e = skeys[i];
            if (!((float)(e >> 16) < thDist)) { const int iL = (int)(e & 0xffff); uR[iL] = -1.f; dep[iL] = -1.f; atomicAdd(&s_removed, 1); }
        }
    }
    __syncthreads();
    if (t == 0) A.nmatched[p] = cnt - s_removed;
}

} // namespace viorb

// ---------------------------------------------------------------------------------------------
// Host side: handle, geometry, launches, C ABI
// ---------------------------------------------------------------------------------------------
using namespace viorb;

struct viorb_extractor {
    viorb_extractor_params p;
    int max_batch = 1, device = 0;
    // ctor tables (reference :410-470)
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> quota;
    int umax[16];
    // geometry for the current image size
    int img_w = 0, img_h = 0;
    std::vector<LevelDev> lv;
    std::vector<CellDesc> cells;
    std::vector<int4> blur_tiles;
    size_t frame_bytes = 0;
    int slot_cap = 0, kp_pitch = 0, out_cap = 0;
    int fast_tile_pitch = 0, fast_tile_rows = 0, fast_tile_bytes = 0, fast_score_bytes = 0, fast_list_cap = 0;
    bool fast_v3 = false;            // every cell fits k_fast_cells3's compile-time LDS geometry
    int fast3_tile_bytes = 0, fast3_score_bytes = 0;
    int oct_ncap = 0, oct_nodecap = 0, oct_sortcap = 0;
    std::vector<std::pair<int, int>> oct_tiers;            // (first level, candidates) of the higher levels' own first quadtree launches
    int oct_first_cap = 0, oct_node_x10 = 50;              // candidates of the common first launch; node slots of the first launches in tenths of the quota (50 = never short)
    int oct_big_cap = 0, oct_big_slots = 0; uint32_t* d_oct_big = nullptr; int* d_oct_big_next = nullptr; int* d_lvl_tot = nullptr;   // over-size levels (k_octree<true>)
    unsigned char* d_stereo_work = nullptr; size_t stereo_work_bytes = 0;                // k_stereo_match<true>
    bool oct_huge = false; uint8_t* d_oct_nodes = nullptr; size_t oct_node_bytes = 0;     // a per-level quota whose node list does not fit LDS: nodes + sort keys in global scratch too
    std::vector<int> rs_pitch_dw, rs_rows;
    // second resize form (k_resize2): per-level tile table, LDS pitch, whether the level qualifies; whether level 1's kernel may also write level 0
    std::vector<int4> rs2_tiles; std::vector<int> rs2_off, rs2_pitch_dw, rs2_ok; bool rs2_copy_ok = false;
    int4* d_rs2_tiles = nullptr;
    // third resize form (k_resize_stream): per-level work items {strip, band}, emit table over the source rows, whether the level qualifies
    std::vector<int4> rss_items; std::vector<uint2> rss_etab; std::vector<int> rss_item_off, rss_etab_off, rss_ok;
    int4* d_rss_items = nullptr; uint2* d_rss_etab = nullptr;
    // device memory
    uint8_t *d_planes = nullptr, *d_blur = nullptr, *d_desc = nullptr, *d_stage = nullptr;
    LevelDev* d_lv = nullptr; CellDesc* d_cells = nullptr; int4* d_blur_tiles = nullptr;
    int2 *d_xtab = nullptr, *d_ytab = nullptr;
    uint32_t *d_slots = nullptr, *d_lvl_kp = nullptr;
    int *d_cell_cnt = nullptr, *d_lvl_cnt = nullptr, *d_lvl_ncand = nullptr, *d_count = nullptr, *d_status = nullptr;
    viorb_keypoint* d_kps = nullptr;
    size_t stage_bytes = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t aux_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // the blur's side stream (launch_all)
    hipStream_t last_stream = nullptr;
    int last_batch = 0;
    unsigned fast_prof_rot = 0;          // which FAST sub-launch the profiler times on this call
    bool tables_uploaded = false;
};

static void free_device(viorb_extractor* h) {
    void* ptrs[] = {h->d_planes, h->d_blur, h->d_desc, h->d_stage, h->d_lv, h->d_cells, h->d_blur_tiles, h->d_xtab,
                    h->d_ytab, h->d_slots, h->d_lvl_kp, h->d_cell_cnt, h->d_lvl_cnt, h->d_lvl_ncand, h->d_count,
                    h->d_status, h->d_kps, h->d_rs2_tiles, h->d_rss_items, h->d_rss_etab, h->d_oct_big, h->d_oct_big_next, h->d_lvl_tot, h->d_oct_nodes};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    h->d_rs2_tiles = nullptr; h->d_rss_items = nullptr; h->d_rss_etab = nullptr; h->d_oct_big = nullptr; h->d_oct_big_next = nullptr; h->d_lvl_tot = nullptr; h->d_oct_nodes = nullptr;
    h->d_planes = h->d_blur = h->d_desc = h->d_stage = nullptr; h->d_lv = nullptr; h->d_cells = nullptr;
    h->d_blur_tiles = nullptr; h->d_xtab = h->d_ytab = nullptr; h->d_slots = h->d_lvl_kp = nullptr;
    h->d_cell_cnt = h->d_lvl_cnt = h->d_lvl_ncand = h->d_count = h->d_status = nullptr; h->d_kps = nullptr;
    h->stage_bytes = 0;
}

// Keypoints an image of w x h can return: per level what the quadtree keeps at most — quota + 2 after a checked round, or every child of the
// unchecked first round (4 per root, roots = round(width / height) of the bordered level, reference :514-552) where that is more.
static int max_keypoints_for_size(const viorb_extractor* h, int w, int hgt) {
    int cap = 0;
    for (int l = 0; l < h->p.nlevels; l++) {
        const float s = h->inv_scale[l];
        const int lw = host_cv_round((double)((float)w * s)), lh = host_cv_round((double)((float)hgt * s));
        const int ow = lw - 2 * MINB, oh = lh - 2 * MINB;
        const int n_ini = (oh > 0 && ow > 0) ? (int)roundf((float)ow / (float)oh) : 0;
        cap += std::max(h->quota[l] + 2, 4 * n_ini);
    }
    return cap;
}

// Build level geometry, FAST cell table, resize tables and buffers for a w x h image.
static int configure(viorb_extractor* h, int w, int hgt) {
    const int nl = h->p.nlevels;
    free_device(h);
    h->lv.assign(nl, LevelDev());
    h->cells.clear(); h->blur_tiles.clear();
    std::vector<int2> xtab, ytab;
    h->rs_pitch_dw.assign(nl, 0); h->rs_rows.assign(nl, 0);
    h->rs2_tiles.clear(); h->rs2_off.assign(nl, 0); h->rs2_pitch_dw.assign(nl, 1); h->rs2_ok.assign(nl, 0); h->rs2_copy_ok = false;
    h->rss_items.clear(); h->rss_etab.clear(); h->rss_item_off.assign(nl + 1, 0); h->rss_etab_off.assign(nl, 0); h->rss_ok.assign(nl, 0);
    size_t off = 0;
    int kp_off = 0, max_cw = 0, max_ch = 0;
    for (int l = 0; l < nl; l++) {
        LevelDev& L = h->lv[l];
        const float s = h->inv_scale[l];
        L.w = host_cv_round((double)((float)w * s));
        L.h = host_cv_round((double)((float)hgt * s));
        if (L.w < 1 || L.h < 1 || L.w > 4095 || L.h > 4095) { set_error("level %d size %dx%d unsupported", l, L.w, L.h); return VIORB_ERR_UNSUPPORTED; }
        L.stride = (int)align_up(L.w, 64);
        L.plane_off = (uint32_t)off;
        off += align_up((size_t)L.stride * L.h, 256);
        L.quota = h->quota[l];
        L.scale = h->scale[l];
        L.kp_size = (float)(int)(PATCH * h->scale[l]);
        // FAST cell grid (reference :769-806)
        const int maxBX = L.w - MINB, maxBY = L.h - MINB;
        const float width = (float)(maxBX - MINB), height = (float)(maxBY - MINB);
        L.oct_w = maxBX - MINB; L.oct_h = maxBY - MINB;
        L.cell_base = (int)h->cells.size();
        const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
        if (nCols >= 1 && nRows >= 1) {
            const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(MINB + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(MINB + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    CellDesc c;
                    c.level = l; c.x0 = (int)iniX; c.y0 = (int)iniY;
                    c.cw = (int)maxX - (int)iniX; c.ch = (int)maxY - (int)iniY;
                    c.shx = j * wCell; c.shy = i * hCell; c.pitch = (int)align_up(c.cw + 3, 4) + 4;
                    c.src_off = (uint32_t)(L.plane_off + (size_t)c.y0 * L.stride + (size_t)(c.x0 & ~3)); c.stride = L.stride;
                    if (c.cw < 7 || c.ch < 7) continue;            // cv::FAST finds nothing in such a sub-image
                    h->cells.push_back(c);
                    max_cw = std::max(max_cw, (int)c.cw); max_ch = std::max(max_ch, (int)c.ch);
                }
            }
        }
        L.ncells = (int)h->cells.size() - L.cell_base;
        // quadtree roots (reference :542-545)
        L.n_ini = (L.oct_h > 0 && L.oct_w > 0) ? (int)roundf((float)L.oct_w / (float)L.oct_h) : 0;
        L.hx = L.n_ini > 0 ? (float)L.oct_w / (float)L.n_ini : 1.f;
        // the first round splits every root unchecked (<= 4*n_ini nodes); afterwards <= quota + 2
        L.kp_off = kp_off;
        kp_off += std::max(L.quota, 4 * L.n_ini) + 4;
        // resize tables from level l-1 (cv::resize INTER_LINEAR 8U, OpenCV 2.4 recipe)
        L.xtab_off = (int)xtab.size(); L.ytab_off = (int)ytab.size();
        if (l > 0) {
            const int sw = h->lv[l - 1].w, sh = h->lv[l - 1].h;
            const double inv_sx = (double)L.w / sw, inv_sy = (double)L.h / sh;
            const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
            for (int dx = 0; dx < L.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = host_cv_floor(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx + 1 >= sw && sx >= sw - 1) { fx = 0; sx = sw - 1; }
                const int a0 = clampi(host_cv_round((double)((1.f - fx) * 2048)), -32768, 32767);
                const int a1 = clampi(host_cv_round((double)(fx * 2048)), -32768, 32767);
                const int sx1 = std::min(sx + 1, sw - 1);
                xtab.push_back(make_int2(sx | (sx1 << 16), (a0 & 0xffff) | (a1 << 16)));
            }
            for (int dy = 0; dy < L.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = host_cv_floor(fy);
                fy -= sy;
                const int b0 = clampi(host_cv_round((double)((1.f - fy) * 2048)), -32768, 32767);
                const int b1 = clampi(host_cv_round((double)(fy * 2048)), -32768, 32767);
                const int sy0 = clampi(sy, 0, sh - 1), sy1 = clampi(sy + 1, 0, sh - 1);
                ytab.push_back(make_int2(sy0 | (sy1 << 16), (b0 & 0xffff) | (b1 << 16)));
            }
            // LDS extents of the worst 64x16 tile
            int maxdw = 1, maxrows = 1;
            const int2* xt = xtab.data() + L.xtab_off; const int2* yt = ytab.data() + L.ytab_off;
            for (int tx = 0; tx < L.w; tx += RS_TW) {
                const int x1 = std::min(tx + RS_TW, L.w) - 1;
                const int xs0 = (xt[tx].x & 0xffff) & ~3, xs1 = xt[x1].x >> 16;
                maxdw = std::max(maxdw, ((xs1 - xs0) >> 2) + 1);
            }
            for (int ty = 0; ty < L.h; ty += RS_TH) {
                const int y1 = std::min(ty + RS_TH, L.h) - 1;
                maxrows = std::max(maxrows, (yt[y1].x >> 16) - (yt[ty].x & 0xffff) + 1);
            }
            h->rs_pitch_dw[l] = maxdw; h->rs_rows[l] = maxrows;
            if ((size_t)maxdw * maxrows * 4 > 60000) { set_error("resize tile does not fit LDS (scale factor too large)"); return VIORB_ERR_UNSUPPORTED; }
            {   // 64x32 tiles of the second form: source rectangle per tile, and whether the rectangles tile the source without gaps
                h->rs2_off[l] = (int)h->rs2_tiles.size();
                int pdw = 1, prow = 1; bool cover = true; int next_x = 0, next_y = 0;
                const int gx2 = (L.w + RS2_TW - 1) / RS2_TW, gy2 = (L.h + RS2_TH - 1) / RS2_TH;
                for (int ty = 0; ty < gy2; ty++) {
                    const int y0t = ty * RS2_TH, y1t = std::min(y0t + RS2_TH, L.h) - 1;
                    const int ys0 = yt[y0t].x & 0xffff, ys1 = yt[y1t].x >> 16;
                    if (ys0 > next_y) cover = false;
                    next_y = std::max(next_y, ys1 + 1);
                    next_x = 0;
                    for (int tx = 0; tx < gx2; tx++) {
                        const int x0t = tx * RS2_TW, x1t = std::min(x0t + RS2_TW, L.w) - 1;
                        const int xs0 = (xt[x0t].x & 0xffff) & ~3, xs1 = xt[x1t].x >> 16;
                        if (xs0 > next_x) cover = false;
                        next_x = std::max(next_x, (xs1 | 3) + 1);
                        const int ndw = ((xs1 - xs0) >> 2) + 1, nrows = ys1 - ys0 + 1;
                        pdw = std::max(pdw, ndw); prow = std::max(prow, nrows);
                        h->rs2_tiles.push_back(make_int4(xs0, ndw, ys0, nrows));
                    }
                    if (next_x < sw) cover = false;
                }
                if (next_y < sh) cover = false;
                h->rs2_pitch_dw[l] = pdw;
                h->rs2_ok[l] = (prow <= 48 && pdw * prow <= 256 * RS2_TRIPS) ? 1 : 0;
                if (l == 1) h->rs2_copy_ok = cover && h->rs2_ok[1];
            }
            {   // streaming form: emit table over the source rows, strips x band pairs, and the conditions it needs
                static const bool no_stream = getenv("VIORB_RESIZE_TILES") != nullptr;           // debugging / tests: keep the tile forms
                bool ok = !no_stream && sw >= 8;
                h->rss_etab_off[l] = (int)h->rss_etab.size();
                std::vector<uint2> et(sh, make_uint2(0x7fffu, 0u));
                for (int dy = 0; dy < L.h && ok; dy++) {
                    const int sy0 = yt[dy].x & 0xffff, sy1 = yt[dy].x >> 16;
                    if ((sy1 != sy0 && sy1 != sy0 + 1) || et[sy1].x != 0x7fffu || dy >= 0x7fff) { ok = false; break; }
                    const int b0 = yt[dy].y & 0xffff, b1 = (yt[dy].y >> 16) & 0xffff;
                    if (b0 > 2048 || b1 > 2048) { ok = false; break; }
                    et[sy1] = make_uint2((uint32_t)dy | (sy1 == sy0 ? 0x8000u : 0u), (uint32_t)b0 | ((uint32_t)b1 << 16));
                }
                const int ndw = (L.w + 3) / 4;
                for (int dcol = 0; dcol < ndw && ok; dcol++) {                                 // the 8-byte source window of every output dword
                    const int first = xt[std::min(4 * dcol, L.w - 1)].x & 0xffff;
                    for (int j = 0; j < 4; j++) {
                        const int2 e = xt[std::min(4 * dcol + j, L.w - 1)];
                        if ((e.x >> 16) - first > 7 || (e.x & 0xffff) < first || (e.y & 0xffff) > 2048 || ((e.y >> 16) & 0xffff) > 2048) ok = false;
                    }
                }
                h->rss_item_off[l] = (int)h->rss_items.size();
                if (ok) {
                    const double sy = (double)sh / L.h;
                    const int rbmax = std::max(1, std::min(100, (int)((RSS_MAX_SRC_ROWS - 2) / sy)));
                    const int nb = 2 * ((L.h + 2 * rbmax - 1) / (2 * rbmax)), rb = (L.h + nb - 1) / nb;
                    for (int k = 0; k < nb && ok; k++) {                                        // every band's source rows fit the lane-held table
                        const int ya = k * rb, yb = std::min(ya + rb, L.h) - 1;
                        if (ya <= yb && (yt[yb].x >> 16) - (yt[ya].x & 0xffff) + 1 > RSS_MAX_SRC_ROWS) ok = false;
                    }
                    for (int sx = 0; sx < ndw && ok; sx += RSS_LANES)
                        for (int k = 0; k < nb; k += 2)
                            if (k * rb < L.h) h->rss_items.push_back(make_int4(sx, k * rb, rb, 0));
                }
                if (!ok) h->rss_items.resize(h->rss_item_off[l]);
                h->rss_ok[l] = ok ? 1 : 0;
                h->rss_etab.insert(h->rss_etab.end(), et.begin(), et.end());
            }
        }
        {   // blur work items: one wavefront = one 32-dword strip x two bands of rows (an even number of bands of <= 64 rows)
            const int nb = 2 * ((L.h + 127) / 128), rb = (L.h + nb - 1) / nb, ndw = (L.w + 3) / 4;
            for (int sx = 0; sx < ndw; sx += BS_LANES)
                for (int k = 0; k < nb; k += 2) h->blur_tiles.push_back(make_int4(l, sx, k * rb, rb));
            if (L.w < 12 || L.h < 4) { set_error("level %d (%dx%d) too small for the blur", l, L.w, L.h); return VIORB_ERR_UNSUPPORTED; }
        }
    }
    h->frame_bytes = align_up(off, 256);
    h->kp_pitch = kp_off;
    h->out_cap = max_keypoints_for_size(h, w, hgt);                       // == viorb_extractor_max_keypoints except on levels with more roots than a quarter of their quota
    const int ncells = (int)h->cells.size();
    if (ncells == 0) { set_error("image %dx%d too small for a FAST cell grid", w, hgt); return VIORB_ERR_UNSUPPORTED; }
    // FAST LDS: tile rows x pitch (dword aligned start => up to 3 extra bytes) + score map
    h->fast_tile_pitch = (int)align_up(max_cw + 3, 4) + 4;
    h->fast_tile_rows = max_ch;
    {   // LDS tile region: the largest cell tile (every cell has its own pitch), and at least the 64 * FAST_FETCH_TRIPS dwords the kernel stores unconditionally
        int tb = 64 * FAST_FETCH_TRIPS * 4;
        for (const CellDesc& c : h->cells) tb = std::max(tb, c.pitch * c.ch);
        h->fast_tile_bytes = (int)align_up((size_t)tb, 16);
    }
    h->fast_score_bytes = (int)align_up((size_t)(max_cw - 6 + 2) * (max_ch - 6 + 2), 4);
    h->slot_cap = ((max_cw - 6 + 1) / 2) * ((max_ch - 6 + 1) / 2);       // independent set of the king's graph
    h->fast_list_cap = (int)align_up((size_t)(max_cw - 6) * (max_ch - 6), 2);
    if (max_cw - 6 > 255 || max_ch - 6 > 255) { set_error("FAST cell larger than 255 px"); return VIORB_ERR_UNSUPPORTED; }
    {   // k_fast_cells3: tile of <= 13 dwords x 44 rows, score map of <= 44 x 40 bytes (every cell of the default 1.2-scale pyramids fits)
        bool ok = true;
        for (const CellDesc& c : h->cells) ok = ok && ((c.x0 & 3) + c.cw <= 4 * F3_TPD - 4) && c.ch <= 4 * F3_TRIPS && c.cw - 6 + 2 <= F3_SP && c.ch - 6 + 2 <= F3_SROWS;
        static const bool force_v2 = getenv("VIORB_FAST_V2") != nullptr;          // A/B switch: the round-2 kernel
        h->fast_v3 = ok && !force_v2;
        h->fast3_tile_bytes = (int)align_up((size_t)(4 * ((max_ch + 3) / 4)) * F3_TP, 16);          // whole trips of 4 rows
        h->fast3_score_bytes = (int)align_up((size_t)(max_ch - 6 + 2) * F3_SP + 16, 16);              // + the 16-byte granule of the clearing stores
    }
    // quadtree capacities
    int maxq = 1; for (int l = 0; l < nl; l++) maxq = std::max(maxq, h->quota[l]);
    h->oct_ncap = 8192;
    h->oct_nodecap = 5 * maxq + 64;
    int sc = 1; while (sc < maxq + 8) sc <<= 1;
    h->oct_sortcap = sc;
    size_t oct_lds = (size_t)h->oct_ncap * 8 + (size_t)h->oct_nodecap * sizeof(OctNode) + (size_t)sc * 4;
    if (h->oct_nodecap > 65535) { set_error("per-level quota %d too large for the quadtree's 16-bit node indices", maxq); return VIORB_ERR_UNSUPPORTED; }
    // A per-level quota above ~1100 (many features on one to three levels): the node list (5 x quota nodes) does not fit LDS beside the
    // candidates. Every level then takes the global-scratch instantiation with its nodes and sort keys in global memory as well —
    // slower (the serial phases walk global memory), but the same decisions and the same output; the reference has no such limit.
    h->oct_huge = oct_lds > 160 * 1024;
    if (!h->oct_huge && oct_lds > 64 * 1024 &&
        raise_dynamic_lds(reinterpret_cast<const void*>(k_octree<false>), oct_lds) != hipSuccess) {
        (void)hipGetLastError();
        h->oct_ncap = 4096;                              // stay inside the default 64 KiB dynamic-LDS window
        oct_lds = (size_t)h->oct_ncap * 8 + (size_t)h->oct_nodecap * sizeof(OctNode) + (size_t)sc * 4;
        if (oct_lds > 64 * 1024) { set_error("quadtree LDS (%zu B) exceeds the dynamic-LDS limit", oct_lds); return VIORB_ERR_UNSUPPORTED; }
    }

    if (!h->oct_huge) {                                  // the global-scratch instantiation keeps nodes and sort keys in LDS (its launch is part of every call)
        const size_t lds_big = (size_t)h->oct_sortcap * 8 + (size_t)h->oct_nodecap * sizeof(OctNodeBig);
        if (lds_big > 160 * 1024) h->oct_huge = true;
        else if (lds_big > 64 * 1024) VIORB_HIP_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(k_octree<true>), lds_big));
    }
    {   // the higher levels' own first quadtree launches (see launch_all): "level:candidates,level:candidates", levels ascending, candidates
        // descending and below OCT_NCAP_SMALL; VIORB_OCT_TIERS overrides ("0" = none)
        const char* e = getenv("VIORB_OCT_TIERS");
        std::string spec = e ? e : "3:2048";
        h->oct_tiers.clear();
        for (int l = 0; l < nl; l++) h->lv[l].oct_tier_cap = 0;
        int prev_l = 0, prev_c = std::min(h->oct_ncap, (int)OCT_NCAP_SMALL);
        h->oct_first_cap = prev_c;                                  // candidates of the common first launch ("0:candidates" lowers it)
        { const char* e2 = getenv("VIORB_OCT_NODE_X10"); h->oct_node_x10 = e2 ? atoi(e2) : 25; if (h->oct_node_x10 < 10 || h->oct_node_x10 > 50) h->oct_node_x10 = 50; }
        size_t pos = 0;
        while (!h->oct_huge && pos < spec.size()) {
            int l = 0, c = 0, used = 0;
            if (sscanf(spec.c_str() + pos, "%d:%d%n", &l, &c, &used) != 2) break;
            if (l == 0 && h->oct_tiers.empty() && c >= 64 && c <= prev_c) { h->oct_first_cap = c; prev_c = c + 1; }
            else if (l <= prev_l || l >= nl || c < 64 || c >= prev_c) break;
            else { h->oct_tiers.push_back(std::make_pair(l, c)); prev_l = l; prev_c = c; }
            pos += (size_t)used; if (pos < spec.size() && spec[pos] == ',') pos++;
        }
        for (size_t ti = 0; ti < h->oct_tiers.size(); ti++) {
            const int l1 = ti + 1 < h->oct_tiers.size() ? h->oct_tiers[ti + 1].first : nl;
            for (int l = h->oct_tiers[ti].first; l < l1; l++) h->lv[l].oct_tier_cap = h->oct_tiers[ti].second;
        }
    }
    const size_t B = (size_t)h->max_batch;
    VIORB_HIP_TRY(hipMalloc(&h->d_planes, B * h->frame_bytes));
    VIORB_HIP_TRY(hipMalloc(&h->d_blur, B * h->frame_bytes));
    VIORB_HIP_TRY(hipMalloc(&h->d_lv, sizeof(LevelDev) * nl));
    VIORB_HIP_TRY(hipMalloc(&h->d_cells, sizeof(CellDesc) * ncells));
    VIORB_HIP_TRY(hipMalloc(&h->d_blur_tiles, sizeof(int4) * h->blur_tiles.size()));
    VIORB_HIP_TRY(hipMalloc(&h->d_rs2_tiles, sizeof(int4) * std::max<size_t>(h->rs2_tiles.size(), 1)));
    VIORB_HIP_TRY(hipMalloc(&h->d_rss_items, sizeof(int4) * std::max<size_t>(h->rss_items.size(), 1)));
    VIORB_HIP_TRY(hipMalloc(&h->d_rss_etab, sizeof(uint2) * std::max<size_t>(h->rss_etab.size(), 1)));
    VIORB_HIP_TRY(hipMalloc(&h->d_xtab, sizeof(int2) * std::max<size_t>(xtab.size(), 1)));
    VIORB_HIP_TRY(hipMalloc(&h->d_ytab, sizeof(int2) * std::max<size_t>(ytab.size(), 1)));
    VIORB_HIP_TRY(hipMalloc(&h->d_slots, B * ncells * h->slot_cap * sizeof(uint32_t)));
    VIORB_HIP_TRY(hipMalloc(&h->d_cell_cnt, B * ncells * sizeof(int)));
    VIORB_HIP_TRY(hipMalloc(&h->d_lvl_kp, B * h->kp_pitch * sizeof(uint32_t)));
    VIORB_HIP_TRY(hipMalloc(&h->d_lvl_cnt, B * nl * sizeof(int)));
    VIORB_HIP_TRY(hipMalloc(&h->d_lvl_ncand, B * nl * sizeof(int)));
    VIORB_HIP_TRY(hipMalloc(&h->d_lvl_tot, B * nl * sizeof(int)));
    {   // scratch of the over-size quadtree levels: a level's candidates are bounded by its cells' slots (strict 3x3 NMS keeps at most every
        // other pixel of every other row); up to 64 such levels per call
        int big = 1;
        for (int l = 0; l < nl; l++) big = std::max(big, h->lv[l].ncells * h->slot_cap);
        h->oct_big_cap = (int)align_up((size_t)big, 64);
        h->oct_big_slots = h->oct_huge ? (int)(B * nl) : (int)std::min<size_t>(B * nl, 64);
        VIORB_HIP_TRY(hipMalloc(&h->d_oct_big, (size_t)h->oct_big_slots * 3 * h->oct_big_cap * sizeof(uint32_t)));
        VIORB_HIP_TRY(hipMalloc(&h->d_oct_big_next, sizeof(int)));
        if (h->oct_huge) {
            h->oct_node_bytes = align_up((size_t)h->oct_sortcap * 8 + (size_t)h->oct_nodecap * sizeof(OctNodeBig), 256);
            VIORB_HIP_TRY(hipMalloc(&h->d_oct_nodes, (size_t)h->oct_big_slots * h->oct_node_bytes));
        }
    }
    VIORB_HIP_TRY(hipMalloc(&h->d_kps, B * h->out_cap * sizeof(viorb_keypoint)));
    VIORB_HIP_TRY(hipMalloc(&h->d_desc, B * h->out_cap * 32));
    VIORB_HIP_TRY(hipMalloc(&h->d_count, B * sizeof(int)));
    VIORB_HIP_TRY(hipMalloc(&h->d_status, B * sizeof(int)));
    VIORB_HIP_TRY(hipMemset(h->d_planes, 0, B * h->frame_bytes));
    VIORB_HIP_TRY(hipMemset(h->d_blur, 0, B * h->frame_bytes));
    VIORB_HIP_TRY(hipMemset(h->d_count, 0, B * sizeof(int)));
    VIORB_HIP_TRY(hipMemset(h->d_status, 0, B * sizeof(int)));
    VIORB_HIP_TRY(hipMemset(h->d_lvl_cnt, 0, B * nl * sizeof(int)));
    VIORB_HIP_TRY(hipMemcpy(h->d_lv, h->lv.data(), sizeof(LevelDev) * nl, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(h->d_cells, h->cells.data(), sizeof(CellDesc) * ncells, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(h->d_blur_tiles, h->blur_tiles.data(), sizeof(int4) * h->blur_tiles.size(), hipMemcpyHostToDevice));
    if (!h->rs2_tiles.empty()) VIORB_HIP_TRY(hipMemcpy(h->d_rs2_tiles, h->rs2_tiles.data(), sizeof(int4) * h->rs2_tiles.size(), hipMemcpyHostToDevice));
    if (!h->rss_items.empty()) VIORB_HIP_TRY(hipMemcpy(h->d_rss_items, h->rss_items.data(), sizeof(int4) * h->rss_items.size(), hipMemcpyHostToDevice));
    if (!h->rss_etab.empty()) VIORB_HIP_TRY(hipMemcpy(h->d_rss_etab, h->rss_etab.data(), sizeof(uint2) * h->rss_etab.size(), hipMemcpyHostToDevice));
    if (!xtab.empty()) VIORB_HIP_TRY(hipMemcpy(h->d_xtab, xtab.data(), sizeof(int2) * xtab.size(), hipMemcpyHostToDevice));
    if (!ytab.empty()) VIORB_HIP_TRY(hipMemcpy(h->d_ytab, ytab.data(), sizeof(int2) * ytab.size(), hipMemcpyHostToDevice));
    if (!h->tables_uploaded) {
        int g[7];
        {   // getGaussianKernel(7, 2, CV_32F) -> convertTo(CV_32S, 256)  (OpenCV 2.4)
            float cf[7]; double sum = 0; const double s2 = -0.5 / (2.0 * 2.0);
            for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(s2 * x * x); sum += cf[i]; }
            sum = 1. / sum;
            for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); g[i] = host_cv_round((double)(cf[i] * 256.f)); }
        }
        uint32_t pat[256];
        for (int i = 0; i < 256; i++)
            pat[i] = (uint32_t)(uint8_t)kPatternHost[4 * i] | ((uint32_t)(uint8_t)kPatternHost[4 * i + 1] << 8) |
                     ((uint32_t)(uint8_t)kPatternHost[4 * i + 2] << 16) | ((uint32_t)(uint8_t)kPatternHost[4 * i + 3] << 24);
        VIORB_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), g, sizeof(g)));
        {
            uint32_t rc[64]; rc[0] = 0;
            for (int d = 1; d < 64; d++) rc[d] = (65536u + d - 1) / d;
            VIORB_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_rcp16), rc, sizeof(rc)));
        }
        VIORB_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), pat, sizeof(pat)));
        VIORB_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_umax), h->umax, sizeof(int) * 16));
        uint32_t om[256];
        for (int row = 0; row < 32; row++)
            for (int j = 0; j < 8; j++) {
                uint32_t m = 0;
                for (int k2 = 0; k2 < 4; k2++) {
                    const int u = -HALF_PATCH + 4 * j + k2, v = row - HALF_PATCH;
                    if (row < PATCH && u <= HALF_PATCH && std::abs(u) <= h->umax[std::abs(v)]) m |= 0xffu << (8 * k2);
                }
                om[row * 8 + j] = m;
            }
        VIORB_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_orient_mask), om, sizeof(om)));
        h->tables_uploaded = true;
    }
    h->img_w = w; h->img_h = hgt;
    return VIORB_OK;
}

// images per FAST launch (see launch_all)
static int fast_launch_images(int batch) {
    static const int nlaunch = getenv("VIORB_FAST_LAUNCHES") ? std::max(1, atoi(getenv("VIORB_FAST_LAUNCHES"))) : FAST_LAUNCHES;
    return std::min(std::max(batch, 1), std::max(8, ((batch + nlaunch - 1) / nlaunch + 7) & ~7));
}

static int launch_all(viorb_extractor* h, const uint8_t* d_images, int batch, int stride, size_t pitch, hipStream_t st) {
    const int nl = h->p.nlevels;
    const int ncells = (int)h->cells.size();
    const LevelDev& L0 = h->lv[0];
    // level 1's resize can also write the level-0 plane when its tiles cover the image and the caller's rows are dword-addressable
    // (small batches only: at 256 streams beside the tracking stream the fused form measured 117 k frames/s against 136 k with the
    // separate copy — the tracking chain's first kernels then start under the heaviest resize launch instead of under the light copy)
    const bool fuse_copy = batch < 16 && nl > 1 && h->rs2_copy_ok && (stride & 3) == 0 && (pitch & 3) == 0 && (((uintptr_t)d_images) & 3) == 0 && (L0.w & 3) == 0;
    if (!fuse_copy) {
        const int gx = (L0.stride / 16 + 63) / 64;
        const XcdPlace PL = make_place(gx * L0.h, batch);
        ProfScope ps("k_copy_level0", st);
        hipLaunchKernelGGL(k_copy_level0, dim3(place_blocks(PL)), dim3(64), 0, st, d_images, L0.w, L0.h, stride, pitch, h->d_planes, h->frame_bytes, L0.stride, h->d_status, PL, gx);
    }
    for (int l = 1; l < nl; l++) {
        const LevelDev& L = h->lv[l];
        if (h->rss_ok[l] && !(l == 1 && fuse_copy)) {
            const LevelDev& P = h->lv[l - 1];
            ResizeStreamArgs A;
            A.planes = h->d_planes; A.frame_bytes = h->frame_bytes; A.src_off = P.plane_off; A.dst_off = L.plane_off;
            A.src_stride = P.stride; A.src_h = P.h; A.dst_stride = L.stride; A.dst_w = L.w; A.dst_h = L.h;
            A.xtab = h->d_xtab + L.xtab_off; A.ytab = h->d_ytab + L.ytab_off; A.etab = h->d_rss_etab + h->rss_etab_off[l];
            A.items = h->d_rss_items + h->rss_item_off[l];
            const int nitems = (l + 1 < nl ? h->rss_item_off[l + 1] : (int)h->rss_items.size()) - h->rss_item_off[l];
            const XcdPlace PL = make_place(nitems, batch);
            ProfScope ps("k_resize", st);
            hipLaunchKernelGGL(k_resize_stream, dim3(place_blocks(PL)), dim3(64), 0, st, A, PL);
            continue;
        }
        if (h->rs2_ok[l]) {
            const LevelDev& P = h->lv[l - 1];
            Resize2Args A;
            const bool from_image = (l == 1 && fuse_copy);
            A.src = from_image ? d_images : h->d_planes + P.plane_off; A.src_pitch = from_image ? pitch : h->frame_bytes; A.src_stride = from_image ? stride : P.stride;
            A.dst = h->d_planes + L.plane_off; A.dst_pitch = h->frame_bytes; A.dst_stride = L.stride; A.dst_w = L.w; A.dst_h = L.h;
            A.copy_dst = from_image ? h->d_planes + L0.plane_off : nullptr; A.copy_pitch = h->frame_bytes; A.copy_stride = L0.stride;
            A.xtab = h->d_xtab + L.xtab_off; A.ytab = h->d_ytab + L.ytab_off; A.tiles = h->d_rs2_tiles + h->rs2_off[l];
            A.gx = (L.w + RS2_TW - 1) / RS2_TW; A.lds_pitch_dw = h->rs2_pitch_dw[l]; A.status = from_image ? h->d_status : nullptr;
            const int gy = (L.h + RS2_TH - 1) / RS2_TH;
            const XcdPlace PL = make_place(A.gx * gy, batch);
            const size_t lds = (size_t)A.lds_pitch_dw * 48 * 4 + (RS2_TW + RS2_TH) * sizeof(int2);
            ProfScope ps("k_resize", st);
            hipLaunchKernelGGL(k_resize2, dim3(place_blocks(PL)), dim3(256), lds, st, A, PL);
            continue;
        }
        const int gx = (L.w + RS_TW - 1) / RS_TW, gy = (L.h + RS_TH - 1) / RS_TH;
        const XcdPlace PL = make_place(gx * gy, batch);
        const size_t lds = (size_t)h->rs_pitch_dw[l] * h->rs_rows[l] * 4;
        ProfScope ps("k_resize", st);
        hipLaunchKernelGGL(k_resize, dim3(place_blocks(PL)), dim3(256), lds, st, h->d_planes, h->frame_bytes, h->d_lv, l, h->d_xtab, h->d_ytab,
                           h->rs_pitch_dw[l], h->rs_rows[l], PL, gx);
    }
    {
        const size_t lds = h->fast_v3 ? (size_t)h->fast3_tile_bytes + h->fast3_score_bytes + (size_t)(F3_SURV_CAP + F3_CORN_CAP) * 2
                                      : (size_t)h->fast_tile_bytes + h->fast_score_bytes + (size_t)h->fast_list_cap * 4;
        // FAST goes out as FAST_LAUNCHES launches over sub-ranges of the batch (multiples of 8 images). A launch boundary is the only point
        // where the tracking stream's large workgroups (a search: 16 waves + 73 KB of LDS; the pose solver: 240 registers per lane) can be
        // placed on a CU: while this kernel still has workgroups to hand out, every slot that frees goes to its next single-wave
        // workgroup, and SearchLocalPoints (54 us alone) finished only when FAST did (380 us). FAST's workgroups live a few
        // microseconds, so a sub-launch drains at once and the cost is a few launch gaps that the other stream fills: 118 k -> 139 k
        // frames/s at 256 streams (4 launches: 129 k, 16: 134 k, 32: 121 k). Splitting the quadtree, blur or descriptor kernels the same
        // way loses (their workgroups live long, every boundary is a tail).
        const int step = fast_launch_images(batch);
        h->fast_prof_rot++;
        for (int i0 = 0; i0 < batch; i0 += step) {
            // with a kernel selection the profiler times ONE sub-launch per call, a different one every call (an event pair costs ~8 us of stream
            // time; the sub-launches differ in what runs beside them, so the rotation is what makes the mean agree with a tracer's per-launch mean)
            ProfScope ps(prof_times_everything() || (i0 / step) == (int)(h->fast_prof_rot % (unsigned)((batch + step - 1) / step)) ? "k_fast_cells" : nullptr, st);
            const XcdPlace PL = make_place((ncells + FAST_CELLS_PER_WAVE - 1) / FAST_CELLS_PER_WAVE, batch, i0, std::min(batch, i0 + step));
            if (h->fast_v3)
                hipLaunchKernelGGL(k_fast_cells3, dim3(place_blocks(PL)), dim3(64), lds, st, h->d_planes, h->frame_bytes, h->d_cells,
                                   h->p.ini_th_fast, h->p.min_th_fast, h->d_slots, h->slot_cap, h->d_cell_cnt, ncells, h->fast3_tile_bytes, h->fast3_score_bytes, PL);
            else
                hipLaunchKernelGGL(k_fast_cells, dim3(place_blocks(PL)), dim3(64), lds, st, h->d_planes, h->frame_bytes, h->d_lv, h->d_cells,
                                   h->p.ini_th_fast, h->p.min_th_fast, h->d_slots, h->slot_cap, h->d_cell_cnt, ncells,
                                   h->fast_tile_bytes, h->fast_score_bytes, h->fast_list_cap, PL);
        }
    }
    // The blur only needs the pyramid, the quadtree only FAST: from here they run side by side, the blur on the handle's second stream.
    // The quadtree is one wavefront per (image, level) walking LDS lists — latency, 4 workgroups per CU by LDS, almost no vector issue —
    // and the streaming blur is pure vector issue without LDS, so the pair costs little more than the longer of the two.
    static const bool fork_blur = !(getenv("VIORB_BLUR_FORK") && atoi(getenv("VIORB_BLUR_FORK")) == 0);
    hipStream_t bst = st;
    if (fork_blur && batch >= 16) {
        if (!h->aux_stream) {
            VIORB_HIP_TRY(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
            VIORB_HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            VIORB_HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        bst = h->aux_stream;
        VIORB_HIP_TRY(hipEventRecord(h->ev_fork, st));
        VIORB_HIP_TRY(hipStreamWaitEvent(bst, h->ev_fork, 0));
    }
    {
        ProfScope ps("k_blur", bst);
        const XcdPlace PL = make_place((int)h->blur_tiles.size(), batch);
        hipLaunchKernelGGL(k_blur, dim3(place_blocks(PL)), dim3(64), 0, bst, h->d_planes, h->d_blur, h->frame_bytes,
                           h->d_lv, h->d_blur_tiles, PL);
    }
    if (bst != st) VIORB_HIP_TRY(hipEventRecord(h->ev_join, bst));
    {
        // common case first: <= OCT_NCAP_SMALL candidates per level fit a ~51 KB footprint (3 single-wave workgroups per CU); the second
        // launch covers the rest with the full capacity (its workgroups return at once when they have no work). Every launch lasts about as
        // long as its longest tree (a level 0: the launch is latency-bound, not slot-bound), so the first one is sized to take every level
        // of a feature-rich frame: with 2048 (4 per CU) the level-0 trees of EuRoC-lens frames (~2400 candidates) all fell to the second
        // launch and the pair took twice as long (DESIGN.md "Round 3 measurements")
        const size_t fixed = (size_t)h->oct_nodecap * sizeof(OctNode) + (size_t)h->oct_sortcap * 4;
        const int small = h->oct_first_cap > 0 ? h->oct_first_cap : std::min(h->oct_ncap, (int)OCT_NCAP_SMALL);
        // Beside the tracking stream's kernels the launch is slot-bound: 3 workgroups of 51 KB per CU (one beside two pose-solver workgroups),
        // every slot sized for level 0 (4096 candidates, 5 x quota(0) + 64 nodes) although candidates and quota fall with the level (lens
        // frames: 2404 2040 1710 1417 1200 972 757 560 candidates on levels 0..7). The higher levels therefore get their own first launches
        // (h->oct_tiers: first level, candidates) with the LDS plan of THEIR largest quota; what exceeds a level's plan goes to the full-capacity
        // launch like an over-size level 0 (LevelDev::oct_tier_cap). Alone the quadtree is not faster for it (latency of the longest tree),
        // in the step it is: 186.1 -> 189.4 k frames/s with {3: 2048}.
        const int ntier = h->oct_huge ? 0 : (int)h->oct_tiers.size();
        // Node slots of the first launches: 5 x quota + 64 can never run out (every node of a round expanding into four slots); a round in
        // bulk mode needs <= 4/3 quota and the one-at-a-time tail typically ends below 2.5 x quota, so the first launches are sized
        // oct_node_x10 / 10 x quota + 64 and a level that runs out of slots is left — before anything of it is written — to the
        // full-capacity launch (bit 30 of its candidate count)
        const bool defer = h->oct_node_x10 < 50;
        auto first_nodes = [&](int mq) { return std::min(5 * mq + 64, (h->oct_node_x10 * mq) / 10 + 64); };
        if (!h->oct_huge) {
            ProfScope ps("k_octree", st);
            const int l_end0 = ntier ? h->oct_tiers[0].first : nl;
            int mq0 = 1; for (int l = 0; l < l_end0; l++) mq0 = std::max(mq0, h->quota[l]);
            const int ndc0 = defer ? first_nodes(mq0) : h->oct_nodecap;
            hipLaunchKernelGGL(k_octree<false>, dim3(batch, l_end0), dim3(64), (size_t)small * 8 + (size_t)ndc0 * sizeof(OctNode) + (size_t)h->oct_sortcap * 4, st,
                               h->d_lv, h->d_slots, h->slot_cap, h->d_cell_cnt,
                               ncells, h->d_lvl_kp, h->kp_pitch, h->d_lvl_cnt, h->d_lvl_ncand, nl, h->d_status, small, ndc0,
                               h->oct_sortcap, -1, small, 0, 0, defer ? 1 : 0, nullptr, nullptr, 0, h->d_lvl_tot);
            for (int ti = 0; ti < ntier; ti++) {
                const int l0 = h->oct_tiers[ti].first, l1 = ti + 1 < ntier ? h->oct_tiers[ti + 1].first : nl, nc = h->oct_tiers[ti].second;
                int mq = 1; for (int l = l0; l < l1; l++) mq = std::max(mq, h->quota[l]);
                const int ndc = defer ? first_nodes(mq) : 5 * mq + 64;
                hipLaunchKernelGGL(k_octree<false>, dim3(batch, l1 - l0), dim3(64), (size_t)nc * 8 + (size_t)ndc * sizeof(OctNode) + (size_t)h->oct_sortcap * 4,
                                   st, h->d_lv, h->d_slots, h->slot_cap, h->d_cell_cnt, ncells, h->d_lvl_kp, h->kp_pitch, h->d_lvl_cnt, h->d_lvl_ncand, nl, h->d_status,
                                   nc, ndc, h->oct_sortcap, -1, nc, l0, 0, defer ? 1 : 0, nullptr, nullptr, 0, h->d_lvl_tot);
            }
        }
        if (!h->oct_huge && (small < h->oct_ncap || ntier || defer)) {
            ProfScope ps("k_octree_large", st);
            hipLaunchKernelGGL(k_octree<false>, dim3(batch, nl), dim3(64), (size_t)h->oct_ncap * 8 + fixed, st, h->d_lv, h->d_slots, h->slot_cap,
                               h->d_cell_cnt, ncells, h->d_lvl_kp, h->kp_pitch, h->d_lvl_cnt, h->d_lvl_ncand, nl, h->d_status, h->oct_ncap,
                               h->oct_nodecap, h->oct_sortcap, small, h->oct_ncap, 0, ntier ? 1 : 0, 0, nullptr, nullptr, 0, h->d_lvl_tot);
        }
        {   // levels with more candidates than the LDS form holds (none on camera images; their workgroups read one count and return)
            const size_t lds_big = h->oct_huge ? 0 : (size_t)h->oct_sortcap * 8 + (size_t)h->oct_nodecap * sizeof(OctNodeBig);
            VIORB_HIP_TRY(hipMemsetAsync(h->d_oct_big_next, 0, sizeof(int), st));
            ProfScope ps("k_octree_big", st);
            // huge quota: this launch takes EVERY level (n_above = -1: it counts the candidates itself), one scratch slot per (image, level)
            hipLaunchKernelGGL(k_octree<true>, dim3(batch, nl), dim3(64), lds_big, st, h->d_lv, h->d_slots, h->slot_cap, h->d_cell_cnt, ncells, h->d_lvl_kp,
                               h->kp_pitch, h->d_lvl_cnt, h->d_lvl_ncand, nl, h->d_status, h->oct_big_cap, h->oct_nodecap, h->oct_sortcap,
                               h->oct_huge ? -1 : std::max(h->oct_ncap, (int)small), 0x7fffffff, 0, 0, 0, h->d_oct_big, h->d_oct_big_next, h->oct_big_slots, h->d_lvl_tot,
                               h->oct_huge ? h->d_oct_nodes : (uint8_t*)nullptr, h->oct_node_bytes);
        }
    }
    if (bst != st) VIORB_HIP_TRY(hipStreamWaitEvent(st, h->ev_join, 0));
    {
        ProfScope ps("k_orient_describe", st);
        const XcdPlace PL = make_place((h->out_cap + 3) / 4, batch);
        hipLaunchKernelGGL(k_orient_describe, dim3(place_blocks(PL)), dim3(256), 0, st, h->d_planes, h->d_blur, h->frame_bytes,
                           h->d_lv, nl, h->d_lvl_kp, h->kp_pitch, h->d_lvl_cnt, h->d_kps, h->d_desc, h->out_cap, h->d_count, h->d_status, PL);
    }
    VIORB_HIP_TRY(hipGetLastError());
    h->last_stream = st; h->last_batch = batch;
    return VIORB_OK;
}

extern "C" {

int viorb_abi_version(void) { return 2; }   // 2: viorb_frontend_config.dist_coef

int viorb_memcpy_dtod_async(void* dst, const void* src, size_t bytes, void* stream) {
    VIORB_REQUIRE(dst && src, "null pointer");
    VIORB_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return VIORB_OK;
}

int viorb_profile_enable(int on) {
    g_prof.enabled = on != 0;
    return VIORB_OK;
}
int viorb_profile_reset(void) {
    g_prof.used = 0;
    return VIORB_OK;
}
int viorb_profile_select(const char* kernel_name) {
    g_prof.only = kernel_name ? kernel_name : "";
    return VIORB_OK;
}
// Synchronises the device and sums the recorded intervals per kernel name. names_buf receives the
// names separated by '\n'.
int viorb_profile_read(char* names_buf, int names_cap, double* total_ms, int* calls, int cap, int* n) {
    VIORB_REQUIRE(names_buf && total_ms && calls && n, "null argument");
    VIORB_HIP_TRY(hipDeviceSynchronize());
    const int k = (int)g_prof.names.size();
    *n = k;
    std::string all;
    for (int i = 0; i < k; i++) { all += g_prof.names[i]; all += '\n'; }
    snprintf(names_buf, names_cap, "%s", all.c_str());
    for (int i = 0; i < k && i < cap; i++) { total_ms[i] = 0; calls[i] = 0; }
    for (size_t r = 0; r < g_prof.used; r++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, g_prof.recs[r].a, g_prof.recs[r].b) != hipSuccess) continue;
        const int slot = g_prof.recs[r].slot;
        if (slot >= 0 && slot < cap) { total_ms[slot] += ms; calls[slot]++; }
    }
    return VIORB_OK;
}
// Start / end of every recorded interval in milliseconds since the first record (the records of both streams share one clock),
// in recording order; slot[i] indexes the names of viorb_profile_read. A timeline without a tracer's per-launch host cost.
int viorb_profile_timeline(double* start_ms, double* end_ms, int* slot, int cap, int* n) {
    VIORB_REQUIRE(start_ms && end_ms && slot && n, "null argument");
    VIORB_HIP_TRY(hipDeviceSynchronize());
    *n = (int)g_prof.used;
    for (size_t r = 0; r < g_prof.used && (int)r < cap; r++) {
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, g_prof.recs[0].a, g_prof.recs[r].a);
        (void)hipEventElapsedTime(&b, g_prof.recs[0].a, g_prof.recs[r].b);
        start_ms[r] = a; end_ms[r] = b; slot[r] = g_prof.recs[r].slot;
    }
    return VIORB_OK;
}
const char* viorb_last_error(void) { return viorb::last_error_buf(); }
int viorb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int viorb_extractor_create(const viorb_extractor_params* params, int max_batch, int device, viorb_extractor** out) {
    VIORB_REQUIRE(params && out, "null params/out");
    VIORB_REQUIRE(params->nlevels >= 1 && params->nlevels <= MAX_LEVELS, "nlevels must be 1..16");
    VIORB_REQUIRE(params->nfeatures >= 1 && params->scale_factor > 1.0f, "nfeatures >= 1, scale_factor > 1");
    VIORB_REQUIRE(max_batch >= 1, "max_batch >= 1");
    viorb_extractor* h = new viorb_extractor();
    h->p = *params; h->max_batch = max_batch; h->device = device;
    const int nl = params->nlevels;
    // reference :415-445 (scaleFactor is held in a double member there; float*double -> float)
    const double sf = (double)params->scale_factor;
    h->scale.resize(nl); h->sigma2.resize(nl); h->inv_scale.resize(nl); h->inv_sigma2.resize(nl); h->quota.resize(nl);
    h->scale[0] = 1.f; h->sigma2[0] = 1.f;
    for (int i = 1; i < nl; i++) { h->scale[i] = (float)(h->scale[i - 1] * sf); h->sigma2[i] = h->scale[i] * h->scale[i]; }
    for (int i = 0; i < nl; i++) { h->inv_scale[i] = 1.0f / h->scale[i]; h->inv_sigma2[i] = 1.0f / h->sigma2[i]; }
    const float factor = (float)(1.0f / sf);
    float nd = params->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) { h->quota[l] = host_cv_round(nd); sum += h->quota[l]; nd *= factor; }
    h->quota[nl - 1] = std::max(params->nfeatures - sum, 0);
    // umax (reference :452-469)
    {
        int v, v0, vmax = host_cv_floor(HALF_PATCH * sqrtf(2.f) / 2 + 1), vmin = host_cv_ceil(HALF_PATCH * sqrtf(2.f) / 2);
        const double hp2 = HALF_PATCH * HALF_PATCH;
        for (v = 0; v < 16; v++) h->umax[v] = 0;
        for (v = 0; v <= vmax; ++v) h->umax[v] = host_cv_round(sqrt(hp2 - v * v));
        for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) { while (h->umax[v0] == h->umax[v0 + 1]) ++v0; h->umax[v] = v0; ++v0; }
    }
    *out = h;
    return VIORB_OK;
}

int viorb_extractor_destroy(viorb_extractor* h) {
    if (!h) return VIORB_OK;
    if (h->d_planes) { (void)hipSetDevice(h->device); free_device(h); }
    if (h->d_stereo_work) (void)hipFree(h->d_stereo_work);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
    return VIORB_OK;
}

int viorb_extractor_tables(const viorb_extractor* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                           int32_t* fpl) {
    VIORB_REQUIRE(h, "null handle");
    for (int i = 0; i < h->p.nlevels; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->inv_scale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
        if (fpl) fpl[i] = h->quota[i];
    }
    return VIORB_OK;
}

int viorb_extractor_fast_launch_images(int batch) { return fast_launch_images(batch); }

int viorb_extractor_max_keypoints(const viorb_extractor* h, int* cap) {
    VIORB_REQUIRE(h && cap, "null handle/cap");
    int c = 0; for (int l = 0; l < h->p.nlevels; l++) c += h->quota[l] + 2;
    *cap = c;
    return VIORB_OK;
}

int viorb_extractor_max_keypoints_for(const viorb_extractor* h, int width, int height, int* cap) {
    VIORB_REQUIRE(h && cap && width > 0 && height > 0, "null handle/cap or empty size");
    *cap = max_keypoints_for_size(h, width, height);
    return VIORB_OK;
}

static int ensure_device(viorb_extractor* h, int w, int hgt) {
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    VIORB_HIP_TRY(hipSetDevice(h->device));
    if (w != h->img_w || hgt != h->img_h || !h->d_planes) {
        int rc = configure(h, w, hgt);
        if (rc != VIORB_OK) { h->img_w = h->img_h = 0; return rc; }
    }
    return VIORB_OK;
}

int viorb_extract_batch_device(viorb_extractor* h, const uint8_t* d_images, int batch, int width, int height, int stride,
                               size_t pitch, void* stream) {
    VIORB_REQUIRE(h && d_images, "null handle/images");
    VIORB_REQUIRE(batch >= 1 && batch <= h->max_batch, "batch out of range");
    VIORB_REQUIRE(width > 0 && height > 0 && stride >= width, "bad image geometry");
    int rc = ensure_device(h, width, height);
    if (rc != VIORB_OK) return rc;
    return launch_all(h, d_images, batch, stride, pitch, (hipStream_t)stream);
}

int viorb_extractor_results_device(const viorb_extractor* h, const viorb_keypoint** d_kps, const uint8_t** d_desc,
                                   const int32_t** d_count, const int32_t** d_status, int* cap) {
    VIORB_REQUIRE(h && h->d_kps, "no results yet");
    if (d_kps) *d_kps = h->d_kps;
    if (d_desc) *d_desc = h->d_desc;
    if (d_count) *d_count = h->d_count;
    if (d_status) *d_status = h->d_status;
    if (cap) *cap = h->out_cap;
    return VIORB_OK;
}

int viorb_extractor_download(viorb_extractor* h, int b, viorb_keypoint* kps, uint8_t* desc, int cap, int* n) {
    VIORB_REQUIRE(h && h->d_kps && n, "no results yet");
    VIORB_REQUIRE(b >= 0 && b < h->last_batch, "image index out of range");
    VIORB_HIP_TRY(hipSetDevice(h->device));
    VIORB_HIP_TRY(hipStreamSynchronize(h->last_stream));
    int cnt = 0, st = 0;
    VIORB_HIP_TRY(hipMemcpy(&cnt, h->d_count + b, sizeof(int), hipMemcpyDeviceToHost));
    VIORB_HIP_TRY(hipMemcpy(&st, h->d_status + b, sizeof(int), hipMemcpyDeviceToHost));
    *n = cnt;
    const int m = std::min(cnt, cap);
    if (m > 0 && kps) VIORB_HIP_TRY(hipMemcpy(kps, h->d_kps + (size_t)b * h->out_cap, sizeof(viorb_keypoint) * m, hipMemcpyDeviceToHost));
    if (m > 0 && desc) VIORB_HIP_TRY(hipMemcpy(desc, h->d_desc + (size_t)b * h->out_cap * 32, (size_t)32 * m, hipMemcpyDeviceToHost));
    if (st != VIORB_OK) { set_error("image %d: a capacity of the extractor was exceeded (quadtree nodes / scratch slots, or more keypoints kept than viorb_extractor_max_keypoints: a level whose quadtree roots outnumber a quarter of its quota)", b); return st; }
    if (cnt > cap) { set_error("caller capacity %d < %d keypoints", cap, cnt); return VIORB_ERR_CAPACITY; }
    return VIORB_OK;
}

int viorb_extract(viorb_extractor* h, const uint8_t* img, int width, int height, int stride, viorb_keypoint* kps,
                  uint8_t* desc, int cap, int* n) {
    VIORB_REQUIRE(h && n, "null handle/n");
    *n = 0;
    if (!img || width <= 0 || height <= 0) return VIORB_OK;       // reference :1046-1047
    VIORB_REQUIRE(stride >= width, "stride < width");
    int rc = ensure_device(h, width, height);
    if (rc != VIORB_OK) return rc;
    if (!h->own_stream) VIORB_HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    const size_t bytes = (size_t)stride * height;
    if (h->stage_bytes < bytes) {
        if (h->d_stage) (void)hipFree(h->d_stage);
        h->d_stage = nullptr; h->stage_bytes = 0;
        VIORB_HIP_TRY(hipMalloc(&h->d_stage, bytes));
        h->stage_bytes = bytes;
    }
    VIORB_HIP_TRY(hipMemcpyAsync(h->d_stage, img, bytes, hipMemcpyHostToDevice, h->own_stream));
    rc = launch_all(h, h->d_stage, 1, stride, bytes, h->own_stream);
    if (rc != VIORB_OK) return rc;
    return viorb_extractor_download(h, 0, kps, desc, cap, n);
}

int viorb_extractor_level_device(const viorb_extractor* h, int b, int level, int blurred, const uint8_t** d_ptr, int* width,
                                 int* height, int* stride) {
    VIORB_REQUIRE(h && h->d_planes, "no pyramid yet");
    VIORB_REQUIRE(level >= 0 && level < h->p.nlevels && b >= 0 && b < h->max_batch, "level/image out of range");
    const LevelDev& L = h->lv[level];
    if (d_ptr) *d_ptr = (blurred ? h->d_blur : h->d_planes) + (size_t)b * h->frame_bytes + L.plane_off;
    if (width) *width = L.w;
    if (height) *height = L.h;
    if (stride) *stride = L.stride;
    return VIORB_OK;
}

int viorb_extractor_level_download(viorb_extractor* h, int b, int level, int blurred, uint8_t* dst, int* width, int* height) {
    const uint8_t* p = nullptr; int w = 0, hh = 0, s = 0;
    int rc = viorb_extractor_level_device(h, b, level, blurred, &p, &w, &hh, &s);
    if (rc != VIORB_OK) return rc;
    VIORB_HIP_TRY(hipSetDevice(h->device));
    VIORB_HIP_TRY(hipStreamSynchronize(h->last_stream));
    if (dst) VIORB_HIP_TRY(hipMemcpy2D(dst, w, p, s, w, hh, hipMemcpyDeviceToHost));
    if (width) *width = w;
    if (height) *height = hh;
    return VIORB_OK;
}

int viorb_stereo_match_device(const viorb_extractor* L, int left_index, const viorb_extractor* R, int right_index, int pairs, float bf, float fx,
                              float* d_uright, float* d_depth, int32_t* d_nmatched, void* stream) {
    VIORB_REQUIRE(L && R && L->d_kps && R->d_kps && d_uright && d_depth && d_nmatched, "extract both images first");
    VIORB_REQUIRE(pairs >= 1 && left_index >= 0 && right_index >= 0 && left_index + pairs <= L->max_batch && right_index + pairs <= R->max_batch,
                  "image indices out of range");
    VIORB_REQUIRE(L->img_w == R->img_w && L->img_h == R->img_h && L->out_cap == R->out_cap && L->p.nlevels == R->p.nlevels && L->device == R->device,
                  "left and right extractors must have the same geometry and parameters");
    VIORB_HIP_TRY(hipSetDevice(L->device));
    StereoArgs A;
    const int cap = L->out_cap;
    A.kl = L->d_kps + (size_t)left_index * cap; A.kr = R->d_kps + (size_t)right_index * cap;
    A.dl = L->d_desc + (size_t)left_index * cap * 32; A.dr = R->d_desc + (size_t)right_index * cap * 32;
    A.nl = L->d_count + left_index; A.nr = R->d_count + right_index;
    A.planesL = L->d_planes + (size_t)left_index * L->frame_bytes; A.planesR = R->d_planes + (size_t)right_index * R->frame_bytes;
    A.frame_bytes = L->frame_bytes; A.lv = L->d_lv; A.cap = cap; A.nlevels = L->p.nlevels; A.bf = bf; A.fx = fx;
    int sn = 64; while (sn < cap) sn <<= 1;
    A.sort_n = sn;
    for (int i = 0; i < 16; i++) { A.scale[i] = L->scale[i < L->p.nlevels ? i : L->p.nlevels - 1]; A.inv_scale[i] = L->inv_scale[i < L->p.nlevels ? i : L->p.nlevels - 1]; }
    A.uright = d_uright; A.depth = d_depth; A.nmatched = d_nmatched;
    const size_t lds = stereo_lds_bytes(cap, sn);
    A.work = nullptr; A.work_bytes = 0;
    ProfScope ps("k_stereo_match", (hipStream_t)stream);
    if (lds > 160 * 1024) {                                            // more features per image (> ~4000) than the sort / SAD arrays fit in LDS
        viorb_extractor* Lm = const_cast<viorb_extractor*>(L);
        const size_t per = (lds + 255) & ~(size_t)255;
        if (Lm->stereo_work_bytes < per * pairs) {
            // the scratch is the handle's: over-size stereo associations of one handle run on one stream at a time (include/viorb.h); a whole-
            // device synchronisation before it is replaced, so that no kernel of another stream still walks the old allocation
            VIORB_HIP_TRY(hipSetDevice(L->device));
            VIORB_HIP_TRY(hipDeviceSynchronize());
            if (Lm->d_stereo_work) (void)hipFree(Lm->d_stereo_work);
            Lm->d_stereo_work = nullptr; Lm->stereo_work_bytes = 0;
            VIORB_HIP_TRY(hipMalloc(&Lm->d_stereo_work, per * pairs));
            Lm->stereo_work_bytes = per * pairs;
        }
        A.work = Lm->d_stereo_work; A.work_bytes = per;
        hipLaunchKernelGGL(k_stereo_match<true>, dim3(pairs), dim3(1024), 0, (hipStream_t)stream, A);
    } else {
        if (lds > 64 * 1024) VIORB_HIP_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(k_stereo_match<false>), lds));
        hipLaunchKernelGGL(k_stereo_match<false>, dim3(pairs), dim3(1024), lds, (hipStream_t)stream, A);
    }
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_stereo_match(viorb_extractor* L, viorb_extractor* R, float bf, float fx, float* uright, float* depth, int cap, int* nmatched) {
    VIORB_REQUIRE(L && R && uright && depth && nmatched, "null argument");
    VIORB_REQUIRE(L->d_kps && R->d_kps, "extract both images first");
    VIORB_HIP_TRY(hipSetDevice(L->device));
    VIORB_HIP_TRY(hipStreamSynchronize(L->last_stream));
    VIORB_HIP_TRY(hipStreamSynchronize(R->last_stream));
    float *d_u = nullptr, *d_d = nullptr; int* d_n = nullptr;
    const int oc = L->out_cap;
    VIORB_HIP_TRY(hipMalloc(&d_u, sizeof(float) * oc)); VIORB_HIP_TRY(hipMalloc(&d_d, sizeof(float) * oc)); VIORB_HIP_TRY(hipMalloc(&d_n, sizeof(int)));
    int rc = viorb_stereo_match_device(L, 0, R, 0, 1, bf, fx, d_u, d_d, d_n, nullptr);
    if (rc == VIORB_OK && hipDeviceSynchronize() != hipSuccess) { set_error("stereo kernel failed"); rc = VIORB_ERR_HIP; }
    if (rc == VIORB_OK) {
        const int m = std::min(cap, oc);
        (void)hipMemcpy(uright, d_u, sizeof(float) * m, hipMemcpyDeviceToHost);
        (void)hipMemcpy(depth, d_d, sizeof(float) * m, hipMemcpyDeviceToHost);
        (void)hipMemcpy(nmatched, d_n, sizeof(int), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_u); (void)hipFree(d_d); (void)hipFree(d_n);
    return rc;
}

int viorb_extractor_debug_level_points(viorb_extractor* h, int b, int level, int which, int32_t* xyr, int cap, int* n) {
    VIORB_REQUIRE(h && h->d_planes && n, "no results yet");
    VIORB_REQUIRE(level >= 0 && level < h->p.nlevels && b >= 0 && b < h->last_batch, "level/image out of range");
    VIORB_HIP_TRY(hipSetDevice(h->device));
    VIORB_HIP_TRY(hipStreamSynchronize(h->last_stream));
    const LevelDev& L = h->lv[level];
    const int ncells = (int)h->cells.size();
    std::vector<uint32_t> keys;
    if (which == 0) {
        std::vector<int> cc(L.ncells);
        std::vector<uint32_t> sl((size_t)L.ncells * h->slot_cap);
        if (L.ncells > 0) {
            VIORB_HIP_TRY(hipMemcpy(cc.data(), h->d_cell_cnt + (size_t)b * ncells + L.cell_base, sizeof(int) * L.ncells, hipMemcpyDeviceToHost));
            VIORB_HIP_TRY(hipMemcpy(sl.data(), h->d_slots + ((size_t)b * ncells + L.cell_base) * h->slot_cap, sl.size() * 4, hipMemcpyDeviceToHost));
        }
        for (int c = 0; c < L.ncells; c++)
            for (int k = 0; k < cc[c]; k++) keys.push_back(sl[(size_t)c * h->slot_cap + k]);
    } else {
        int cnt = 0;
        VIORB_HIP_TRY(hipMemcpy(&cnt, h->d_lvl_cnt + b * h->p.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
        keys.resize(cnt);
        if (cnt > 0) VIORB_HIP_TRY(hipMemcpy(keys.data(), h->d_lvl_kp + (size_t)b * h->kp_pitch + L.kp_off, sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost));
    }
    *n = (int)keys.size();
    for (int i = 0; i < (int)keys.size() && i < cap; i++) {
        xyr[3 * i] = (int)(keys[i] & 0xfff); xyr[3 * i + 1] = (int)((keys[i] >> 12) & 0xfff); xyr[3 * i + 2] = (int)(keys[i] >> 24);
    }
    return VIORB_OK;
}

// Host-only test hooks (no GPU needed): the array formulation of the quadtree and the scalar math
// used by the kernels, so the CPU test-suite can compare them with the oracle.
int viorb_debug_octree_host(const uint32_t* keys, int n, int width, int height, int N, uint32_t* out, int cap, int* nout) {
    VIORB_REQUIRE(keys && out && nout && n >= 0 && n <= 65535, "bad arguments");
    std::vector<uint32_t> k(keys, keys + n);
    std::vector<uint32_t> r = distribute_octree_arrays(k, width, height, N);
    *nout = (int)r.size();
    for (int i = 0; i < (int)r.size() && i < cap; i++) out[i] = r[i];
    return VIORB_OK;
}
float viorb_debug_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
void viorb_debug_sincos(float r, float* s, float* c) { sincos_f32(r, s, c); }

} // extern "C"
