// viorb_amd/csrc/bow_matcher.hip — the bag-of-words side of the matcher (SURVEY.md §8 a13 and "next" rank 2):
//   k_bow_transform    DBoW2 TemplatedVocabulary::transform, the per-descriptor vocabulary-tree descent
//                      (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1231-1272, FORB.cpp:80-101) that
//                      Frame::ComputeBoW / KeyFrame::ComputeBoW run with levelsup = 4 (src/Frame.cc:575-582)
//   k_search_by_bow    ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:159-288)
// Integer work, bit-exact against the oracle. The tree descent is a dependent chain of L (= 6) steps of k (= 10) random
// 32-byte reads: 16 lanes share a descriptor and take one child each, so a step is one gather + one 16-lane min.
// SearchByBoW is greedy inside a vocabulary node but nodes are disjoint: one wave owns a node at a time and walks its
// key-frame features in index order, the 64 lanes scanning that node's frame features.
#include <hip/hip_runtime.h>
#include <vector>
#include <algorithm>
#include "viorb_common.h"
#include "orb_math.h"

namespace viorb {

#define BOW_HISTO 30
#define BOW_TH_LOW 50

struct BowVoc { const int* child_start; const int* child_ids; const uint8_t* desc; const int* word_id; const double* weight; int n_nodes, L; };

__global__ __launch_bounds__(256) void k_bow_transform(BowVoc V, const uint8_t* __restrict__ desc, const int* __restrict__ count, int cap, int batch,
                                                       int levelsup, int* __restrict__ word, double* __restrict__ weight, int* __restrict__ node) {
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    const int b = g / cap, i = g - b * cap;
    const bool live = b < batch && i < count[b];
    // (all 16 lanes of a group take the same path; dead groups still run the shuffles below with harmless values)
    uint32_t f[8];
    const uint4* fp = reinterpret_cast<const uint4*>(desc + (size_t)32 * (live ? g : 0));
    const uint4 f0 = fp[0], f1 = fp[1];
    f[0] = f0.x; f[1] = f0.y; f[2] = f0.z; f[3] = f0.w; f[4] = f1.x; f[5] = f1.y; f[6] = f1.z; f[7] = f1.w;
    const int nid_level = V.L - levelsup;
    int cur = 0, nid = 0;
    for (int lvl = 1; lvl <= 64; lvl++) {                     // a well-formed tree ends at lvl == L; 64 bounds a malformed one
        const int c0 = V.child_start[cur], c1 = V.child_start[cur + 1];
        if (c1 <= c0) break;
        uint32_t best = 0xFFFFFFFFu;                           // (distance << 20) | position: min = first minimum, as 'd < best_d'
        for (int p = c0 + sub; p < c1; p += 16) {
            const int id = V.child_ids[p];
            const uint4* cp = reinterpret_cast<const uint4*>(V.desc + (size_t)32 * id);
            const uint4 a = cp[0], c = cp[1];
            const int d = __popc(f[0] ^ a.x) + __popc(f[1] ^ a.y) + __popc(f[2] ^ a.z) + __popc(f[3] ^ a.w) +
                          __popc(f[4] ^ c.x) + __popc(f[5] ^ c.y) + __popc(f[6] ^ c.z) + __popc(f[7] ^ c.w);
            const uint32_t key = ((uint32_t)d << 20) | (uint32_t)(p - c0);
            best = key < best ? key : best;
        }
#pragma unroll
        for (int s = 8; s > 0; s >>= 1) { const uint32_t o = __shfl_xor(best, s, 16); best = o < best ? o : best; }
        cur = V.child_ids[c0 + (int)(best & 0xFFFFFu)];
        if (lvl == nid_level) nid = cur;
    }
    if (live && sub == 0) { word[g] = V.word_id[cur]; weight[g] = V.weight[cur]; node[g] = nid; }
}

struct BowSearchArgs {
    const viorb_keypoint* kf_kps; const uint8_t* kf_desc; const int* kf_node; const uint8_t* kf_has_point; const int* kf_count;
    const viorb_keypoint* f_kps; const uint8_t* f_desc; const int* f_node; const int* f_count;
    int* match; int* nmatches;
    int cap; float nnratio; int check_ori;
    unsigned char* work; size_t work_bytes;      // k_search_by_bow<true>: the work arrays in global memory (more than ~7100 keypoints)
};
__host__ __device__ inline size_t bow_search_lds_bytes(int cap) { return (size_t)cap * (4 + 4 + 4 + 2 + 4 * 2 + 1) + 256; }

template <bool GW>
__global__ __launch_bounds__(256) void k_search_by_bow(BowSearchArgs A) {
    extern __shared__ __align__(16) unsigned char smem_lds[];
    unsigned char* smem = GW ? A.work + (size_t)blockIdx.x * A.work_bytes : smem_lds;
    const int cap = A.cap, b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int* s_knode = reinterpret_cast<int*>(smem);                       // [cap]
    int* s_fnode = s_knode + cap;                                      // [cap]
    int* s_match = s_fnode + cap;                                      // [cap] frame feature -> key-frame feature
    unsigned short* s_leader = reinterpret_cast<unsigned short*>(s_match + cap);   // [cap] first key-frame feature of each node
    unsigned short* s_list = s_leader + cap;                           // [4][cap] frame features of the node a wave works on
    unsigned char* s_bin = reinterpret_cast<unsigned char*>(s_list + 4 * cap);     // [cap]
    __shared__ int s_nlead, s_hist[BOW_HISTO], s_keep[BOW_HISTO], s_nm;
    const int nK = min(A.kf_count[b], cap), nF = min(A.f_count[b], cap);
    const size_t o = (size_t)b * cap;
    for (int i = t; i < cap; i += blockDim.x) { s_knode[i] = i < nK ? A.kf_node[o + i] : -1; s_fnode[i] = i < nF ? A.f_node[o + i] : -1; s_match[i] = -1; s_bin[i] = 0; }
    if (t < BOW_HISTO) { s_hist[t] = 0; s_keep[t] = 1; }
    if (t == 0) { s_nlead = 0; s_nm = 0; }
    __syncthreads();
    for (int i = t; i < nK; i += blockDim.x) {
        const int nd = s_knode[i];
        bool lead = nd >= 0;
        for (int j = 0; j < i && lead; j++) lead = s_knode[j] != nd;
        if (lead) s_leader[atomicAdd(&s_nlead, 1)] = (unsigned short)i;
    }
    __syncthreads();
    const int nlead = s_nlead;
    unsigned short* list = s_list + (size_t)wv * cap;
    const float factor = 1.0f / BOW_HISTO;
    for (int g = wv; g < nlead; g += 4) {
        const int i0 = s_leader[g], nd = s_knode[i0];
        // frame features of this node, ascending (the order of FeatureVector's index vector)
        int nl = 0;
        for (int base = 0; base < nF; base += 64) {
            const int j = base + lane;
            const bool in = j < nF && s_fnode[j] == nd;
            const unsigned long long m = __ballot(in);
            if (in) list[nl + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)j;
            nl += __popcll(m);
        }
        if (nl == 0) continue;                                          // node absent from the frame: lower_bound skips it
        __builtin_amdgcn_wave_barrier();
        for (int kbase = (i0 & ~63); kbase < nK; kbase += 64) {
            const int kj = kbase + lane;
            unsigned long long members = __ballot(kj >= i0 && kj < nK && s_knode[kj] == nd);
            while (members) {
                const int k = kbase + __ffsll((long long)members) - 1;
                members &= members - 1;
                if (!A.kf_has_point[o + k]) continue;
                const uint4* kp = reinterpret_cast<const uint4*>(A.kf_desc + (o + k) * 32);
                const uint4 k0 = kp[0], k1 = kp[1];
                uint32_t best = (256u << 16) | 0xFFFFu; int d2 = 256;
                for (int c = lane; c < nl; c += 64) {
                    const int idx = list[c];
                    if (s_match[idx] >= 0) continue;
                    const uint4* fp = reinterpret_cast<const uint4*>(A.f_desc + (o + idx) * 32);
                    const uint4 a = fp[0], e = fp[1];
                    const int d = __popc(k0.x ^ a.x) + __popc(k0.y ^ a.y) + __popc(k0.z ^ a.z) + __popc(k0.w ^ a.w) +
                                  __popc(k1.x ^ e.x) + __popc(k1.y ^ e.y) + __popc(k1.z ^ e.z) + __popc(k1.w ^ e.w);
                    const uint32_t key = ((uint32_t)d << 16) | (uint32_t)c;
                    if (key < best) { d2 = (int)(best >> 16); best = key; } else if (d < d2) d2 = d;
                }
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) {
                    const uint32_t ob = __shfl_xor(best, s); const int od2 = __shfl_xor(d2, s);
                    const uint32_t lo = ob < best ? ob : best, hi = ob < best ? best : ob;
                    best = lo; d2 = min(min(d2, od2), (int)(hi >> 16));
                }
                const int b1 = (int)(best >> 16);
                if (b1 <= BOW_TH_LOW && (float)b1 < A.nnratio * (float)d2) {
                    const int idx = list[best & 0xFFFFu];
                    if (lane == 0) {
                        s_match[idx] = k;
                        float rot = A.kf_kps[o + k].angle - A.f_kps[o + idx].angle;
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == BOW_HISTO) bin = 0;
                        s_bin[idx] = (unsigned char)bin;
                    }
                    __threadfence_block();
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    __syncthreads();
    int mine = 0;
    for (int j = t; j < nF; j += blockDim.x) if (s_match[j] >= 0) { mine++; if (A.check_ori) atomicAdd(&s_hist[s_bin[j]], 1); }
    if (mine) atomicAdd(&s_nm, mine);
    __syncthreads();
    if (A.check_ori && t == 0) {                                       // ComputeThreeMaxima, src/ORBmatcher.cc:1602-1643
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < BOW_HISTO; i++) {
            const int sz = s_hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        int removed = 0;
        for (int i = 0; i < BOW_HISTO; i++) { const int keep = (i == ind1 || i == ind2 || i == ind3); s_keep[i] = keep; if (!keep) removed += s_hist[i]; }
        s_nm -= removed;
    }
    __syncthreads();
    for (int j = t; j < cap; j += blockDim.x) {
        const int m = j < nF ? s_match[j] : -1;
        A.match[o + j] = (m >= 0 && s_keep[s_bin[j]]) ? m : -1;
    }
    if (t == 0) A.nmatches[b] = s_nm;
}

// ---------------------------------------------------------------------------------------------
// ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:657-823, CheckDistEpipolarLine :138-156): for every key-frame-1
// feature without a map point, the lowest-Hamming key-frame-2 feature of the same vocabulary node that has no map point, is not
// too close to the epipole (mono) and lies on the epipolar line within 3.84 sigma^2; rotation-histogram filter at the end. The
// reference never sets vbMatched2, so key-frame-1 features do not compete: one thread per feature. Key frame 2's features are
// sorted by (node, index) in LDS so that a node is a contiguous run walked in index order ("dist > bestDist -> continue" lets
// the LAST of equal-distance candidates win).
struct TriArgs {
    const viorb_keypoint *k1, *k2; const uint8_t *d1, *d2, *hp1, *hp2; const float *ur1, *ur2; const int *node1, *node2, *n1, *n2;
    const float *F12, *Cw1, *pose2;
    int* match12; int* nmatches;
    int cap, sort_n, only_stereo, check_ori;
    float fx, fy, cx, cy, scale[16], level_sigma2[16];
};
__host__ __device__ inline size_t tri_lds_bytes(int cap, int sort_n) { return (size_t)sort_n * 8 + (size_t)cap + 256; }

__global__ __launch_bounds__(1024) void k_search_triangulation(TriArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int b = blockIdx.x, t = threadIdx.x, cap = A.cap, sn = A.sort_n;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);          // [sn] ((node + 1) << 32 | index), absent nodes sort first
    unsigned char* s_bin = reinterpret_cast<unsigned char*>(keys + sn);              // [cap]
    __shared__ int s_hist[BOW_HISTO], s_keep[BOW_HISTO], s_nm;
    __shared__ float s_e[2];
    const int N1 = min(A.n1[b], cap), N2 = min(A.n2[b], cap);
    const size_t o = (size_t)b * cap;
    for (int i = t; i < sn; i += blockDim.x)
        keys[i] = i < N2 ? (((unsigned long long)(unsigned)(A.node2[o + i] + 1) << 32) | (unsigned)i) : ~0ull;
    if (t < BOW_HISTO) { s_hist[t] = 0; s_keep[t] = 1; }
    if (t == 0) {
        s_nm = 0;
        const float* P = A.pose2 + (size_t)b * 12; const float* C = A.Cw1 + (size_t)b * 3;
        float c2[3];
        for (int r = 0; r < 3; r++) { const float tt = P[3 * r] * C[0] + P[3 * r + 1] * C[1] + P[3 * r + 2] * C[2]; c2[r] = tt + P[9 + r]; }
        const float invz = 1.0f / c2[2];
        s_e[0] = A.fx * c2[0] * invz + A.cx; s_e[1] = A.fy * c2[1] * invz + A.cy;
    }
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
    const float ex = s_e[0], ey = s_e[1];
    const float* F = A.F12 + (size_t)b * 9;
    const float factor = 1.0f / BOW_HISTO;
    for (int i1 = t; i1 < cap; i1 += blockDim.x) {
        int bestIdx2 = -1;
        if (i1 < N1) {
            const int nd = A.node1[o + i1];
            const bool st1 = A.ur1[o + i1] >= 0;
            if (nd >= 0 && !A.hp1[o + i1] && !(A.only_stereo && !st1)) {
                const viorb_keypoint kp1 = A.k1[o + i1];
                const float la = kp1.x * F[0] + kp1.y * F[3] + F[6], lb = kp1.x * F[1] + kp1.y * F[4] + F[7], lc = kp1.x * F[2] + kp1.y * F[5] + F[8];
                const float den = la * la + lb * lb;
                const uint4* p1 = reinterpret_cast<const uint4*>(A.d1 + (o + i1) * 32);
                const uint4 da = p1[0], db = p1[1];
                const unsigned long long want = (unsigned long long)(unsigned)(nd + 1) << 32;
                int lo = 0, hi = N2;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
                int bestDist = BOW_TH_LOW;
                for (int k = lo; k < N2 && (keys[k] >> 32) == (unsigned)(nd + 1); k++) {
                    const int i2 = (int)(keys[k] & 0xffffffffu);
                    if (A.hp2[o + i2]) continue;
                    const bool st2 = A.ur2[o + i2] >= 0;
                    if (A.only_stereo && !st2) continue;
                    const uint4* p2 = reinterpret_cast<const uint4*>(A.d2 + (o + i2) * 32);
                    const uint4 ea = p2[0], eb = p2[1];
                    const int dist = __popc(da.x ^ ea.x) + __popc(da.y ^ ea.y) + __popc(da.z ^ ea.z) + __popc(da.w ^ ea.w) +
                                     __popc(db.x ^ eb.x) + __popc(db.y ^ eb.y) + __popc(db.z ^ eb.z) + __popc(db.w ^ eb.w);
                    if (dist > BOW_TH_LOW || dist > bestDist) continue;
                    const viorb_keypoint kp2 = A.k2[o + i2];
                    if (!st1 && !st2) {
                        const float dx = ex - kp2.x, dy = ey - kp2.y;
                        if (dx * dx + dy * dy < 100 * A.scale[kp2.octave]) continue;
                    }
                    const float num = la * kp2.x + lb * kp2.y + lc;
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    if ((double)dsqr < 3.84 * (double)A.level_sigma2[kp2.octave]) { bestIdx2 = i2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    float rot = kp1.angle - A.k2[o + bestIdx2].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == BOW_HISTO) bin = 0;
                    s_bin[i1] = (unsigned char)bin;
                    atomicAdd(&s_nm, 1);
                    if (A.check_ori) atomicAdd(&s_hist[bin], 1);
                }
            }
        }
        A.match12[o + i1] = bestIdx2;
    }
    __syncthreads();
    if (A.check_ori) {
        if (t == 0) {                                                // ComputeThreeMaxima, src/ORBmatcher.cc:1602-1643
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < BOW_HISTO; i++) {
                const int sz = s_hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
            int removed = 0;
            for (int i = 0; i < BOW_HISTO; i++) { const int keep = (i == ind1 || i == ind2 || i == ind3); s_keep[i] = keep; if (!keep) removed += s_hist[i]; }
            s_nm -= removed;
        }
        __syncthreads();
        for (int i1 = t; i1 < N1; i1 += blockDim.x)
            if (A.match12[o + i1] >= 0 && !s_keep[s_bin[i1]]) A.match12[o + i1] = -1;
    }
    if (t == 0) A.nmatches[b] = s_nm;
}

} // namespace viorb

using namespace viorb;

struct viorb_vocabulary {
    int n_nodes, L, n_edges;
    int *child_start, *child_ids, *word_id; uint8_t* desc; double* weight;
};

namespace {
struct BowBuf {
    std::vector<void*> ptrs;
    ~BowBuf() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> bool up(T** d, const T* src, size_t n) {
        if (hipMalloc((void**)d, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return false;
        ptrs.push_back(*d);
        if (src && n) return hipMemcpy(*d, src, n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
        return hipMemset(*d, 0, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess;
    }
};
}

extern "C" {

int viorb_vocabulary_create(int n_nodes, int L, const int32_t* child_start, const int32_t* child_ids, const uint8_t* desc,
                            const int32_t* word_id, const double* weight, viorb_vocabulary** out) {
    VIORB_REQUIRE(out && child_start && child_ids && desc && word_id && weight, "null array");
    VIORB_REQUIRE(n_nodes >= 2 && L >= 1 && L <= 32, "a vocabulary has a root, at least one word and 1..32 levels");
    *out = nullptr;
    // every index the descent can follow is checked here, once, so that the kernel never needs to
    VIORB_REQUIRE(child_start[0] == 0, "child_start[0] must be 0");
    for (int n = 0; n < n_nodes; n++) VIORB_REQUIRE(child_start[n + 1] >= child_start[n], "child_start must be non-decreasing");
    const int ne = child_start[n_nodes];
    for (int e = 0; e < ne; e++) VIORB_REQUIRE(child_ids[e] > 0 && child_ids[e] < n_nodes, "child id out of range");
    for (int n = 0; n < n_nodes; n++) if (child_start[n + 1] == child_start[n]) VIORB_REQUIRE(word_id[n] >= 0, "a leaf needs a word id");
    VIORB_REQUIRE(child_start[1] > 0, "the root has no children");
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    viorb_vocabulary* v = new viorb_vocabulary();
    v->n_nodes = n_nodes; v->L = L; v->n_edges = ne;
    v->child_start = v->child_ids = v->word_id = nullptr; v->desc = nullptr; v->weight = nullptr;
    hipError_t e = hipSuccess;
    auto up = [&](void** d, const void* s, size_t bytes) { if (e != hipSuccess) return; e = hipMalloc(d, std::max<size_t>(bytes, 16)); if (e == hipSuccess && bytes) e = hipMemcpy(*d, s, bytes, hipMemcpyHostToDevice); };
    up((void**)&v->child_start, child_start, sizeof(int) * ((size_t)n_nodes + 1));
    up((void**)&v->child_ids, child_ids, sizeof(int) * (size_t)ne);
    up((void**)&v->word_id, word_id, sizeof(int) * (size_t)n_nodes);
    up((void**)&v->desc, desc, (size_t)32 * n_nodes);
    up((void**)&v->weight, weight, sizeof(double) * (size_t)n_nodes);
    if (e != hipSuccess) { set_error("vocabulary upload failed: %s", hipGetErrorString(e)); viorb_vocabulary_destroy(v); return VIORB_ERR_HIP; }
    *out = v;
    return VIORB_OK;
}

int viorb_vocabulary_destroy(viorb_vocabulary* v) {
    if (!v) return VIORB_OK;
    (void)hipFree(v->child_start); (void)hipFree(v->child_ids); (void)hipFree(v->word_id); (void)hipFree(v->desc); (void)hipFree(v->weight);
    delete v;
    return VIORB_OK;
}

int viorb_bow_transform_device(const viorb_vocabulary* v, const uint8_t* desc, const int32_t* count, int cap, int batch, int levelsup,
                               int32_t* word, double* weight, int32_t* node, void* stream) {
    VIORB_REQUIRE(v && desc && count && word && weight && node, "null array");
    VIORB_REQUIRE(cap >= 1 && batch >= 1 && (long long)cap * batch < (1ll << 27), "cap * batch out of range");
    BowVoc V; V.child_start = v->child_start; V.child_ids = v->child_ids; V.desc = v->desc; V.word_id = v->word_id; V.weight = v->weight; V.n_nodes = v->n_nodes; V.L = v->L;
    const long long groups = (long long)cap * batch;
    ProfScope ps("k_bow_transform", (hipStream_t)stream);
    hipLaunchKernelGGL(k_bow_transform, dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, (hipStream_t)stream, V, desc, count, cap, batch, levelsup, word, weight, node);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_search_by_bow_device(const viorb_keypoint* kf_kps, const uint8_t* kf_desc, const int32_t* kf_node, const uint8_t* kf_has_point,
                               const int32_t* kf_count, const viorb_keypoint* f_kps, const uint8_t* f_desc, const int32_t* f_node,
                               const int32_t* f_count, int cap, int batch, float nnratio, int check_orientation, int32_t* match,
                               int32_t* nmatches, void* stream) {
    VIORB_REQUIRE(kf_kps && kf_desc && kf_node && kf_has_point && kf_count && f_kps && f_desc && f_node && f_count && match && nmatches, "null array");
    VIORB_REQUIRE(cap >= 1 && cap <= 65535 && batch >= 1, "1 <= cap <= 65535");
    const size_t lds = bow_search_lds_bytes(cap);
    const bool gw = lds > 160 * 1024;                                    // more keypoints than the work arrays fit in LDS (~7100): global memory
    if (!gw && lds > 64 * 1024) VIORB_HIP_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(k_search_by_bow<false>), lds));
    BowSearchArgs A;
    A.kf_kps = kf_kps; A.kf_desc = kf_desc; A.kf_node = kf_node; A.kf_has_point = kf_has_point; A.kf_count = kf_count;
    A.f_kps = f_kps; A.f_desc = f_desc; A.f_node = f_node; A.f_count = f_count; A.match = match; A.nmatches = nmatches;
    A.cap = cap; A.nnratio = nnratio; A.check_ori = check_orientation;
    A.work = nullptr; A.work_bytes = 0;
    ProfScope ps("k_search_by_bow", (hipStream_t)stream);
    if (gw) {
        // grow-only scratch of the calling thread (this entry point has no handle to keep it in)
        struct Scratch { unsigned char* p = nullptr; size_t bytes = 0; int device = -1; ~Scratch() { if (p) (void)hipFree(p); } };
        static thread_local Scratch sc;
        int dev = 0; VIORB_HIP_TRY(hipGetDevice(&dev));
        const size_t per = (lds + 255) & ~(size_t)255;
        if (sc.device != dev || sc.bytes < per * batch) {
            VIORB_HIP_TRY(hipDeviceSynchronize());
            if (sc.p) (void)hipFree(sc.p);
            sc.p = nullptr; sc.bytes = 0; sc.device = dev;
            VIORB_HIP_TRY(hipMalloc(&sc.p, per * batch));
            sc.bytes = per * batch;
        }
        A.work = sc.p; A.work_bytes = per;
        hipLaunchKernelGGL(k_search_by_bow<true>, dim3(batch), dim3(256), 0, (hipStream_t)stream, A);
    } else
    hipLaunchKernelGGL(k_search_by_bow<false>, dim3(batch), dim3(256), lds, (hipStream_t)stream, A);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_bow_transform(const viorb_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word, double* weight, int32_t* node) {
    VIORB_REQUIRE(v && n >= 0, "null vocabulary");
    if (n == 0) return VIORB_OK;
    VIORB_REQUIRE(desc && word && weight && node, "null array");
    BowBuf B; uint8_t* d_desc; int *d_cnt, *d_word, *d_node; double* d_w;
    if (!(B.up(&d_desc, desc, (size_t)32 * n) && B.up(&d_cnt, &n, 1) && B.up(&d_word, (const int*)nullptr, n) && B.up(&d_node, (const int*)nullptr, n) && B.up(&d_w, (const double*)nullptr, n))) {
        set_error("device allocation / upload failed"); return VIORB_ERR_HIP;
    }
    int rc = viorb_bow_transform_device(v, d_desc, d_cnt, n, 1, levelsup, d_word, d_w, d_node, nullptr);
    if (rc != VIORB_OK) return rc;
    VIORB_HIP_TRY(hipDeviceSynchronize());
    VIORB_HIP_TRY(hipMemcpy(word, d_word, sizeof(int) * n, hipMemcpyDeviceToHost));
    VIORB_HIP_TRY(hipMemcpy(weight, d_w, sizeof(double) * n, hipMemcpyDeviceToHost));
    VIORB_HIP_TRY(hipMemcpy(node, d_node, sizeof(int) * n, hipMemcpyDeviceToHost));
    return VIORB_OK;
}

int viorb_search_by_bow(const viorb_keypoint* kf_kps, const uint8_t* kf_desc, const int32_t* kf_node, const uint8_t* kf_has_point, int nkf,
                        const viorb_keypoint* f_kps, const uint8_t* f_desc, const int32_t* f_node, int nf, float nnratio,
                        int check_orientation, int32_t* match, int* nmatches) {
    VIORB_REQUIRE(nmatches && nkf >= 0 && nf >= 0, "null array");
    *nmatches = 0;
    for (int i = 0; i < nf; i++) match[i] = -1;
    if (nkf == 0 || nf == 0) return VIORB_OK;
    VIORB_REQUIRE(kf_kps && kf_desc && kf_node && kf_has_point && f_kps && f_desc && f_node && match, "null array");
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    const int cap = std::max(nkf, nf);
    BowBuf B; viorb_keypoint *d_kk, *d_fk; uint8_t *d_kd, *d_fd, *d_kh; int *d_kn, *d_fn, *d_kc, *d_fc, *d_m, *d_nm;
    bool ok = B.up(&d_kk, (const viorb_keypoint*)nullptr, cap) && B.up(&d_fk, (const viorb_keypoint*)nullptr, cap) && B.up(&d_kd, (const uint8_t*)nullptr, (size_t)32 * cap) &&
              B.up(&d_fd, (const uint8_t*)nullptr, (size_t)32 * cap) && B.up(&d_kh, (const uint8_t*)nullptr, cap) && B.up(&d_kn, (const int*)nullptr, cap) &&
              B.up(&d_fn, (const int*)nullptr, cap) && B.up(&d_kc, &nkf, 1) && B.up(&d_fc, &nf, 1) && B.up(&d_m, (const int*)nullptr, cap) && B.up(&d_nm, (const int*)nullptr, 1);
    if (!ok) { set_error("device allocation / upload failed"); return VIORB_ERR_HIP; }
    VIORB_HIP_TRY(hipMemcpy(d_kk, kf_kps, sizeof(viorb_keypoint) * nkf, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(d_fk, f_kps, sizeof(viorb_keypoint) * nf, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(d_kd, kf_desc, (size_t)32 * nkf, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(d_fd, f_desc, (size_t)32 * nf, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(d_kh, kf_has_point, nkf, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(d_kn, kf_node, sizeof(int) * nkf, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(d_fn, f_node, sizeof(int) * nf, hipMemcpyHostToDevice));
    int rc = viorb_search_by_bow_device(d_kk, d_kd, d_kn, d_kh, d_kc, d_fk, d_fd, d_fn, d_fc, cap, 1, nnratio, check_orientation, d_m, d_nm, nullptr);
    if (rc != VIORB_OK) return rc;
    VIORB_HIP_TRY(hipDeviceSynchronize());
    VIORB_HIP_TRY(hipMemcpy(match, d_m, sizeof(int) * nf, hipMemcpyDeviceToHost));
    VIORB_HIP_TRY(hipMemcpy(nmatches, d_nm, sizeof(int), hipMemcpyDeviceToHost));
    return VIORB_OK;
}

int viorb_search_for_triangulation_device(const viorb_keypoint* k1, const uint8_t* d1, const uint8_t* has_point1, const float* uright1,
                                          const int32_t* node1, const int32_t* n1, const viorb_keypoint* k2, const uint8_t* d2,
                                          const uint8_t* has_point2, const float* uright2, const int32_t* node2, const int32_t* n2,
                                          const float* F12, const float* Cw1, const float* pose12_2, const float intr4[4],
                                          const float* scale_factors2, const float* level_sigma2_2, int nlevels, int only_stereo,
                                          int check_orientation, int cap, int batch, int32_t* match12, int32_t* nmatches, void* stream) {
    VIORB_REQUIRE(k1 && d1 && has_point1 && uright1 && node1 && n1 && k2 && d2 && has_point2 && uright2 && node2 && n2 && F12 && Cw1 && pose12_2 &&
                  intr4 && scale_factors2 && level_sigma2_2 && match12 && nmatches, "null array");
    VIORB_REQUIRE(cap >= 1 && cap <= 16384 && batch >= 1 && nlevels >= 1 && nlevels <= 16, "1 <= cap <= 16384, 1 <= nlevels <= 16");
    int sn = 1; while (sn < cap) sn <<= 1;
    const size_t lds = tri_lds_bytes(cap, sn);
    if (lds > 160 * 1024) { set_error("cap %d needs %zu B of LDS for SearchForTriangulation", cap, lds); return VIORB_ERR_UNSUPPORTED; }
    if (lds > 64 * 1024) VIORB_HIP_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(k_search_triangulation), lds));
    TriArgs A;
    A.k1 = k1; A.k2 = k2; A.d1 = d1; A.d2 = d2; A.hp1 = has_point1; A.hp2 = has_point2; A.ur1 = uright1; A.ur2 = uright2; A.node1 = node1; A.node2 = node2;
    A.n1 = n1; A.n2 = n2; A.F12 = F12; A.Cw1 = Cw1; A.pose2 = pose12_2; A.match12 = match12; A.nmatches = nmatches;
    A.cap = cap; A.sort_n = sn; A.only_stereo = only_stereo; A.check_ori = check_orientation;
    A.fx = intr4[0]; A.fy = intr4[1]; A.cx = intr4[2]; A.cy = intr4[3];
    for (int i = 0; i < 16; i++) { A.scale[i] = scale_factors2[i < nlevels ? i : nlevels - 1]; A.level_sigma2[i] = level_sigma2_2[i < nlevels ? i : nlevels - 1]; }
    ProfScope ps("k_search_triangulation", (hipStream_t)stream);
    hipLaunchKernelGGL(k_search_triangulation, dim3(batch), dim3(1024), lds, (hipStream_t)stream, A);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_search_for_triangulation(const viorb_keypoint* k1, const uint8_t* d1, const uint8_t* has_point1, const float* uright1,
                                   const int32_t* node1, int n1, const viorb_keypoint* k2, const uint8_t* d2, const uint8_t* has_point2,
                                   const float* uright2, const int32_t* node2, int n2, const float F12[9], const float Cw1[3],
                                   const float pose12_2[12], const float intr4[4], const float* scale_factors2,
                                   const float* level_sigma2_2, int nlevels, int only_stereo, int check_orientation, int32_t* match12,
                                   int* nmatches) {
    VIORB_REQUIRE(nmatches && n1 >= 0 && n2 >= 0, "null array");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0) return VIORB_OK;
    VIORB_REQUIRE(k1 && d1 && has_point1 && uright1 && node1 && k2 && d2 && has_point2 && uright2 && node2 && F12 && Cw1 && pose12_2 && match12, "null array");
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    const int cap = std::max(n1, n2);
    BowBuf B; viorb_keypoint *dk1, *dk2; uint8_t *dd1, *dd2, *dh1, *dh2; float *du1, *du2, *dF, *dC, *dP; int *dn1, *dn2, *dc1, *dc2, *dm, *dnm;
    bool ok = B.up(&dk1, (const viorb_keypoint*)nullptr, cap) && B.up(&dk2, (const viorb_keypoint*)nullptr, cap) && B.up(&dd1, (const uint8_t*)nullptr, (size_t)32 * cap) &&
              B.up(&dd2, (const uint8_t*)nullptr, (size_t)32 * cap) && B.up(&dh1, (const uint8_t*)nullptr, cap) && B.up(&dh2, (const uint8_t*)nullptr, cap) &&
              B.up(&du1, (const float*)nullptr, cap) && B.up(&du2, (const float*)nullptr, cap) && B.up(&dF, F12, 9) && B.up(&dC, Cw1, 3) && B.up(&dP, pose12_2, 12) &&
              B.up(&dn1, (const int*)nullptr, cap) && B.up(&dn2, (const int*)nullptr, cap) && B.up(&dc1, &n1, 1) && B.up(&dc2, &n2, 1) &&
              B.up(&dm, (const int*)nullptr, cap) && B.up(&dnm, (const int*)nullptr, 1);
    if (!ok) { set_error("device allocation / upload failed"); return VIORB_ERR_HIP; }
    VIORB_HIP_TRY(hipMemcpy(dk1, k1, sizeof(viorb_keypoint) * n1, hipMemcpyHostToDevice)); VIORB_HIP_TRY(hipMemcpy(dk2, k2, sizeof(viorb_keypoint) * n2, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(dd1, d1, (size_t)32 * n1, hipMemcpyHostToDevice)); VIORB_HIP_TRY(hipMemcpy(dd2, d2, (size_t)32 * n2, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(dh1, has_point1, n1, hipMemcpyHostToDevice)); VIORB_HIP_TRY(hipMemcpy(dh2, has_point2, n2, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(du1, uright1, sizeof(float) * n1, hipMemcpyHostToDevice)); VIORB_HIP_TRY(hipMemcpy(du2, uright2, sizeof(float) * n2, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(dn1, node1, sizeof(int) * n1, hipMemcpyHostToDevice)); VIORB_HIP_TRY(hipMemcpy(dn2, node2, sizeof(int) * n2, hipMemcpyHostToDevice));
    int rc = viorb_search_for_triangulation_device(dk1, dd1, dh1, du1, dn1, dc1, dk2, dd2, dh2, du2, dn2, dc2, dF, dC, dP, intr4, scale_factors2, level_sigma2_2,
                                                   nlevels, only_stereo, check_orientation, cap, 1, dm, dnm, nullptr);
    if (rc != VIORB_OK) return rc;
    VIORB_HIP_TRY(hipDeviceSynchronize());
    VIORB_HIP_TRY(hipMemcpy(match12, dm, sizeof(int) * n1, hipMemcpyDeviceToHost));
    VIORB_HIP_TRY(hipMemcpy(nmatches, dnm, sizeof(int), hipMemcpyDeviceToHost));
    return VIORB_OK;
}

} // extern "C"
