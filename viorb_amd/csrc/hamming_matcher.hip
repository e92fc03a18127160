// viorb_amd/csrc/hamming_matcher.hip — brute-force Hamming tile matcher (north_star "Hamming brute-force", SURVEY.md §8b
// viorb_match_bruteforce, §8d op count Nq * Nc * 8):
//   k_match_bruteforce   for every query descriptor the smallest and second-smallest ORBmatcher::DescriptorDistance
//                        (reference src/ORBmatcher.cc:1648-1664: 8 x 32-bit xor + SWAR popcount) over all candidates and the index
//                        of the first candidate at the smallest distance — the bestDist1 / bestDist2 / bestIdx scan every
//                        ORBmatcher::Search* runs over its candidate list (e.g. src/ORBmatcher.cc:204-222, strict '<': the first
//                        candidate wins a tie), here over ALL candidates of the other frame.
// One lane per query (its 8 descriptor words stay in registers), 256 queries per workgroup; the candidates go through LDS in tiles of
// 256 descriptors (8 KB, loaded as two 16-byte vectors per thread, next tile's loads in flight under the current tile's arithmetic)
// and every lane of a wave reads the same candidate (LDS broadcast, two ds_read_b128 per candidate per wave). Per pair: 8 v_xor +
// 8 v_bcnt_u32_b32 (popcount with accumulate) + 5 compare / select = 21 vector instructions, so the kernel is bound by integer
// vector issue, not by LDS or HBM: the roofline the bench of tools/bruteforce_bench.py reports is popcounts per second against the
// chip's 32-bit integer issue rate.
#include <hip/hip_runtime.h>
#include <vector>
#include "viorb_common.h"

namespace viorb {

#define BF_THREADS 256
#define BF_TILE 256

__global__ __launch_bounds__(BF_THREADS) void k_match_bruteforce(const uint8_t* __restrict__ q_desc, const int* __restrict__ nq, int qcap,
                                                                 const uint8_t* __restrict__ c_desc, const int* __restrict__ nc, int ccap,
                                                                 int* __restrict__ best, int* __restrict__ second, int* __restrict__ idx) {
    __shared__ uint4 tile[2][BF_TILE * 2];
    const int b = blockIdx.y, t = threadIdx.x, qi = blockIdx.x * BF_THREADS + t;
    const int n_q = min(nq[b], qcap), n_c = min(nc[b], ccap);
    if (blockIdx.x * BF_THREADS >= n_q) return;                            // workgroup-uniform
    const uint4* qp = reinterpret_cast<const uint4*>(q_desc + ((size_t)b * qcap + min(qi, n_q - 1)) * 32);
    const uint4 q0 = qp[0], q1 = qp[1];
    const uint4* cbase = reinterpret_cast<const uint4*>(c_desc + (size_t)b * ccap * 32);
    int d1 = 256, d2 = 256, i1 = -1;                                       // bestDist1 = bestDist2 = 256, no index yet
    const int ntiles = (n_c + BF_TILE - 1) / BF_TILE;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
    if (ntiles > 0) { const int c = min(t, n_c - 1); r0 = cbase[2 * c]; r1 = cbase[2 * c + 1]; }
    for (int tl = 0; tl < ntiles; tl++) {
        uint4* T = tile[tl & 1];
        T[2 * t] = r0; T[2 * t + 1] = r1;
        __syncthreads();                                                   // tile tl complete; tile tl-1's readers are past it (two buffers)
        if (tl + 1 < ntiles) { const int c = min((tl + 1) * BF_TILE + t, n_c - 1); r0 = cbase[2 * c]; r1 = cbase[2 * c + 1]; }
        const int base = tl * BF_TILE, m = min(BF_TILE, n_c - base);
#pragma unroll 4
        for (int j = 0; j < m; j++) {
            const uint4 a = T[2 * j], c = T[2 * j + 1];                    // same address in every lane: broadcast
            int d = __popc(q0.x ^ a.x);
            d += __popc(q0.y ^ a.y); d += __popc(q0.z ^ a.z); d += __popc(q0.w ^ a.w);
            d += __popc(q1.x ^ c.x); d += __popc(q1.y ^ c.y); d += __popc(q1.z ^ c.z); d += __popc(q1.w ^ c.w);
            // if (d < d1) { d2 = d1; d1 = d; i1 = j } else if (d < d2) d2 = d;   branch-free
            const bool lt1 = d < d1;
            d2 = lt1 ? d1 : min(d2, d);
            i1 = lt1 ? base + j : i1;
            d1 = lt1 ? d : d1;
        }
    }
    if (qi < n_q) {
        const size_t o = (size_t)b * qcap + qi;
        best[o] = d1; second[o] = d2; idx[o] = i1;
    }
}

struct BfBuf {
    std::vector<void*> p;
    ~BfBuf() { for (void* x : p) (void)hipFree(x); }
    template <class T> bool get(T** out, size_t n) { void* d = nullptr; if (hipMalloc(&d, sizeof(T) * (n ? n : 1)) != hipSuccess) return false; p.push_back(d); *out = (T*)d; return true; }
};

} // namespace viorb

using namespace viorb;

extern "C" {

int viorb_match_bruteforce_device(const uint8_t* q_desc, const int32_t* nq, int qcap, const uint8_t* c_desc, const int32_t* nc, int ccap,
                                  int batch, int32_t* best, int32_t* second, int32_t* idx, void* stream) {
    VIORB_REQUIRE(q_desc && nq && c_desc && nc && best && second && idx, "null array");
    VIORB_REQUIRE(qcap >= 1 && ccap >= 1 && batch >= 1 && batch <= 65535, "qcap, ccap >= 1, 1 <= batch <= 65535");
    ProfScope ps("k_match_bruteforce", (hipStream_t)stream);
    hipLaunchKernelGGL(k_match_bruteforce, dim3((qcap + BF_THREADS - 1) / BF_THREADS, batch), dim3(BF_THREADS), 0, (hipStream_t)stream,
                       q_desc, nq, qcap, c_desc, nc, ccap, best, second, idx);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}

int viorb_match_bruteforce(const uint8_t* q, int nq, const uint8_t* c, int nc, int32_t* best, int32_t* second, int32_t* idx) {
    VIORB_REQUIRE(nq >= 0 && nc >= 0, "negative count");
    if (nq == 0) return VIORB_OK;
    VIORB_REQUIRE(q && best && second && idx && (c || nc == 0), "null array");
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    BfBuf B; uint8_t *dq, *dc; int *dn, *db, *ds, *di;
    if (!(B.get(&dq, (size_t)32 * nq) && B.get(&dc, (size_t)32 * (nc ? nc : 1)) && B.get(&dn, 2) && B.get(&db, nq) && B.get(&ds, nq) && B.get(&di, nq))) {
        set_error("device allocation failed"); return VIORB_ERR_HIP;
    }
    const int cnt[2] = {nq, nc};
    VIORB_HIP_TRY(hipMemcpy(dq, q, (size_t)32 * nq, hipMemcpyHostToDevice));
    if (nc) VIORB_HIP_TRY(hipMemcpy(dc, c, (size_t)32 * nc, hipMemcpyHostToDevice));
    VIORB_HIP_TRY(hipMemcpy(dn, cnt, sizeof(cnt), hipMemcpyHostToDevice));
    int rc = viorb_match_bruteforce_device(dq, dn, nq, dc, dn + 1, nc ? nc : 1, 1, db, ds, di, nullptr);
    if (rc != VIORB_OK) return rc;
    VIORB_HIP_TRY(hipDeviceSynchronize());
    VIORB_HIP_TRY(hipMemcpy(best, db, sizeof(int) * nq, hipMemcpyDeviceToHost));
    VIORB_HIP_TRY(hipMemcpy(second, ds, sizeof(int) * nq, hipMemcpyDeviceToHost));
    VIORB_HIP_TRY(hipMemcpy(idx, di, sizeof(int) * nq, hipMemcpyDeviceToHost));
    return VIORB_OK;
}

} // extern "C"
