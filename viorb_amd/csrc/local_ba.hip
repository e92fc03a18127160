// viorb_amd/csrc/local_ba.hip — Optimizer::LocalBundleAdjustmentNavState (reference src/Optimizer.cc:1690-2241) on the GPU.
//
// Graph: W local key frames with free PVR(9)+Bias(3) vertices, fixed key frames, marginalised 3-D points, one IMU factor
// (src/IMU/g2otypes.cpp:8-229, information = cov^-1, not inflated: Optimizer.cc:1901-1902) and one bias factor per local
// key frame, one EdgeNavStatePVRPointXYZ (src/IMU/g2otypes.h:129-203, g2otypes.cpp:299-354) per observation.
// Solver: g2o's Levenberg-Marquardt (optimization_algorithm_levenberg.cpp:61-189) with the Schur complement of the point
// block (block_solver.hpp:367-486), 5 + 10 iterations with the chi2 / depth gate in between.
//
// Kernels (all FP64): k_ba_errors (per edge residuals + robust chi2), k_ba_lin_points (one thread per point: Jacobians,
// weights, Hll, bl), k_ba_hpp (one workgroup per local key frame: its 6x6 (P,Phi) block), k_ba_imu (one workgroup per IMU
// factor), k_ba_dinv (one thread per point: (Hll + lambda I)^-1), k_ba_schur (one wavefront per key-frame pair: its 6x6 block of
// -W D^-1 W^T gathered over the points both key frames observe, W blocks precomputed with the linearisation, no atomics), k_ba_chol_solve (dense Cholesky of the <= 240x240 reduced system by
// one 512-thread workgroup, register tiles + v_mfma_f64_16x16x4), k_ba_backsub, k_ba_update. The LM control flow (accept / reject, lambda
// schedule, stop rule) runs on the device (k_ba_decide / the lock-step batch's control block); the host polls a few scalars per chunk.
#include <hip/hip_runtime.h>
#include <memory>
#include <string>
#include <atomic>
#include <thread>
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <mutex>
#include "viorb_common.h"
#include "vio_core.h"

namespace viorb {

struct BaDev {
    int W, NK, NP, NE, np, ld, prev_kf;
    int pose_dim, rows, kf_stride;  // unknowns per key frame (12: PVR+bias, 6: SE3), residual rows per edge (2, or 3 with stereo), doubles per key-frame state
    double *kf, *kf_bak;            // [NK][22]
    double *pt, *pt_bak;            // [NP][3]
    const int *e_pt, *e_kf;         // [NE]
    const double *e_obs;            // [NE][3] u v invSigma2
    uint8_t* level;                 // [NE]
    double *err, *Jp, *Jk, *wgt;    // [NE][2], [NE][6], [NE][12], [NE]
    double *We;                     // [NE][6][3] = wgt * Jk^T Jp, the edge's block of W (block_solver.hpp's Hpl), written with the linearisation
    int *pr_start, *pr_ent;         // CSR over the key-frame pairs (a >= b, pair a (a + 1) / 2 + b): (edge of a, edge of b) of every point both observe (k_ba_pairs_*)
    const int *pt_start;            // [NP+1]
    const int *kf_start, *kf_list;  // CSR of the edges of each local key frame
    double *Hll, *bl, *Dinv, *db;   // [NP][9], [NP][3], [NP][9], [NP][3] = Dinv bl
    double *Hpp, *bp, *S, *bs, *xp, *xl;
    const double *preint, *info_pvr; // [W][142], [W][81]
    double *e_pvr, *e_b;            // [W][9], [W][3]
    double *scal;                   // [8]: 0 chi2, 1 scale, 2 ok, 3 max diag
    double *ctl;                    // device-side LM control (BA_CTL_*), used when use_ctl != 0: kernels return at once while ctl[HALT] != 0
    int *ticket;                    // block counter of k_ba_f_errors_decide (the last block to add its chi2 takes the trial's decision)
    int use_ctl;                    // and take lambda from ctl[LAMBDA] instead of their argument
    const unsigned long long* abort_host;   // page-locked word the waiting host thread sets when the caller's pbStopFlag goes up (nullptr: no flag)
    double cam[16], gw[3];
    double acc_bias_rw2;
};

// Device-side Levenberg control (g2o optimization_algorithm_levenberg.cpp:61-164) for the common case of an LM iteration whose first
// trial is accepted: the host enqueues several iterations back to back, k_ba_decide takes the accept / stop decisions on the device, and
// after a REJECTED trial every later kernel of the chunk returns at once (ctl[HALT] = 1) so that the host finds the system, the trial
// state and the trial's scalars untouched and continues with its own trial loop (lambda *= ni, restore, retry).
enum { BA_CTL_LAMBDA = 0, BA_CTL_NI, BA_CTL_CHI, BA_CTL_INICHI, BA_CTL_NBAD, BA_CTL_HALT, BA_CTL_ITS, BA_CTL_IT, BA_CTL_RHO, BA_CTL_N = 32 };
__device__ __forceinline__ bool ba_skip(const BaDev& D) { return D.use_ctl && D.ctl[BA_CTL_HALT] != 0.0; }
// g2o polls terminate() once per iteration and once per LM trial (optimization_algorithm_levenberg.cpp / sparse_optimizer.cpp:354-432); the
// device-side LM control does the same through a system-scope load of the mirrored flag, so an abort costs at most the trial in flight
__device__ __forceinline__ bool ba_abort_requested(const BaDev& D) {
    return D.abort_host && __hip_atomic_load(D.abort_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull;
}
__device__ __forceinline__ double ba_lambda(const BaDev& D, double arg) { return D.use_ctl ? D.ctl[BA_CTL_LAMBDA] : arg; }

__device__ __forceinline__ void ba_edge_geom(const BaDev& D, int k, const double* kfv, const double* ptv, d3& Pc, m33& RwbT, d3& Paux, cam_t& K) {
    K = ld_cam(D.cam);
    const pvr s = ld_pvr(kfv + (size_t)D.e_kf[k] * 22);
    RwbT = tr(qmat(s.q));
    Paux = mulv(K.Rcb, mulv(RwbT, ld3(ptv + (size_t)D.e_pt[k] * 3) - s.P));
    Pc = Paux - K.RcbPbc;
}
__device__ __forceinline__ int ba_pred(const BaDev& D, int i) { return i == 0 ? D.prev_kf : i - 1; }
// position, inside a key frame's block of unknowns, of the r-th of the six coordinates a reprojection edge depends on:
// NavState block = [P V Phi | bias] -> P at 0..2, Phi at 6..8; SE3 block = [omega upsilon] -> 0..5
__device__ __forceinline__ int ba_loc(const BaDev& D, int r) { return D.pose_dim == 12 ? (r < 3 ? r : r + 3) : r; }

// residuals of the active edges + robust chi2 (mono kernel optional) + IMU / bias factors
__device__ __forceinline__ void k_ba_errors_body(const BaDev& D, int mono_kernel) {
    if (ba_skip(D)) return;
    __shared__ double s_red[8];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double c = 0;
    const double d_mono = (double)(float)sqrt(5.991);
    if (k < D.NE && D.level[k] == 0) {
        d3 Pc, Paux; m33 RT; cam_t K;
        ba_edge_geom(D, k, D.kf, D.pt, Pc, RT, Paux, K);
        const double e0 = D.e_obs[3 * k] - (Pc.x / Pc.z * K.fx + K.cx), e1 = D.e_obs[3 * k + 1] - (Pc.y / Pc.z * K.fy + K.cy);
        D.err[2 * k] = e0; D.err[2 * k + 1] = e1;
        const double chi = D.e_obs[3 * k + 2] * (e0 * e0 + e1 * e1);
        double r0 = chi, r1;
        if (mono_kernel) huber(chi, d_mono, &r0, &r1);
        c = r0;
    }
    if (blockIdx.x == 0 && threadIdx.x < D.W) {               // W <= blockDim.x
        const int i = threadIdx.x, j = ba_pred(D, i);
        if (j >= 0) {
            const double* ki = D.kf + (size_t)i * 22; const double* kj = D.kf + (size_t)j * 22;
            double e[9];
            pvr_edge(ld_pvr(kj), ld_pvr(ki), ld3(kj + 16), ld3(kj + 19), D.preint + (size_t)i * 142, ld3(D.gw), e, nullptr);
            double chi = 0;
            for (int a = 0; a < 9; a++) { double t = 0; for (int b = 0; b < 9; b++) t += D.info_pvr[i * 81 + a * 9 + b] * e[b]; chi += e[a] * t; D.e_pvr[i * 9 + a] = e[a]; }
            double r0, r1; huber(chi, (double)(float)sqrt(21.666), &r0, &r1); c += r0;
            const d3 eb = (ld3(ki + 13) + ld3(ki + 19)) - (ld3(kj + 13) + ld3(kj + 19));
            st3(D.e_b + i * 3, eb);
            huber(dot3(eb, eb) / D.acc_bias_rw2 / D.preint[(size_t)i * 142 + 141], (double)(float)sqrt(16.812), &r0, &r1); c += r0;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += s_red[w]; atomicAdd(&D.scal[0], t); }
}

// one thread per point: Jacobians + weights of its active edges (stored per edge), Hll and bl of the point
__device__ __forceinline__ void k_ba_lin_points_body(const BaDev& D, int mono_kernel) {
    if (ba_skip(D)) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.NP) return;
    double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    const double d_mono = (double)(float)sqrt(5.991);
    for (int k = D.pt_start[p]; k < D.pt_start[p + 1]; k++) {
        if (D.level[k] != 0) continue;
        d3 Pc, Paux; m33 RT; cam_t K;
        ba_edge_geom(D, k, D.kf, D.pt, Pc, RT, Paux, K);
        const double x = Pc.x, y = Pc.y, z = Pc.z;
        const double j00 = K.fx / z, j02 = -x / z * K.fx / z, j11 = K.fy / z, j12 = -y / z * K.fy / z;
        const m33 RR = mul(K.Rcb, RT), HR = mul(hat3(Paux), K.Rcb);
        double Jp[6], Jk[12];
        // point block: -Jpi * Rcb * Rwb^T ; key-frame block: Jpi * Rcb | -Jpi * hat(Paux) * Rcb
        Jp[0] = -(j00 * RR.a00 + j02 * RR.a20); Jp[1] = -(j00 * RR.a01 + j02 * RR.a21); Jp[2] = -(j00 * RR.a02 + j02 * RR.a22);
        Jp[3] = -(j11 * RR.a10 + j12 * RR.a20); Jp[4] = -(j11 * RR.a11 + j12 * RR.a21); Jp[5] = -(j11 * RR.a12 + j12 * RR.a22);
        Jk[0] = j00 * K.Rcb.a00 + j02 * K.Rcb.a20; Jk[1] = j00 * K.Rcb.a01 + j02 * K.Rcb.a21; Jk[2] = j00 * K.Rcb.a02 + j02 * K.Rcb.a22;
        Jk[3] = -(j00 * HR.a00 + j02 * HR.a20); Jk[4] = -(j00 * HR.a01 + j02 * HR.a21); Jk[5] = -(j00 * HR.a02 + j02 * HR.a22);
        Jk[6] = j11 * K.Rcb.a10 + j12 * K.Rcb.a20; Jk[7] = j11 * K.Rcb.a11 + j12 * K.Rcb.a21; Jk[8] = j11 * K.Rcb.a12 + j12 * K.Rcb.a22;
        Jk[9] = -(j11 * HR.a10 + j12 * HR.a20); Jk[10] = -(j11 * HR.a11 + j12 * HR.a21); Jk[11] = -(j11 * HR.a12 + j12 * HR.a22);
        const double e0 = D.err[2 * k], e1 = D.err[2 * k + 1], is2 = D.e_obs[3 * k + 2];
        double r0, r1 = 1;
        if (mono_kernel) huber(is2 * (e0 * e0 + e1 * e1), d_mono, &r0, &r1);
        const double w = r1 * is2;
        D.wgt[k] = w;
        for (int a = 0; a < 6; a++) D.Jp[6 * k + a] = Jp[a];
        for (int a = 0; a < 12; a++) D.Jk[12 * k + a] = Jk[a];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) D.We[18 * (size_t)k + 3 * r + c] = w * (Jk[r] * Jp[c] + Jk[6 + r] * Jp[3 + c]);
        H[0] += w * (Jp[0] * Jp[0] + Jp[3] * Jp[3]); H[1] += w * (Jp[0] * Jp[1] + Jp[3] * Jp[4]); H[2] += w * (Jp[0] * Jp[2] + Jp[3] * Jp[5]);
        H[3] += w * (Jp[1] * Jp[1] + Jp[4] * Jp[4]); H[4] += w * (Jp[1] * Jp[2] + Jp[4] * Jp[5]); H[5] += w * (Jp[2] * Jp[2] + Jp[5] * Jp[5]);
        for (int a = 0; a < 3; a++) b[a] -= w * (Jp[a] * e0 + Jp[3 + a] * e1);
    }
    double* Ho = D.Hll + (size_t)p * 9;
    Ho[0] = H[0]; Ho[1] = H[1]; Ho[2] = H[2]; Ho[3] = H[1]; Ho[4] = H[3]; Ho[5] = H[4]; Ho[6] = H[2]; Ho[7] = H[4]; Ho[8] = H[5];
    for (int a = 0; a < 3; a++) D.bl[(size_t)p * 3 + a] = b[a];
}

// The same linearisation with one thread per EDGE (Jacobians, weight, W block) and the point blocks summed afterwards by one thread per
// point from the stored Jacobians (k_ba_hll_body: same terms in the same order as k_ba_lin_points_body, bit for bit): a single window
// has 2000 points = 8 workgroups for the version above but 43 for this one.
__device__ __forceinline__ void k_ba_lin_edges_body(const BaDev& D, int mono_kernel) {
    if (ba_skip(D)) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= D.NE || D.level[k] != 0) return;
    const double d_mono = (double)(float)sqrt(5.991);
    d3 Pc, Paux; m33 RT; cam_t K;
    ba_edge_geom(D, k, D.kf, D.pt, Pc, RT, Paux, K);
    const double x = Pc.x, y = Pc.y, z = Pc.z;
    const double j00 = K.fx / z, j02 = -x / z * K.fx / z, j11 = K.fy / z, j12 = -y / z * K.fy / z;
    const m33 RR = mul(K.Rcb, RT), HR = mul(hat3(Paux), K.Rcb);
    double Jp[6], Jk[12];
    Jp[0] = -(j00 * RR.a00 + j02 * RR.a20); Jp[1] = -(j00 * RR.a01 + j02 * RR.a21); Jp[2] = -(j00 * RR.a02 + j02 * RR.a22);
    Jp[3] = -(j11 * RR.a10 + j12 * RR.a20); Jp[4] = -(j11 * RR.a11 + j12 * RR.a21); Jp[5] = -(j11 * RR.a12 + j12 * RR.a22);
    Jk[0] = j00 * K.Rcb.a00 + j02 * K.Rcb.a20; Jk[1] = j00 * K.Rcb.a01 + j02 * K.Rcb.a21; Jk[2] = j00 * K.Rcb.a02 + j02 * K.Rcb.a22;
    Jk[3] = -(j00 * HR.a00 + j02 * HR.a20); Jk[4] = -(j00 * HR.a01 + j02 * HR.a21); Jk[5] = -(j00 * HR.a02 + j02 * HR.a22);
    Jk[6] = j11 * K.Rcb.a10 + j12 * K.Rcb.a20; Jk[7] = j11 * K.Rcb.a11 + j12 * K.Rcb.a21; Jk[8] = j11 * K.Rcb.a12 + j12 * K.Rcb.a22;
    Jk[9] = -(j11 * HR.a10 + j12 * HR.a20); Jk[10] = -(j11 * HR.a11 + j12 * HR.a21); Jk[11] = -(j11 * HR.a12 + j12 * HR.a22);
    const double e0 = D.err[2 * k], e1 = D.err[2 * k + 1], is2 = D.e_obs[3 * k + 2];
    double r0, r1 = 1;
    if (mono_kernel) huber(is2 * (e0 * e0 + e1 * e1), d_mono, &r0, &r1);
    const double w = r1 * is2;
    D.wgt[k] = w;
    // 16-byte stores (the blocks are 48, 96 and 144 bytes: 16-byte aligned): the kernel is bound by the number of per-lane memory operations
    double2* Jpo = reinterpret_cast<double2*>(D.Jp + 6 * (size_t)k); double2* Jko = reinterpret_cast<double2*>(D.Jk + 12 * (size_t)k);
    double2* Wo = reinterpret_cast<double2*>(D.We + 18 * (size_t)k);
#pragma unroll
    for (int a = 0; a < 3; a++) Jpo[a] = make_double2(Jp[2 * a], Jp[2 * a + 1]);
#pragma unroll
    for (int a = 0; a < 6; a++) Jko[a] = make_double2(Jk[2 * a], Jk[2 * a + 1]);
    double We[18];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) We[3 * r + c] = w * (Jk[r] * Jp[c] + Jk[6 + r] * Jp[3 + c]);
#pragma unroll
    for (int a = 0; a < 9; a++) Wo[a] = make_double2(We[2 * a], We[2 * a + 1]);
}
__device__ __forceinline__ void k_ba_hll_body(const BaDev& D, int bid) {
    if (ba_skip(D)) return;
    const int p = bid * blockDim.x + threadIdx.x;
    if (p >= D.NP) return;
    double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int k = D.pt_start[p]; k < D.pt_start[p + 1]; k++) {
        if (D.level[k] != 0) continue;
        const double* Jp = D.Jp + 6 * (size_t)k;
        const double w = D.wgt[k], e0 = D.err[2 * k], e1 = D.err[2 * k + 1];
        H[0] += w * (Jp[0] * Jp[0] + Jp[3] * Jp[3]); H[1] += w * (Jp[0] * Jp[1] + Jp[3] * Jp[4]); H[2] += w * (Jp[0] * Jp[2] + Jp[3] * Jp[5]);
        H[3] += w * (Jp[1] * Jp[1] + Jp[4] * Jp[4]); H[4] += w * (Jp[1] * Jp[2] + Jp[4] * Jp[5]); H[5] += w * (Jp[2] * Jp[2] + Jp[5] * Jp[5]);
        for (int a = 0; a < 3; a++) b[a] -= w * (Jp[a] * e0 + Jp[3 + a] * e1);
    }
    double* Ho = D.Hll + (size_t)p * 9;
    Ho[0] = H[0]; Ho[1] = H[1]; Ho[2] = H[2]; Ho[3] = H[1]; Ho[4] = H[3]; Ho[5] = H[4]; Ho[6] = H[2]; Ho[7] = H[4]; Ho[8] = H[5];
    for (int a = 0; a < 3; a++) D.bl[(size_t)p * 3 + a] = b[a];
}

// one workgroup per local key frame: sum of Jk^T w Jk / Jk^T w e over its active edges -> the 6x6 reprojection block of Hpp, bp
__device__ __forceinline__ void k_ba_hpp_body(const BaDev& D, int i = blockIdx.x) {
    if (ba_skip(D)) return;
    __shared__ double s_red[4][27];
    const int t = threadIdx.x, rows = D.rows;
    double a[27];
#pragma unroll
    for (int k = 0; k < 27; k++) a[k] = 0;
    for (int q = D.kf_start[i] + t; q < D.kf_start[i + 1]; q += blockDim.x) {
        const int k = D.kf_list[q];
        if (D.level[k] != 0) continue;
        const double* J = D.Jk + (size_t)6 * rows * k; const double* e = D.err + (size_t)rows * k; const double w = D.wgt[k];
        for (int row = 0; row < rows; row++) {
            const double* Jr = J + 6 * row; const double er = e[row];
            int c = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int cc = r; cc < 6; cc++) a[c++] += w * (Jr[r] * Jr[cc]);
#pragma unroll
            for (int r = 0; r < 6; r++) a[21 + r] -= w * (Jr[r] * er);
        }
    }
#pragma unroll
    for (int k = 0; k < 27; k++) {
        double v = a[k];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
        if ((t & 63) == 0) s_red[t >> 6][k] = v;
    }
    __syncthreads();
    if (t < 27) {
        const double v = s_red[0][t] + s_red[1][t] + s_red[2][t] + s_red[3][t];
        const int base = D.pose_dim * i, n = D.np;
        if (t < 21) {
            int kk = 0, rr = 0, cc = 0;
            for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) { if (kk == t) { rr = r; cc = c; } kk++; }
            atomicAdd(&D.Hpp[(size_t)(base + ba_loc(D, rr)) * n + base + ba_loc(D, cc)], v);
            if (rr != cc) atomicAdd(&D.Hpp[(size_t)(base + ba_loc(D, cc)) * n + base + ba_loc(D, rr)], v);
        } else atomicAdd(&D.bp[base + ba_loc(D, t - 21)], v);
    }
}

// one workgroup per local key frame i: IMU factor (pred(i) -> i) and bias factor
__device__ __forceinline__ void k_ba_imu_body(const BaDev& D, int i = blockIdx.x) {
    if (ba_skip(D)) return;
    __shared__ double J[9 * 21], OJ[9 * 21], e[9];
    __shared__ int map[21];
    __shared__ double s_w;
    const int t = threadIdx.x, j = ba_pred(D, i), n = D.np;
    if (j < 0) return;
    const double* ki = D.kf + (size_t)i * 22; const double* kj = D.kf + (size_t)j * 22;
    if (t == 0) {
        pvr_edge(ld_pvr(kj), ld_pvr(ki), ld3(kj + 16), ld3(kj + 19), D.preint + (size_t)i * 142, ld3(D.gw), e, J);
        double chi = 0;
        for (int a = 0; a < 9; a++) { double s = 0; for (int b = 0; b < 9; b++) s += D.info_pvr[i * 81 + a * 9 + b] * e[b]; chi += e[a] * s; }
        double r0, r1; huber(chi, (double)(float)sqrt(21.666), &r0, &r1); s_w = r1;
        for (int c = 0; c < 9; c++) { map[c] = j < D.W ? 12 * j + c : -1; map[9 + c] = 12 * i + c; }
        for (int c = 0; c < 3; c++) map[18 + c] = j < D.W ? 12 * j + 9 + c : -1;
        // bias factor
        const d3 eb = (ld3(ki + 13) + ld3(ki + 19)) - (ld3(kj + 13) + ld3(kj + 19));
        const double binfo = 1.0 / D.acc_bias_rw2 / D.preint[(size_t)i * 142 + 141];
        huber(binfo * dot3(eb, eb), (double)(float)sqrt(16.812), &r0, &r1);
        const double wb = r1 * binfo, ev[3] = {eb.x, eb.y, eb.z};
        for (int c = 0; c < 3; c++) {
            const int ic = 12 * i + 9 + c, jc = j < D.W ? 12 * j + 9 + c : -1;
            atomicAdd(&D.Hpp[(size_t)ic * n + ic], wb); atomicAdd(&D.bp[ic], -wb * ev[c]);
            if (jc >= 0) { atomicAdd(&D.Hpp[(size_t)jc * n + jc], wb); atomicAdd(&D.Hpp[(size_t)ic * n + jc], -wb); atomicAdd(&D.Hpp[(size_t)jc * n + ic], -wb); atomicAdd(&D.bp[jc], wb * ev[c]); }
        }
    }
    __syncthreads();
    for (int q = t; q < 189; q += blockDim.x) { const int r = q / 21, c = q % 21; double s = 0; for (int k = 0; k < 9; k++) s += D.info_pvr[i * 81 + r * 9 + k] * J[k * 21 + c]; OJ[q] = s; }
    __syncthreads();
    const double w = s_w;
    for (int q = t; q < 441 + 21; q += blockDim.x) {
        if (q < 441) { const int r = q / 21, c = q % 21; if (map[r] >= 0 && map[c] >= 0) { double s = 0; for (int k = 0; k < 9; k++) s += J[k * 21 + r] * OJ[k * 21 + c]; atomicAdd(&D.Hpp[(size_t)map[r] * n + map[c]], w * s); } }
        else { const int r = q - 441; if (map[r] >= 0) { double s = 0; for (int k = 0; k < 9; k++) s += OJ[k * 21 + r] * e[k]; atomicAdd(&D.bp[map[r]], -w * s); } }
    }
}

// S = Hpp + lambda I, bs = bp, written with the padded leading dimension ld (a multiple of 16: identity on the padding)
__device__ __forceinline__ void k_ba_init_reduced_body(const BaDev& D, double lambda_arg) {
    if (ba_skip(D)) return;
    const double lambda = ba_lambda(D, lambda_arg);
    const int q = blockIdx.x * blockDim.x + threadIdx.x, n = D.np, ld = D.ld;
    if (q < ld * ld) {
        const int i = q / ld, j = q - i * ld;
        D.S[q] = (i < n && j < n) ? D.Hpp[(size_t)i * n + j] + (i == j ? lambda : 0.0) : (i == j ? 1.0 : 0.0);
    }
    if (q < ld) D.bs[q] = q < n ? D.bp[q] : 0.0;
}
__device__ __forceinline__ void k_ba_max_diag_body(const BaDev& D) {
    if (ba_skip(D)) return;
    __shared__ double s_red[4];
    double m = 0;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < D.np + 3 * D.NP; q += gridDim.x * blockDim.x)
        m = fmax(m, fabs(q < D.np ? D.Hpp[(size_t)q * D.np + q] : D.Hll[(size_t)((q - D.np) / 3) * 9 + ((q - D.np) % 3) * 4]));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmax(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
        // fmax over non-negative doubles == max over their bit patterns as unsigned integers
        atomicMax(reinterpret_cast<unsigned long long*>(&D.scal[3]), (unsigned long long)__double_as_longlong(t));
    }
}

// one thread per point: Dinv = (Hll + lambda I)^-1 and Dinv * bl
__device__ __forceinline__ void k_ba_dinv_body(const BaDev& D, double lambda_arg, int bid = blockIdx.x) {
    if (ba_skip(D)) return;
    const double lambda = ba_lambda(D, lambda_arg);
    const int p = bid * blockDim.x + threadIdx.x;
    if (p >= D.NP) return;
    const double* H = D.Hll + (size_t)p * 9;
    const double a = H[0] + lambda, b = H[1], c = H[2], d = H[4] + lambda, e = H[5], f = H[8] + lambda;
    const double det = a * (d * f - e * e) - b * (b * f - c * e) + c * (b * e - c * d), id = 1.0 / det;
    const double i00 = (d * f - e * e) * id, i01 = (c * e - b * f) * id, i02 = (b * e - c * d) * id, i11 = (a * f - c * c) * id, i12 = (b * c - a * e) * id, i22 = (a * d - b * b) * id;
    double* Di = D.Dinv + (size_t)p * 9;
    Di[0] = i00; Di[1] = i01; Di[2] = i02; Di[3] = i01; Di[4] = i11; Di[5] = i12; Di[6] = i02; Di[7] = i12; Di[8] = i22;
    const double b0 = D.bl[3 * p], b1 = D.bl[3 * p + 1], b2 = D.bl[3 * p + 2];
    D.db[3 * p] = i00 * b0 + i01 * b1 + i02 * b2; D.db[3 * p + 1] = i01 * b0 + i11 * b1 + i12 * b2; D.db[3 * p + 2] = i02 * b0 + i12 * b1 + i22 * b2;
}
// Schur complement of the point block (block_solver.hpp:381-432), gathered per key-frame PAIR: wavefront (a, b), a >= b, owns the 6 x 6 block
// S[a][b] -= sum_p (W_pa Dinv_p) W_pb^T over the points p both observe (W = wgt Jk^T Jp from the linearisation, Dinv from k_ba_dinv; the
// Cholesky never reads above the diagonal) and, on the diagonal pairs, bs[a] -= sum_p W_pa Dinv_p bl_p. The pair's list of (edge of a,
// edge of b) comes from the host (ba_build_pairs: the graph is fixed over the solve; an edge gated out after the first phase is skipped by
// its level). One lane per list entry: both W blocks and Dinv as 16-byte loads, 162 FMA, 36 (+ 6) running sums in registers; the lanes' sums
// meet through an LDS transpose (row k = the 64 lanes' k-th sums, padded against bank conflicts), lane k adds row k and updates its
// element. One writer per block: no atomics, neither in LDS nor on S. (The round-2 form walked a key frame's observations with one thread
// each and added every product to an LDS block row by ds_add_f64 — 256 threads on the 36 addresses of each partner block: 0.6 atomics
// per cycle and CU, 475 us for 128 windows.)
#define BA_SCHUR_VALS 42
__device__ __forceinline__ void k_ba_schur_body(const BaDev& D, int pid) {
    if (ba_skip(D)) return;
    __shared__ double s_t[BA_SCHUR_VALS][65];
    const int lane = threadIdx.x;
    int a = (int)((sqrt(8.0 * (double)pid + 1.0) - 1.0) * 0.5);
    while (a * (a + 1) / 2 > pid) a--;
    while ((a + 1) * (a + 2) / 2 <= pid) a++;
    const int b = pid - a * (a + 1) / 2;
    const int e0 = D.pr_start[pid], e1 = D.pr_start[pid + 1];
    double acc[BA_SCHUR_VALS];
#pragma unroll
    for (int k = 0; k < BA_SCHUR_VALS; k++) acc[k] = 0.0;
    for (int i = e0 + lane; i < e1; i += 64) {
        const int2 ent = reinterpret_cast<const int2*>(D.pr_ent)[i];
        const int ea = ent.x, eb = ent.y;
        if (D.level[ea] != 0 || D.level[eb] != 0) continue;
        const int p = D.e_pt[ea];
        double wa[18], wb[18], di[9];
        {
            const double2* A2 = reinterpret_cast<const double2*>(D.We + (size_t)18 * ea); const double2* B2 = reinterpret_cast<const double2*>(D.We + (size_t)18 * eb);
#pragma unroll
            for (int k = 0; k < 9; k++) { const double2 u = A2[k], v = B2[k]; wa[2 * k] = u.x; wa[2 * k + 1] = u.y; wb[2 * k] = v.x; wb[2 * k + 1] = v.y; }
            const double* Dp = D.Dinv + (size_t)9 * p;
#pragma unroll
            for (int k = 0; k < 9; k++) di[k] = Dp[k];
        }
        double y[18];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) y[3 * r + c] = wa[3 * r] * di[3 * c] + wa[3 * r + 1] * di[3 * c + 1] + wa[3 * r + 2] * di[3 * c + 2];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) acc[6 * r + c] += y[3 * r] * wb[3 * c] + y[3 * r + 1] * wb[3 * c + 1] + y[3 * r + 2] * wb[3 * c + 2];
        if (ea == eb) {                                              // once per observation of a: W_pa (Dinv_p bl_p)
            const double d0 = D.db[3 * p], d1 = D.db[3 * p + 1], d2 = D.db[3 * p + 2];
#pragma unroll
            for (int r = 0; r < 6; r++) acc[36 + r] += wa[3 * r] * d0 + wa[3 * r + 1] * d1 + wa[3 * r + 2] * d2;
        }
    }
#pragma unroll
    for (int k = 0; k < BA_SCHUR_VALS; k++) s_t[k][lane] = acc[k];
    __syncthreads();
    if (lane < BA_SCHUR_VALS) {
        double v = 0.0;
#pragma unroll 16
        for (int l = 0; l < 64; l++) v += s_t[lane][l];
        const int pd = D.pose_dim;
        if (lane < 36) {
            const int row = pd * a + ba_loc(D, lane / 6), col = pd * b + ba_loc(D, lane % 6);
            if (col <= row && v != 0.0) D.S[(size_t)row * D.ld + col] -= v;
        } else if (a == b) D.bs[pd * a + ba_loc(D, lane - 36)] -= v;
    }
}

// Dense Cholesky solve S xp = bs of the reduced system by ONE 512-thread workgroup, blocked by 16 (n <= 240, padded to ld): the lower
// block triangle lives in the registers of seven waves as 16x16 tiles in the layout of v_mfma_f64_16x16x4_f64 (operands A[i][k]: lane
// 16k+i, B[k][j]: lane 16k+j; result row = lane/16 + 4*reg, col = lane%16 — probed on gfx950, tools/ubench/mfma_f64_layout.hip), the
// eighth wave factors the 16x16 diagonal blocks one block column ahead (see the body). The right-hand side lives in LDS and its forward
// substitution L y = bs rides along with the panels; the backward substitution uses the inverted diagonal blocks and is a 16x16 mat-vec
// + a rank-16 update per block. scal[2] = 1 on success; a pivot that is not positive and finite fails the solve like the reference's
// LLT (linear_solver_eigen.h / Eigen info()).
typedef double v4d __attribute__((ext_vector_type(4)));
#define BA_CHOL_THREADS 512                  // 8 waves: 7 tile owners + the factor wave
#define BA_CHOL_WAVES (BA_CHOL_THREADS / 64)
#define BA_CHOL_OWNERS (BA_CHOL_WAVES - 1)
#define BA_MAX_TILES_PER_WAVE 18             // ceil((15 * 16 / 2) / 7) register tiles (144 registers) per owner wave at ld = 240 (the lower block triangle; the right-hand side lives in LDS)
// lane broadcast of a double and a full-precision reciprocal square root (hardware estimate + two Newton steps) for the factorisation's
// diagonal blocks
__device__ __forceinline__ double ba_readlane(double v, int l) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], l); u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return u.d;
}
__device__ __forceinline__ double ba_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);
    y = fma(0.5 * y, fma(-(d * y), y, 1.0), y);
    return y;
}
#ifdef VIORB_CHOL_TIMING
#define CT_LAP(i) do { if ((threadIdx.x & 63) == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ct_[i] += now_ - ct_last; ct_last = now_; } } while (0)     /* per wave, no barrier of its own */
#else
#define CT_LAP(i) do { } while (0)
#endif
// dynamic LDS of the Cholesky kernels: two panel buffers [256][17], the inverted diagonal block, the next diagonal tile, y
#define BA_CHOL_LDS_BYTES ((2 * 256 * 17 + 2 * 16 * 17 + 256) * sizeof(double))
__device__ __forceinline__ void k_ba_chol_solve_body(const BaDev& D, double* s_dyn) {
    if (ba_skip(D)) return;
#ifdef VIORB_CHOL_TIMING
    unsigned long long ct_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ct_last = __builtin_amdgcn_s_memtime();
#endif
    typedef double row17[17];
    row17* s_P0 = reinterpret_cast<row17*>(s_dyn);               // [256][17] panel of an even block column
    row17* s_P1 = s_P0 + 256;                                    // ... of an odd one
    row17* s_L = s_P1 + 256;                                     // [16][17] L11^-1 of the current block column
    row17* s_D = s_L + 16;                                       // [16][17] the next diagonal tile, as published
    double* s_y = reinterpret_cast<double*>(s_D + 16);           // [256]
    __shared__ int s_ok;
    const int n = D.np, ld = D.ld, nb = ld >> 4, t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int rows = ld + 16;                                    // + the block row that carries bs^T in its first row
    double* A = D.S;
    double* inv = A + (size_t)rows * ld;                         // [nb][16][16] inverted diagonal blocks
    if (t == 0) s_ok = 1;
    for (int i = t; i < 256; i += blockDim.x) s_y[i] = i < ld ? D.bs[i] : 0.0;             // the right-hand side; the forward substitution rides along with the panels
    // Two roles. Waves 0 .. BA_CHOL_OWNERS-1 own the lower block triangle in REGISTERS for the whole factorisation:
    // tile idx = wv + BA_CHOL_OWNERS u of the row-major enumeration (I, J <= I) is slot u of wave
    // wv, 4 doubles per lane in the MFMA accumulator layout (row = lane / 16 + 4 reg, col = lane % 16). The last wave owns nothing and
    // factors the diagonal blocks, ONE BLOCK COLUMN AHEAD of the owners. Per block column kb, three phases between barriers:
    //   P   the owners multiply their panel tiles of column kb by L11^-T (v_mfma_f64_16x16x4; operands in LDS) -> L21, final;
    //   3a  they apply panel kb's rank-16 update to their tiles of column kb + 1 FIRST and publish those to LDS (diagonal tile, next panel);
    //   3b  they update the rest of the trailing matrix WHILE the factor wave turns the published diagonal tile into L11 and L11^-1.
    // No global-memory round trip inside the loop; the factorisation of a diagonal block (a chain of 16 dependent 1/sqrt) hides behind
    // the trailing update. The two roles run separate loops with the same barriers, so the factor wave's ~100 live registers (block row,
    // inverse row, broadcasts) and the owners' 160 accumulator registers never have to fit one allocation. L and the solved rhs row are
    // written to global memory for the backward substitution.
    auto factor_block = [&](int kb) {
        // Lane li holds row li of the (symmetrised) block. A column step's dependent chain is pivot -> 1/sqrt (rsq + two Newton steps) ->
        // s = a_ij / d -> up to 15 independent FMAs; the broadcast of row j (v_readlane from lane j's registers: A[j][k] = A[k][j] is the
        // multiplier column k needs) does not depend on it. L11^-1 follows by a right-looking forward substitution with lane c on column
        // c: L goes through LDS once and comes back as broadcast reads that do not depend on the substitution's chain (two dependent
        // operations per row). (Building the inverse inside the factorisation loop doubled its v_readlane traffic: 8.8 k cycles per
        // block; row broadcasts through LDS inside the loop 11.8 k; the first form, broadcasting l_kj after the chain, 6.5 k without inverse.)
        const int li = lane & 15, k0 = kb << 4;
        double a[16], rinv[16];
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = c <= li ? s_D[li][c] : s_D[c][li];
        bool good = true;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const double d = ba_readlane(a[j], j);
            double u[16];
#pragma unroll
            for (int k = j + 1; k < 16; k++) u[k] = ba_readlane(a[k], j);
            const bool gj = (d > 0) && isfinite(d);
            good = good && gj;
            const double ri = ba_rsqrt(gj ? d : 1.0);
            rinv[j] = ri;
            const double sj = a[j] * (ri * ri);                                         // l_ij / l_jj
            a[j] = a[j] * ri;                                                           // l_ij (rows >= j; the rest never reaches a result)
#pragma unroll
            for (int k = j + 1; k < 16; k++) a[k] = fma(-sj, u[k], a[k]);
        }
        if (!good && lane == 0) s_ok = 0;
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 16; c++) {
                s_D[li][c] = a[c];                                                      // L11 (lower triangle valid)
                if (c <= li) A[(size_t)(k0 + li) * ld + k0 + c] = a[c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // X = L11^-1, column li per lane: x_u = acc_u / l_uu, then acc_i -= l_iu x_u for the rows below
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = i == li ? 1.0 : 0.0;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            x[u] *= rinv[u];
#pragma unroll
            for (int i = u + 1; i < 16; i++) x[i] = fma(-s_D[i][u], x[u], x[i]);
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                s_L[i][li] = x[i];                                                      // L11^-1 (lower triangular: x[i] = 0 for i < li)
                inv[(size_t)kb * 256 + i * 16 + li] = x[i];                             // for the backward substitution
            }
        }
    };
    if (wv == BA_CHOL_OWNERS) {
        __syncthreads();                                         // column 0 is published
        factor_block(0);
        __syncthreads();
        for (int kb = 0; kb < nb; kb++) {
            __syncthreads();                                     // (owners: P)
            __syncthreads();                                     // (owners: 3a) — the next diagonal tile is in s_D
            CT_LAP(1);
            if (kb + 1 < nb) factor_block(kb + 1);
            CT_LAP(2);
            __syncthreads();                                     // (owners: 3b)
            CT_LAP(3);
        }
    } else {
        v4d acc[BA_MAX_TILES_PER_WAVE]; int tI[BA_MAX_TILES_PER_WAVE], tJ[BA_MAX_TILES_PER_WAVE];
        const int ntiles_all = nb * (nb + 1) / 2;
#pragma unroll
        for (int u = 0; u < BA_MAX_TILES_PER_WAVE; u++) {
            const int idx = wv + BA_CHOL_OWNERS * u;
            tI[u] = -1; tJ[u] = -1;
            acc[u] = (v4d){0.0, 0.0, 0.0, 0.0};
            if (idx < ntiles_all) {
                int I = 0, rem = idx;
                while (rem > I) { rem -= I + 1; I++; }
                const int J = rem;
                tI[u] = I; tJ[u] = J;
#pragma unroll
                for (int r = 0; r < 4; r++) acc[u][r] = A[(size_t)(I * 16 + (lane >> 4) + 4 * r) * ld + J * 16 + (lane & 15)];
            }
        }
        // publish the tiles of block column c: the diagonal tile to s_D, the tiles below it to the panel buffer of that column's parity
        auto publish = [&](int c) {
            row17* Pn = (c & 1) ? s_P1 : s_P0;
#pragma unroll
            for (int u = 0; u < BA_MAX_TILES_PER_WAVE; u++) {
                if (tJ[u] != c) continue;
                if (tI[u] == c) {
#pragma unroll
                    for (int r = 0; r < 4; r++) s_D[(lane >> 4) + 4 * r][lane & 15] = acc[u][r];
                } else {
                    const int rb = (tI[u] - c - 1) * 16;
#pragma unroll
                    for (int r = 0; r < 4; r++) Pn[rb + (lane >> 4) + 4 * r][lane & 15] = acc[u][r];
                }
            }
        };
        publish(0);
        __syncthreads();
        __syncthreads();                                         // (factor wave: block 0)
        double yn = 0;
        for (int kb = 0; kb < nb; kb++) {
            const int k0 = kb << 4;
            row17* Pc = (kb & 1) ? s_P1 : s_P0;
            // P: L21 = A21 L11^-T per owned tile of block column kb of block column kb: A operand = the published tile, B[k][j] =
            //    L11^-1[j][k]; the result overwrites the tile's rows of the panel buffer (they belong to this wave alone in this phase). (A thread per row solving against L11 was a chain of 120 dependent FMAs fed by
            //    LDS reads: 7.3 k cycles per panel.)
#pragma unroll
            for (int u = 0; u < BA_MAX_TILES_PER_WAVE; u++) {
                if (tJ[u] != kb || tI[u] == kb) continue;
                const int rb = (tI[u] - kb - 1) * 16;
                v4d xt = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kc = 0; kc < 4; kc++)
                    xt = __builtin_amdgcn_mfma_f64_16x16x4f64(Pc[rb + (lane & 15)][4 * kc + (lane >> 4)], s_L[lane & 15][4 * kc + (lane >> 4)], xt, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) Pc[rb + (lane >> 4) + 4 * r][lane & 15] = xt[r];
            }
            if (t < 16) {                                        // forward substitution of the right-hand side: y_kb <- L11^-1 y_kb
                double yk = 0;
#pragma unroll
                for (int c = 0; c < 16; c++) yk += s_L[t][c] * s_y[k0 + c];
                yn = yk;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (t < 16) s_y[k0 + t] = yn;
            CT_LAP(1);
            __syncthreads();
            // 3a: y_I -= L21_I y_kb for the rows below (one thread per row of the panel), the rank-16 update of the tiles of block column kb + 1,
            //     then their publication
            if (t < ld - k0 - 16) {
                double yv = s_y[k0 + 16 + t];
                double* Arow = A + (size_t)(k0 + 16 + t) * ld + k0;      // L21 goes to global memory here, a row per thread (per-tile stores in P kept
#pragma unroll                                                           // 4 hoisted 64-bit addresses per tile alive: spills)
                for (int c = 0; c < 16; c++) { const double l = Pc[t][c]; yv -= l * s_y[k0 + c]; Arow[c] = l; }
                s_y[k0 + 16 + t] = yv;
            }
#pragma unroll
            for (int u = 0; u < BA_MAX_TILES_PER_WAVE; u++) {
                if (tJ[u] != kb + 1) continue;                   // tJ > kb implies tI > kb
                const int Io = (tI[u] - kb - 1) * 16 + (lane & 15), Jo = (tJ[u] - kb - 1) * 16 + (lane & 15);
#pragma unroll
                for (int kc = 0; kc < 4; kc++)
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pc[Io][4 * kc + (lane >> 4)], Pc[Jo][4 * kc + (lane >> 4)], acc[u], 0, 0, 0);
            }
            if (kb + 1 < nb) publish(kb + 1);
            CT_LAP(2);
            __syncthreads();
            // 3b: the rest of the trailing matrix, beside the factor wave
#pragma unroll
            for (int u = 0; u < BA_MAX_TILES_PER_WAVE; u++) {
                if (tJ[u] <= kb + 1) continue;
                const int Io = (tI[u] - kb - 1) * 16 + (lane & 15), Jo = (tJ[u] - kb - 1) * 16 + (lane & 15);
#pragma unroll
                for (int kc = 0; kc < 4; kc++)
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pc[Io][4 * kc + (lane >> 4)], Pc[Jo][4 * kc + (lane >> 4)], acc[u], 0, 0, 0);
            }
            CT_LAP(3);
            __syncthreads();
            CT_LAP(4);
        }
    }
    __syncthreads();
    CT_LAP(0);
    // y = L^-1 bs now sits in s_y; backward substitution L^T x = y, 16 unknowns at a time
    for (int kb = nb - 1; kb >= 0; kb--) {
        const int k0 = kb << 4;
        double xi = 0;
        if (t < 16) {                                            // x_blk = L11^-T y_blk
#pragma unroll
            for (int c = 0; c < 16; c++) xi += inv[(size_t)kb * 256 + c * 16 + t] * s_y[k0 + c];
        }
        __syncthreads();
        if (t < 16) s_y[k0 + t] = xi;
        __syncthreads();
        if (t < k0) {
            double v = s_y[t];
#pragma unroll
            for (int c = 0; c < 16; c++) v -= A[(size_t)(k0 + c) * ld + t] * s_y[k0 + c];
            s_y[t] = v;
        }
        __syncthreads();
    }
    CT_LAP(5);
    for (int i = t; i < n; i += blockDim.x) D.xp[i] = s_ok ? s_y[i] : 0.0;
#ifdef VIORB_CHOL_TIMING
    if (lane == 0 && (wv == 0 || wv == BA_CHOL_OWNERS) && blockIdx.x == 0) printf("chol cycles wave %d: [0] %llu [1] %llu [2] %llu [3] %llu [4] %llu [5] %llu [6] %llu\n", wv, ct_[0], ct_[1], ct_[2], ct_[3], ct_[4], ct_[5], ct_[6]);
#endif
    if (t == 0) { D.scal[2] = s_ok ? 1.0 : 0.0; D.scal[1] = 0.0; D.scal[0] = 0.0; }      // [1], [0]: accumulators of k_ba_backsub and k_ba_*_errors, which follow
}

// xl = Dinv (bl - W^T xp) per point, and the LM scale term sum x (lambda x + b)
__device__ __forceinline__ void k_ba_backsub_body(const BaDev& D, double lambda_arg) {
    if (ba_skip(D)) return;
    const double lambda = ba_lambda(D, lambda_arg);
    __shared__ double s_red[4];
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double sc = 0;
    if (p < D.NP) {
        double c0 = D.bl[3 * p], c1 = D.bl[3 * p + 1], c2 = D.bl[3 * p + 2];
        for (int k = D.pt_start[p]; k < D.pt_start[p + 1]; k++) {
            if (D.level[k] != 0 || D.e_kf[k] >= D.W) continue;
            const double* Wk = D.We + (size_t)18 * k;
            const int ba = D.pose_dim * D.e_kf[k];
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const double x = D.xp[ba + ba_loc(D, r)];
                c0 -= Wk[3 * r] * x; c1 -= Wk[3 * r + 1] * x; c2 -= Wk[3 * r + 2] * x;
            }
        }
        const double* Di = D.Dinv + (size_t)p * 9;
        const double x0 = Di[0] * c0 + Di[1] * c1 + Di[2] * c2, x1 = Di[3] * c0 + Di[4] * c1 + Di[5] * c2, x2 = Di[6] * c0 + Di[7] * c1 + Di[8] * c2;
        D.xl[3 * p] = x0; D.xl[3 * p + 1] = x1; D.xl[3 * p + 2] = x2;
        sc = x0 * (lambda * x0 + D.bl[3 * p]) + x1 * (lambda * x1 + D.bl[3 * p + 1]) + x2 * (lambda * x2 + D.bl[3 * p + 2]);
    }
    if (blockIdx.x == 0) for (int q = threadIdx.x; q < D.np; q += blockDim.x) sc += D.xp[q] * (lambda * D.xp[q] + D.bp[q]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sc += __shfl_xor(sc, d);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&D.scal[1], s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

__device__ __forceinline__ void k_ba_update_body(const BaDev& D) {
    if (ba_skip(D)) return;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < D.W) {
        double* k = D.kf + (size_t)q * 22;
        for (int c = 0; c < 22; c++) D.kf_bak[(size_t)q * 22 + c] = k[c];      // what a rejected LM trial goes back to (k_ba_restore)
        const pvr s = inc_small_pvr(ld_pvr(k), D.xp + 12 * q);
        st_pvr(k, s);
        for (int c = 0; c < 3; c++) k[19 + c] += D.xp[12 * q + 9 + c];
    }
    if (q < D.NP) for (int c = 0; c < 3; c++) { D.pt_bak[3 * q + c] = D.pt[3 * q + c]; D.pt[3 * q + c] += D.xl[3 * q + c]; }
}
// a rejected trial: local key frames and points back to what k_ba_update / k_ba_se3_update saved
__device__ __forceinline__ void k_ba_restore_body(const BaDev& D) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x, stride = D.kf_stride;
    if (q < D.W) for (int c = 0; c < stride; c++) D.kf[(size_t)q * stride + c] = D.kf_bak[(size_t)q * stride + c];
    if (q < D.NP) for (int c = 0; c < 3; c++) D.pt[3 * q + c] = D.pt_bak[3 * q + c];
}

// chi2 / depth gate on every edge (stale error on excluded edges, fresh depth), Optimizer.cc:2037-2051 and :2105-2118
__device__ __forceinline__ void k_ba_gate_body(const BaDev& D, uint8_t* out, int set_level) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= D.NE) return;
    d3 Pc, Paux; m33 RT; cam_t K;
    ba_edge_geom(D, k, D.kf, D.pt, Pc, RT, Paux, K);
    const double chi = D.e_obs[3 * k + 2] * (D.err[2 * k] * D.err[2 * k] + D.err[2 * k + 1] * D.err[2 * k + 1]);
    const int bad = (chi > 5.991 || !(Pc.z > 0.0)) ? 1 : 0;
    if (set_level) { if (bad) D.level[k] = 1; } else out[k] = (uint8_t)bad;
}

// zero Hpp / bp (and the max-diagonal accumulator) for the linearisation of an iteration; first = 1 on the first iteration of a phase
__device__ __forceinline__ void k_ba_clear_body(const BaDev& D, int first, int bid = blockIdx.x, int nblk = gridDim.x) {
    if (ba_skip(D)) return;
    const size_t n2 = (size_t)D.np * D.np;
    for (size_t q = (size_t)bid * blockDim.x + threadIdx.x; q < n2; q += (size_t)nblk * blockDim.x) D.Hpp[q] = 0.0;
    if (bid == 0) {
        for (int q = threadIdx.x; q < D.np; q += blockDim.x) D.bp[q] = 0.0;
        if (threadIdx.x == 0) {
            if (first) { D.scal[3] = 0.0; D.ctl[BA_CTL_CHI] = D.scal[0]; D.ctl[BA_CTL_NBAD] = 0.0; }   // chi2 of the phase's first computeActiveErrors
            D.ctl[BA_CTL_INICHI] = D.ctl[BA_CTL_CHI];
        }
    }
}
// lambda_0 = 1e-5 * max diagonal (:166-180), after k_ba_max_diag on the first iteration of a phase
__device__ __forceinline__ void k_ba_lambda0_body(const BaDev& D) {
    if (ba_skip(D)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) { D.ctl[BA_CTL_LAMBDA] = 1e-5 * D.scal[3]; D.ctl[BA_CTL_NI] = 2.0; }
}
// after the trial's k_ba_*_errors: rho = (chi - chi_trial) / (sum x (lambda x + b) + 1e-3) (:129-132); accepted -> lambda update, stop
// tests; rejected -> HALT = 1 and nothing else changes (the host takes over from ST_AFTER_TRIAL with the scalars as they are)
__device__ __forceinline__ void k_ba_decide_body(const BaDev& D, int last_of_phase) {     // one thread
    if (ba_skip(D)) return;
    double* c = D.ctl;
    double tempChi = __hip_atomic_load(&D.scal[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok2 = D.scal[2] > 0.5;
    if (!ok2) tempChi = 1.7976931348623157e308;
    const double scale = (ok2 ? D.scal[1] : 0.0) + 1e-3;
    const double rho = (c[BA_CTL_CHI] - tempChi) / scale;
    c[BA_CTL_RHO] = rho;
    if (!(rho > 0 && isfinite(tempChi))) { c[BA_CTL_HALT] = 1.0; return; }
    const double t = 2 * rho - 1;
    double alpha = 1. - t * t * t;
    alpha = fmin(alpha, 2. / 3.);
    c[BA_CTL_LAMBDA] *= fmax(1. / 3., alpha); c[BA_CTL_NI] = 2.0;
    const double iniChi = c[BA_CTL_INICHI];
    c[BA_CTL_CHI] = tempChi;
    c[BA_CTL_ITS] += 1.0; c[BA_CTL_IT] += 1.0;
    // "if ((iniChi - currentChi) * 1e3 < iniChi) nBad++ else nBad = 0; if (nBad >= 3) stop" (:154-161)
    double nBad = c[BA_CTL_NBAD];
    if ((iniChi - tempChi) * 1e3 < iniChi) nBad += 1.0; else nBad = 0.0;
    c[BA_CTL_NBAD] = nBad;
    if (nBad >= 3.0 || last_of_phase || ba_abort_requested(D)) c[BA_CTL_HALT] = 2.0;      // this optimize() call is over (or "!terminate()" failed)
}
__global__ void k_ba_decide(BaDev D, int last_of_phase) { if (blockIdx.x == 0 && threadIdx.x == 0) k_ba_decide_body(D, last_of_phase); }


// ---- by-value kernels of the single-window driver ---------------------------------------------------------------------------------
__global__ void k_ba_errors(BaDev D, int mono_kernel) { k_ba_errors_body(D, mono_kernel); }
__global__ void k_ba_lin_points(BaDev D, int mono_kernel) { k_ba_lin_points_body(D, mono_kernel); }
__global__ __launch_bounds__(256) void k_ba_hpp(BaDev D) { k_ba_hpp_body(D); }
__global__ __launch_bounds__(256) void k_ba_imu(BaDev D) { k_ba_imu_body(D); }
__global__ void k_ba_init_reduced(BaDev D, double lambda_arg) { k_ba_init_reduced_body(D, lambda_arg); }
__global__ void k_ba_max_diag(BaDev D) { k_ba_max_diag_body(D); }
__global__ void k_ba_dinv(BaDev D, double lambda_arg) { k_ba_dinv_body(D, lambda_arg); }
__global__ __launch_bounds__(64) void k_ba_schur(BaDev D) { k_ba_schur_body(D, blockIdx.x); }
__global__ __launch_bounds__(BA_CHOL_THREADS) void k_ba_chol_solve(BaDev D) { extern __shared__ __attribute__((aligned(16))) double s_chol[]; k_ba_chol_solve_body(D, s_chol); }
__global__ void k_ba_backsub(BaDev D, double lambda_arg) { k_ba_backsub_body(D, lambda_arg); }
__global__ void k_ba_update(BaDev D) { k_ba_update_body(D); }
__global__ void k_ba_restore(BaDev D) { k_ba_restore_body(D); }
__global__ void k_ba_gate(BaDev D, uint8_t* out, int set_level) { k_ba_gate_body(D, out, set_level); }
__global__ void k_ba_clear(BaDev D, int first) { k_ba_clear_body(D, first); }
__global__ void k_ba_lambda0(BaDev D) { k_ba_lambda0_body(D); }

// ---- lock-step batch: one launch covers every window of a batch (blockIdx.y = window) ------------------------------------------------
// viorb_local_ba_navstate_batch used to give every window its own stream and host-driven LM loop; the streams share four hardware queues,
// so at most four windows' kernel chains really overlapped (436 windows/s for 64 W = 20 windows). Here all windows advance through the
// same ROUND of launches — [chi2 of a new phase] [linearise if the last trial was accepted] [lambda_0 on a phase's first iteration]
// [one LM trial] [decide] [restore a rejected trial] [phase gate] — and the per-window state machine of g2o's Levenberg loop
// (optimization_algorithm_levenberg.cpp:61-164) plus LocalBundleAdjustmentNavState's two optimize() calls (src/Optimizer.cc:2026-2099)
// lives in the window's control block on the device. A round is 15 launches for the whole batch (the k_bab_f_* ones cover several solver steps by block range); the host only counts finished windows
// every few rounds.
enum { BA_B_DONE = 16, BA_B_NEED_CHI, BA_B_NEED_LIN, BA_B_FIRST, BA_B_PHASE, BA_B_MONO, BA_B_QMAX, BA_B_GATE, BA_B_RESTORE, BA_B_ITERS,
       BA_B_ITS0, BA_B_ITS1, BA_B_CHI0, BA_B_CHI1, BA_B_ABORT, BA_B_N = 32 };
#define BA_B_WINDOW() const BaDev& D = Dv[blockIdx.y]; const double* c = D.ctl; if (c[BA_B_DONE] != 0.0 || c[BA_B_ABORT] != 0.0) return
__global__ void k_bab_round_begin(const BaDev* __restrict__ Dv, int nwin) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    const BaDev& D = Dv[w]; double* c = D.ctl;
    if (c[BA_B_DONE] != 0.0 || c[BA_B_ABORT] != 0.0) return;
    if (c[BA_B_NEED_CHI] != 0.0) D.scal[0] = 0.0;                        // accumulator of the phase's first computeActiveErrors
    if (c[BA_B_NEED_LIN] != 0.0 && c[BA_B_FIRST] != 0.0) D.scal[3] = 0.0;  // accumulator of k_ba_max_diag
}
__global__ void k_bab_take_chi(const BaDev* __restrict__ Dv, int nwin) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    const BaDev& D = Dv[w]; double* c = D.ctl;
    if (c[BA_B_DONE] != 0.0 || c[BA_B_ABORT] != 0.0 || c[BA_B_NEED_CHI] == 0.0) return;
    c[BA_CTL_CHI] = D.scal[0]; c[BA_B_NEED_CHI] = 0.0;
}
__global__ void k_bab_max_diag(const BaDev* __restrict__ Dv) { BA_B_WINDOW(); if (c[BA_B_NEED_LIN] == 0.0 || c[BA_B_FIRST] == 0.0) return; k_ba_max_diag_body(D); }
__global__ void k_bab_lambda0(const BaDev* __restrict__ Dv, int nwin) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    const BaDev& D = Dv[w]; double* c = D.ctl;
    if (c[BA_B_DONE] != 0.0 || c[BA_B_ABORT] != 0.0 || c[BA_B_NEED_LIN] == 0.0) return;
    if (c[BA_B_FIRST] != 0.0) { c[BA_CTL_LAMBDA] = 1e-5 * D.scal[3]; c[BA_CTL_NI] = 2.0; c[BA_CTL_NBAD] = 0.0; c[BA_B_FIRST] = 0.0; }
    c[BA_B_NEED_LIN] = 0.0;
}
// The batch's grid is (pairs, windows) linearised and dealt so that all pairs of a window run on ONE XCD (workgroup ids go round-robin over the
// eight XCDs): a window's W blocks (1.6 MB at 10.9 k edges) are read by ~5 pairs each and stay in that XCD's L2 meanwhile.
__global__ __launch_bounds__(64) void k_bab_schur(const BaDev* __restrict__ Dv, int npairs, int nwin) {
    const int L = blockIdx.x, xcd = L & 7, s = L >> 3;
    const int win = (s / npairs) * 8 + xcd, pid = s - (s / npairs) * npairs;
    if (win >= nwin) return;
    const BaDev& D = Dv[win]; const double* c = D.ctl;
    if (c[BA_B_DONE] != 0.0 || c[BA_B_ABORT] != 0.0) return;
    if (pid >= D.W * (D.W + 1) / 2) return;
    k_ba_schur_body(D, pid);
}
__global__ __launch_bounds__(BA_CHOL_THREADS) void k_bab_chol_solve(const BaDev* __restrict__ Dv) { extern __shared__ __attribute__((aligned(16))) double s_chol[]; BA_B_WINDOW(); k_ba_chol_solve_body(D, s_chol); }
// the Levenberg decisions of one trial, per window (:129-161), and the end of an optimize() call
__global__ void k_bab_decide(const BaDev* __restrict__ Dv, int nwin) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    const BaDev& D = Dv[w]; double* c = D.ctl;
    if (c[BA_B_DONE] != 0.0) return;
    if (c[BA_B_ABORT] != 0.0) { c[BA_B_GATE] = 2.0; return; }            // the host saw the stop flag between rounds (this round's trial kernels were skipped): final gate on the state as it is
    const bool abort_now = ba_abort_requested(D);                        // the flag went up during this trial: take the trial's decision, then leave both loops ("&& !terminate()")
    double tempChi = D.scal[0];
    const bool ok2 = D.scal[2] > 0.5;
    if (!ok2) tempChi = 1.7976931348623157e308;
    const double scale = (ok2 ? D.scal[1] : 0.0) + 1e-3;
    const double rho = (c[BA_CTL_CHI] - tempChi) / scale;
    c[BA_CTL_RHO] = rho;
    int qmax = (int)c[BA_B_QMAX];
    if (rho > 0 && isfinite(tempChi)) {
        const double t = 2 * rho - 1;
        double alpha = 1. - t * t * t;
        alpha = fmin(alpha, 2. / 3.);
        c[BA_CTL_LAMBDA] *= fmax(1. / 3., alpha); c[BA_CTL_NI] = 2.0; c[BA_CTL_CHI] = tempChi;
    } else {
        c[BA_CTL_LAMBDA] *= c[BA_CTL_NI]; c[BA_CTL_NI] *= 2.0;
        c[BA_B_RESTORE] = 1.0;
    }
    qmax++;
    if (abort_now) { c[BA_B_GATE] = 2.0; return; }                       // a rejected trial is restored by k_bab_restore before the gate
    if (rho < 0 && qmax < 10) { c[BA_B_QMAX] = qmax; return; }           // another trial of the same iteration (no re-linearisation)
    // the iteration is over
    const int phase = (int)c[BA_B_PHASE];
    c[BA_B_ITS0 + phase] += 1.0; c[BA_B_CHI0 + phase] = c[BA_CTL_CHI];
    bool stop_opt = (qmax == 10 || rho == 0);
    if (!stop_opt) {
        double nBad = c[BA_CTL_NBAD];
        if ((c[BA_CTL_INICHI] - c[BA_CTL_CHI]) * 1e3 < c[BA_CTL_INICHI]) nBad += 1.0; else nBad = 0.0;
        c[BA_CTL_NBAD] = nBad;
        if (nBad >= 3.0) stop_opt = true;
    }
    c[BA_CTL_IT] += 1.0; c[BA_B_QMAX] = 0.0;
    if (stop_opt || c[BA_CTL_IT] >= c[BA_B_ITERS]) c[BA_B_GATE] = phase == 0 ? 1.0 : 2.0;
    else c[BA_B_NEED_LIN] = 1.0;
}
__global__ void k_bab_restore(const BaDev* __restrict__ Dv) { BA_B_WINDOW(); if (c[BA_B_RESTORE] == 0.0) return; k_ba_restore_body(D); }
// after the gate: phase 0 -> second optimize() without the mono kernel (src/Optimizer.cc:2037-2099), phase 1 -> the window is finished
__global__ void k_bab_phase(const BaDev* __restrict__ Dv, int nwin, int* __restrict__ n_done) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    const BaDev& D = Dv[w]; double* c = D.ctl;
    if (c[BA_B_DONE] != 0.0) return;
    c[BA_B_RESTORE] = 0.0;
    if (c[BA_B_GATE] == 1.0) {
        c[BA_B_PHASE] = 1.0; c[BA_B_MONO] = 0.0; c[BA_B_ITERS] = 10.0; c[BA_CTL_IT] = 0.0; c[BA_B_QMAX] = 0.0;
        c[BA_B_NEED_CHI] = 1.0; c[BA_B_NEED_LIN] = 1.0; c[BA_B_FIRST] = 1.0; c[BA_B_GATE] = 0.0;
    } else if (c[BA_B_GATE] == 2.0) {
        c[BA_B_GATE] = 0.0; c[BA_B_DONE] = 1.0;
        atomicAdd(n_done, 1);
    }
}

// ---- vision-only LocalBundleAdjustment (reference src/Optimizer.cc:3980-4311): SE3 key frames (kf = qx qy qz qw tx ty tz of Tcw),
// EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ (Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:66-250). e_obs[k] = u v uRight invSigma2
// (uRight < 0: mono); cam[0..4] = fx fy cx cy bf. Three residual rows per edge (the third is zero for mono edges).
__device__ __forceinline__ se3q ba_ld_se3(const double* k) { se3q s; s.r = mkq(k[0], k[1], k[2], k[3]); s.t = mk3(k[4], k[5], k[6]); return s; }
__device__ __forceinline__ void k_ba_se3_errors_body(const BaDev& D, int kernels) {
    if (ba_skip(D)) return;
    __shared__ double s_red[8];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double c = 0;
    if (k < D.NE && D.level[k] == 0) {
        const double* ob = D.e_obs + 4 * (size_t)k;
        double e[3];
        se3_edge(ba_ld_se3(D.kf + (size_t)D.e_kf[k] * 7), ld3(D.pt + (size_t)D.e_pt[k] * 3), ob[0], ob[1], ob[2], D.cam[0], D.cam[1], D.cam[2], D.cam[3], D.cam[4], false, e, nullptr);
        D.err[3 * k] = e[0]; D.err[3 * k + 1] = e[1]; D.err[3 * k + 2] = e[2];
        const double chi = ob[3] * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        double r0 = chi, r1;
        if (kernels) huber(chi, (double)(float)sqrt(ob[2] < 0 ? 5.991 : 7.815), &r0, &r1);
        c = r0;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += s_red[w]; atomicAdd(&D.scal[0], t); }
}
__device__ __forceinline__ void k_ba_se3_lin_points_body(const BaDev& D, int kernels) {
    if (ba_skip(D)) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.NP) return;
    double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    const double fx = D.cam[0], fy = D.cam[1], bf = D.cam[4];
    for (int k = D.pt_start[p]; k < D.pt_start[p + 1]; k++) {
        if (D.level[k] != 0) continue;
        const double* ob = D.e_obs + 4 * (size_t)k;
        const bool stereo = !(ob[2] < 0);
        const se3q T = ba_ld_se3(D.kf + (size_t)D.e_kf[k] * 7);
        const d3 pc = se3_map(T, ld3(D.pt + (size_t)p * 3));
        const m33 R = qmat(T.r);
        const double x = pc.x, y = pc.y, z = pc.z, z_2 = z * z;
        double Jp[9], Jk[18];
        const double Rr[9] = {R.a00, R.a01, R.a02, R.a10, R.a11, R.a12, R.a20, R.a21, R.a22};
        if (!stereo) {
            const double t0[3] = {fx, 0, -x / z * fx}, t1[3] = {0, fy, -y / z * fy};
            for (int c = 0; c < 3; c++) {
                double s0 = 0, s1 = 0;
                for (int q = 0; q < 3; q++) { s0 += (-1. / z * t0[q]) * Rr[3 * q + c]; s1 += (-1. / z * t1[q]) * Rr[3 * q + c]; }
                Jp[c] = s0; Jp[3 + c] = s1; Jp[6 + c] = 0;
            }
        } else {
            for (int c = 0; c < 3; c++) {
                Jp[c] = -fx * Rr[c] / z + fx * x * Rr[6 + c] / z_2;
                Jp[3 + c] = -fy * Rr[3 + c] / z + fy * y * Rr[6 + c] / z_2;
                Jp[6 + c] = Jp[c] - bf * Rr[6 + c] / z_2;
            }
        }
        Jk[0] = x * y / z_2 * fx; Jk[1] = -(1 + (x * x / z_2)) * fx; Jk[2] = y / z * fx; Jk[3] = -1. / z * fx; Jk[4] = 0; Jk[5] = x / z_2 * fx;
        Jk[6] = (1 + y * y / z_2) * fy; Jk[7] = -x * y / z_2 * fy; Jk[8] = -x / z * fy; Jk[9] = 0; Jk[10] = -1. / z * fy; Jk[11] = y / z_2 * fy;
        if (stereo) { Jk[12] = Jk[0] - bf * y / z_2; Jk[13] = Jk[1] + bf * x / z_2; Jk[14] = Jk[2]; Jk[15] = Jk[3]; Jk[16] = 0; Jk[17] = Jk[5] - bf / z_2; }
        else { for (int q = 12; q < 18; q++) Jk[q] = 0; }
        const double e0 = D.err[3 * k], e1 = D.err[3 * k + 1], e2 = D.err[3 * k + 2], is2 = ob[3];
        double r0, r1 = 1;
        if (kernels) huber(is2 * (e0 * e0 + e1 * e1 + e2 * e2), (double)(float)sqrt(stereo ? 7.815 : 5.991), &r0, &r1);
        const double w = r1 * is2;
        D.wgt[k] = w;
        for (int a = 0; a < 9; a++) D.Jp[9 * (size_t)k + a] = Jp[a];
        for (int a = 0; a < 18; a++) D.Jk[18 * (size_t)k + a] = Jk[a];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) D.We[18 * (size_t)k + 3 * r + c] = w * (Jk[r] * Jp[c] + Jk[6 + r] * Jp[3 + c] + Jk[12 + r] * Jp[6 + c]);
        H[0] += w * (Jp[0] * Jp[0] + Jp[3] * Jp[3] + Jp[6] * Jp[6]); H[1] += w * (Jp[0] * Jp[1] + Jp[3] * Jp[4] + Jp[6] * Jp[7]); H[2] += w * (Jp[0] * Jp[2] + Jp[3] * Jp[5] + Jp[6] * Jp[8]);
        H[3] += w * (Jp[1] * Jp[1] + Jp[4] * Jp[4] + Jp[7] * Jp[7]); H[4] += w * (Jp[1] * Jp[2] + Jp[4] * Jp[5] + Jp[7] * Jp[8]); H[5] += w * (Jp[2] * Jp[2] + Jp[5] * Jp[5] + Jp[8] * Jp[8]);
        for (int a = 0; a < 3; a++) b[a] -= w * (Jp[a] * e0 + Jp[3 + a] * e1 + Jp[6 + a] * e2);
    }
    double* Ho = D.Hll + (size_t)p * 9;
    Ho[0] = H[0]; Ho[1] = H[1]; Ho[2] = H[2]; Ho[3] = H[1]; Ho[4] = H[3]; Ho[5] = H[4]; Ho[6] = H[2]; Ho[7] = H[4]; Ho[8] = H[5];
    for (int a = 0; a < 3; a++) D.bl[(size_t)p * 3 + a] = b[a];
}
__device__ __forceinline__ void k_ba_se3_update_body(const BaDev& D) {
    if (ba_skip(D)) return;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < D.W) {                                                   // VertexSE3Expmap::oplusImpl: T <- exp(update) * T
        double* k = D.kf + (size_t)q * 7;
        for (int c = 0; c < 7; c++) D.kf_bak[(size_t)q * 7 + c] = k[c];
        const se3q s = se3_mul(se3_exp(D.xp + 6 * q), ba_ld_se3(k));
        k[0] = s.r.x; k[1] = s.r.y; k[2] = s.r.z; k[3] = s.r.w; k[4] = s.t.x; k[5] = s.t.y; k[6] = s.t.z;
    }
    if (q < D.NP) for (int c = 0; c < 3; c++) { D.pt_bak[3 * q + c] = D.pt[3 * q + c]; D.pt[3 * q + c] += D.xl[3 * q + c]; }
}
// chi2 (5.991 mono / 7.815 stereo, stale error on excluded edges) / depth gate, Optimizer.cc:4170-4200 and :4207-4235
__device__ __forceinline__ void k_ba_se3_gate_body(const BaDev& D, uint8_t* out, int set_level) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= D.NE) return;
    const double* ob = D.e_obs + 4 * (size_t)k;
    const d3 pc = se3_map(ba_ld_se3(D.kf + (size_t)D.e_kf[k] * 7), ld3(D.pt + (size_t)D.e_pt[k] * 3));
    const double chi = ob[3] * (D.err[3 * k] * D.err[3 * k] + D.err[3 * k + 1] * D.err[3 * k + 1] + D.err[3 * k + 2] * D.err[3 * k + 2]);
    const int bad = (chi > (ob[2] < 0 ? 5.991 : 7.815) || !(pc.z > 0.0)) ? 1 : 0;
    if (set_level) { if (bad) D.level[k] = 1; } else out[k] = (uint8_t)bad;
}
__global__ void k_ba_se3_errors(BaDev D, int kernels) { k_ba_se3_errors_body(D, kernels); }
__global__ void k_ba_se3_lin_points(BaDev D, int kernels) { k_ba_se3_lin_points_body(D, kernels); }
__global__ void k_ba_se3_update(BaDev D) { k_ba_se3_update_body(D); }
__global__ void k_ba_se3_gate(BaDev D, uint8_t* out, int set_level) { k_ba_se3_gate_body(D, out, set_level); }

// ---- fused launches of the single window's device-side LM chunk (BaSolve::advance, ST_FAST_CHUNK): kernels with no dependency between
// them share a launch (by block range), the per-point back-substitution applies its own update, and the last block of the error pass
// takes the trial's decision — 8 launches per LM iteration instead of 13. A single window is a chain of latency-bound launches of
// 5-25 us each: every launch removed is ~7 us of the ~230 us a trial takes.
__global__ __launch_bounds__(256) void k_ba_f_lin_clear(BaDev D, int mono_kernel, int first, int gP, int gC) {
    if ((int)blockIdx.x < gP) { if (D.pose_dim == 12) k_ba_lin_points_body(D, mono_kernel); else k_ba_se3_lin_points_body(D, mono_kernel); }
    else k_ba_clear_body(D, first, (int)blockIdx.x - gP, gC);
}
__global__ __launch_bounds__(256) void k_ba_f_line_clear(BaDev D, int mono_kernel, int first, int gE, int gC) {      // NavState window: one thread per edge
    if ((int)blockIdx.x < gE) k_ba_lin_edges_body(D, mono_kernel); else k_ba_clear_body(D, first, (int)blockIdx.x - gE, gC);
}
__global__ __launch_bounds__(256) void k_ba_f_hpp_imu_hll(BaDev D) {
    const int b = blockIdx.x;
    if (b < D.W) k_ba_hpp_body(D, b); else if (b < 2 * D.W) k_ba_imu_body(D, b - D.W); else k_ba_hll_body(D, b - 2 * D.W);
}
__global__ __launch_bounds__(256) void k_ba_f_hpp_imu(BaDev D) {
    if ((int)blockIdx.x < D.W) k_ba_hpp_body(D, blockIdx.x); else k_ba_imu_body(D, (int)blockIdx.x - D.W);
}
__global__ __launch_bounds__(256) void k_ba_f_init_dinv(BaDev D, int gR) {
    if ((int)blockIdx.x < gR) k_ba_init_reduced_body(D, 0.0); else k_ba_dinv_body(D, 0.0, (int)blockIdx.x - gR);
}
// back-substitution of the point block with EIGHT lanes per point (one observation each per trip, shuffle reduction) + the update: the
// thread-per-point form (k_ba_backsub_body + k_ba_*_update_body) is 8 workgroups for a 2000-point window and the latency of a lane
// walking its point's observations one after the other
__device__ __forceinline__ void k_ba_backsub8_update_body(const BaDev& D) {
    if (ba_skip(D)) return;
    const double lambda = ba_lambda(D, 0.0);
    __shared__ double s_red[4];
    const int g = blockIdx.x * blockDim.x + threadIdx.x, p = g >> 3, sub = g & 7;
    double a0 = 0, a1 = 0, a2 = 0;
    if (p < D.NP)
        for (int k = D.pt_start[p] + sub; k < D.pt_start[p + 1]; k += 8) {
            if (D.level[k] != 0 || D.e_kf[k] >= D.W) continue;
            const double* Wk = D.We + (size_t)18 * k;
            const int ba = D.pose_dim * D.e_kf[k];
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const double x = D.xp[ba + ba_loc(D, r)];
                a0 += Wk[3 * r] * x; a1 += Wk[3 * r + 1] * x; a2 += Wk[3 * r + 2] * x;
            }
        }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) { a0 += __shfl_xor(a0, d); a1 += __shfl_xor(a1, d); a2 += __shfl_xor(a2, d); }
    double sc = 0;
    if (p < D.NP && sub == 0) {
        const double b0 = D.bl[3 * p], b1 = D.bl[3 * p + 1], b2 = D.bl[3 * p + 2];
        const double c0 = b0 - a0, c1 = b1 - a1, c2 = b2 - a2;
        const double* Di = D.Dinv + (size_t)p * 9;
        const double x0 = Di[0] * c0 + Di[1] * c1 + Di[2] * c2, x1 = Di[3] * c0 + Di[4] * c1 + Di[5] * c2, x2 = Di[6] * c0 + Di[7] * c1 + Di[8] * c2;
        D.xl[3 * p] = x0; D.xl[3 * p + 1] = x1; D.xl[3 * p + 2] = x2;
        sc = x0 * (lambda * x0 + b0) + x1 * (lambda * x1 + b1) + x2 * (lambda * x2 + b2);
        const double xs[3] = {x0, x1, x2};
        for (int c = 0; c < 3; c++) { D.pt_bak[3 * p + c] = D.pt[3 * p + c]; D.pt[3 * p + c] += xs[c]; }      // k_ba_*_update_body's point part
    }
    if (blockIdx.x == 0) for (int q = threadIdx.x; q < D.np; q += blockDim.x) sc += D.xp[q] * (lambda * D.xp[q] + D.bp[q]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sc += __shfl_xor(sc, d);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&D.scal[1], s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    if (g < D.W) {                                                       // the key frames (k_ba_update_body / k_ba_se3_update_body)
        if (D.pose_dim == 12) {
            double* k = D.kf + (size_t)g * 22;
            for (int c = 0; c < 22; c++) D.kf_bak[(size_t)g * 22 + c] = k[c];
            const pvr s = inc_small_pvr(ld_pvr(k), D.xp + 12 * g);
            st_pvr(k, s);
            for (int c = 0; c < 3; c++) k[19 + c] += D.xp[12 * g + 9 + c];
        } else {
            double* k = D.kf + (size_t)g * 7;
            for (int c = 0; c < 7; c++) D.kf_bak[(size_t)g * 7 + c] = k[c];
            const se3q s = se3_mul(se3_exp(D.xp + 6 * g), ba_ld_se3(k));
            k[0] = s.r.x; k[1] = s.r.y; k[2] = s.r.z; k[3] = s.r.w; k[4] = s.t.x; k[5] = s.t.y; k[6] = s.t.z;
        }
    }
}
__global__ __launch_bounds__(256) void k_ba_f_backsub8_update(BaDev D) { k_ba_backsub8_update_body(D); }
// ---- the lock-step batch's fused launches (blockIdx.y = window; the same bodies, gated by the window's control block) ----
__global__ __launch_bounds__(256) void k_bab_f_lin_clear(const BaDev* __restrict__ Dv, int gL, int gC) {   // gL = blocks of the linearisation: edges (NavState) or points (SE3)
    BA_B_WINDOW(); if (c[BA_B_NEED_LIN] == 0.0) return;
    if ((int)blockIdx.x < gL) { if (D.pose_dim == 12) k_ba_lin_edges_body(D, c[BA_B_MONO] != 0.0); else k_ba_se3_lin_points_body(D, c[BA_B_MONO] != 0.0); }
    else k_ba_clear_body(D, 0, (int)blockIdx.x - gL, gC);
}
__global__ __launch_bounds__(256) void k_bab_f_hpp_imu_hll(const BaDev* __restrict__ Dv, int Wmax) {
    BA_B_WINDOW(); if (c[BA_B_NEED_LIN] == 0.0) return;
    const int b = blockIdx.x;
    if (b < Wmax) { if (b < D.W) k_ba_hpp_body(D, b); }
    else if (b < 2 * Wmax) { if (b - Wmax < D.W && D.pose_dim == 12) k_ba_imu_body(D, b - Wmax); }
    else if (D.pose_dim == 12) k_ba_hll_body(D, b - 2 * Wmax);           // the SE3 linearisation (one thread per point) wrote its point blocks itself
}
__global__ __launch_bounds__(256) void k_bab_f_init_dinv(const BaDev* __restrict__ Dv, int gR) {
    BA_B_WINDOW();
    if ((int)blockIdx.x < gR) k_ba_init_reduced_body(D, 0.0); else k_ba_dinv_body(D, 0.0, (int)blockIdx.x - gR);
}
__global__ __launch_bounds__(256) void k_bab_f_backsub8_update(const BaDev* __restrict__ Dv) { BA_B_WINDOW(); k_ba_backsub8_update_body(D); }
__global__ __launch_bounds__(256) void k_ba_f_errors_decide(BaDev D, int mono_kernel, int last_of_phase) {
    if (D.pose_dim == 12) k_ba_errors_body(D, mono_kernel); else k_ba_se3_errors_body(D, mono_kernel);
    __shared__ int s_last;
    if (threadIdx.x == 0) { __threadfence(); s_last = atomicAdd(D.ticket, 1) == (int)gridDim.x - 1; }
    __syncthreads();
    if (s_last && threadIdx.x == 0) { *D.ticket = 0; __threadfence(); k_ba_decide_body(D, last_of_phase); }
}
// lock-step batch, the launches whose body depends on the kind of window (pose_dim 12: NavState, 6: vision-only SE3)
__global__ void k_bab_errors_chi(const BaDev* __restrict__ Dv) {
    BA_B_WINDOW(); if (c[BA_B_NEED_CHI] == 0.0) return;
    if (D.pose_dim == 12) k_ba_errors_body(D, c[BA_B_MONO] != 0.0); else k_ba_se3_errors_body(D, c[BA_B_MONO] != 0.0);
}
__global__ void k_bab_errors(const BaDev* __restrict__ Dv) {
    BA_B_WINDOW();
    if (D.pose_dim == 12) k_ba_errors_body(D, c[BA_B_MONO] != 0.0); else k_ba_se3_errors_body(D, c[BA_B_MONO] != 0.0);
}
// control blocks of a group's windows at the start of the solve (one launch instead of one small copy per window)
__global__ void k_bab_ctl_init(const BaDev* __restrict__ Dv) {
    double* c = Dv[blockIdx.x].ctl;
    const int t = threadIdx.x;
    if (t < BA_B_N) c[t] = (t == BA_B_NEED_CHI || t == BA_B_NEED_LIN || t == BA_B_FIRST || t == BA_B_MONO) ? 1.0 : (t == BA_B_ITERS ? 5.0 : 0.0);
}
// results of a group's windows into one device buffer (one copy back for the group instead of four per window): per window, at offs[4 w ..],
// key frames | points | erase flags | control block
__global__ void k_bab_pack(const BaDev* __restrict__ Dv, uint8_t* const* __restrict__ erase, uint8_t* __restrict__ out, const unsigned long long* __restrict__ offs) {
    const BaDev& D = Dv[blockIdx.y];
    const unsigned long long* o = offs + 4 * (size_t)blockIdx.y;
    const size_t nkf = (size_t)D.W * D.kf_stride, npt = (size_t)D.NP * 3, ner = (size_t)D.NE;
    double* okf = reinterpret_cast<double*>(out + o[0]); double* opt = reinterpret_cast<double*>(out + o[1]); uint8_t* oer = out + o[2]; double* oct = reinterpret_cast<double*>(out + o[3]);
    size_t n = ner > npt ? ner : npt; n = n > nkf ? n : nkf; n = n > (size_t)BA_B_N ? n : (size_t)BA_B_N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (i < ner) oer[i] = erase[blockIdx.y][i];
        if (i < nkf) okf[i] = D.kf[i];
        if (i < npt) opt[i] = D.pt[i];
        if (i < BA_B_N) oct[i] = D.ctl[i];
    }
}
__global__ void k_bab_gate(const BaDev* __restrict__ Dv, uint8_t* const* __restrict__ erase) {
    const BaDev& D = Dv[blockIdx.y]; const double* c = D.ctl;
    if (c[BA_B_DONE] != 0.0 || c[BA_B_GATE] == 0.0) return;
    if (D.pose_dim == 12) k_ba_gate_body(D, erase[blockIdx.y], c[BA_B_GATE] == 1.0 ? 1 : 0);
    else k_ba_se3_gate_body(D, erase[blockIdx.y], c[BA_B_GATE] == 1.0 ? 1 : 0);
}

} // namespace viorb

using namespace viorb;

namespace {
// A solve borrows a context (a HIP stream + a device arena) from a small pool, so that concurrent callers (the LocalMapping threads
// of several SLAM instances) run on different streams and no call pays hipMalloc / hipFree, which synchronise the whole device.
struct BaCtx {
    hipStream_t st = nullptr; void* arena = nullptr; size_t bytes = 0;
    double* pinned = nullptr;           // 64 doubles of page-locked host memory for the LM scalars
    uint8_t* stage = nullptr; size_t stage_bytes = 0;       // page-locked staging copy of a window's inputs (lives until the context's next window)
    int device = 0;
};
std::mutex g_ctx_mu;
std::vector<BaCtx*> g_ctx_free;
std::atomic<int> g_lba_device{-1};
// The device the window solves run on: viorb_local_ba_set_device(), else the calling thread's current HIP device (what
// torch.cuda.set_device(LOCAL_RANK) / hipSetDevice selected), so that under torchrun every rank's windows land on its own GPU.
static int lba_device() {
    int d = g_lba_device.load();
    if (d < 0 && hipGetDevice(&d) != hipSuccess) d = 0;
    return d;
}
struct BaCtxLease {
    BaCtx* c = nullptr;
    BaCtxLease() {}
    ~BaCtxLease() { if (c) { std::lock_guard<std::mutex> lk(g_ctx_mu); g_ctx_free.push_back(c); } }
    bool ready() {                     // the calling thread has selected the device (hipSetDevice(lba_device()))
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (!c) {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            for (size_t i = 0; i < g_ctx_free.size(); i++)
                if (g_ctx_free[i]->device == dev) { c = g_ctx_free[i]; g_ctx_free.erase(g_ctx_free.begin() + i); break; }
        }
        if (!c) {
            c = new BaCtx();
            c->device = dev;
            if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess || hipHostMalloc(reinterpret_cast<void**>(&c->pinned), 64 * sizeof(double)) != hipSuccess) {
                delete c; c = nullptr; return false;
            }
        }
        return true;
    }
};
// alloc() lays the arrays out in a host mirror (inputs copied, work arrays zero); commit() uploads the mirror into the context's
// arena in one copy and patches the recorded pointers.
struct BaBuf {
    // Inputs are laid out in a host mirror and uploaded in one copy; work arrays follow them in the arena and are only zeroed on the
    // device (uploading their zeros from the host was most of the per-window host time of a batch).
    struct Item { void** slot; size_t off; int kind; };            // 0 input, 1 zeroed work array, 2 work array that is written before it is read
    std::vector<Item> items;
    std::vector<uint8_t> mirror;
    size_t work_bytes = 0, raw_bytes = 0;
    template <class T> bool alloc(T** d, size_t n, const T* src = nullptr) {
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        if (src) {
            const size_t off = (mirror.size() + 255) & ~(size_t)255;
            mirror.resize(off + bytes, 0);
            if (n) memcpy(mirror.data() + off, src, n * sizeof(T));
            items.push_back({reinterpret_cast<void**>(d), off, 0});
        } else {
            const size_t off = (work_bytes + 255) & ~(size_t)255;
            work_bytes = off + bytes;
            items.push_back({reinterpret_cast<void**>(d), off, 1});
        }
        *d = nullptr;
        return true;
    }
    // the large per-edge / per-point work arrays: every element a kernel reads has been written by an earlier kernel of the same solve
    // (zeroing them was 4 MB of memset per window, 8 % of a batch's stream time)
    template <class T> bool alloc_raw(T** d, size_t n) {
        const size_t off = (raw_bytes + 255) & ~(size_t)255;
        raw_bytes = off + std::max<size_t>(n, 1) * sizeof(T);
        items.push_back({reinterpret_cast<void**>(d), off, 2});
        *d = nullptr;
        return true;
    }
    bool commit(BaCtx* c) {
        const size_t in_bytes = (mirror.size() + 255) & ~(size_t)255, zero_bytes = (work_bytes + 255) & ~(size_t)255, total = in_bytes + zero_bytes + raw_bytes;
        if (c->bytes < total) {
            if (c->arena) (void)hipFree(c->arena);
            c->arena = nullptr; c->bytes = 0;
            const size_t want = total + total / 4;
            if (hipMalloc(&c->arena, want) != hipSuccess) { c->arena = nullptr; return false; }
            c->bytes = want;
        }
        uint8_t* base = static_cast<uint8_t*>(c->arena);
        // The inputs go through the context's page-locked staging buffer: a truly asynchronous copy, and no synchronisation per window (a
        // pageable source made every upload a blocking staged copy + a stream synchronise: half of a batch's wall time was spent here).
        // The staging buffer belongs to the context, and a context serves one window at a time.
        if (!mirror.empty()) {
            if (c->stage_bytes < mirror.size()) {
                if (c->stage) (void)hipHostFree(c->stage);
                c->stage = nullptr; c->stage_bytes = 0;
                const size_t want = mirror.size() + mirror.size() / 4;
                if (hipHostMalloc(reinterpret_cast<void**>(&c->stage), want) != hipSuccess) { c->stage = nullptr; return false; }
                c->stage_bytes = want;
            }
            memcpy(c->stage, mirror.data(), mirror.size());
            if (hipMemcpyAsync(base, c->stage, mirror.size(), hipMemcpyHostToDevice, c->st) != hipSuccess) return false;
        }
        if (work_bytes && hipMemsetAsync(base + in_bytes, 0, work_bytes, c->st) != hipSuccess) return false;
        for (const Item& it : items) *it.slot = base + (it.kind == 0 ? it.off : (it.kind == 1 ? in_bytes + it.off : in_bytes + zero_bytes + it.off));
        return true;
    }
};
bool host_inverse9(const double* a_in, double* inv) {
    double a[81]; for (int i = 0; i < 81; i++) { a[i] = a_in[i]; inv[i] = (i / 9 == i % 9) ? 1.0 : 0.0; }
    for (int col = 0; col < 9; col++) {
        int p = col; double best = std::fabs(a[col * 9 + col]);
        for (int i = col + 1; i < 9; i++) if (std::fabs(a[i * 9 + col]) > best) { best = std::fabs(a[i * 9 + col]); p = i; }
        if (best == 0) return false;
        if (p != col) for (int j = 0; j < 9; j++) { std::swap(a[p * 9 + j], a[col * 9 + j]); std::swap(inv[p * 9 + j], inv[col * 9 + j]); }
        const double iv = 1.0 / a[col * 9 + col];
        for (int j = 0; j < 9; j++) { a[col * 9 + j] *= iv; inv[col * 9 + j] *= iv; }
        for (int i = 0; i < 9; i++) if (i != col) { const double f = a[i * 9 + col]; if (f == 0) continue; for (int j = 0; j < 9; j++) { a[i * 9 + j] -= f * a[col * 9 + j]; inv[i * 9 + j] -= f * inv[col * 9 + j]; } }
    }
    return true;
}
} // namespace

// The Levenberg-Marquardt control flow of g2o (optimization_algorithm_levenberg.cpp:61-189) shared by both window solves:
// optimize(5) -> gate / drop kernels -> optimize(10) -> erase flags. model 0: NavState window, 1: SE3 (vision-only) window.
// The LM driver reads three scalars per trial: it polls the stream instead of blocking in hipStreamSynchronize (whose wake-up costs more
// than the kernels of a trial take).
struct BaSolve;
static int ba_launch_pairs(const BaDev& D, hipStream_t st);      // the Schur complement's gather lists (k_ba_pairs_build, below)
static void ba_mirror_stop_flags(BaSolve* const* S, int n);
// Waits for the stream; meanwhile mirrors the callers' stop flags into the page-locked words the device-side LM control polls.
static hipError_t ba_wait(hipStream_t st, BaSolve* const* S = nullptr, int n = 0) {
    for (;;) {
        if (n) ba_mirror_stop_flags(S, n);
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
    }
}
// The LM driver (g2o's optimize(5) -> gate -> optimize(10), reference src/Optimizer.cc:2100-2160 and the vision-only twin) as a
// resumable state machine: advance() runs the host logic up to the next point where device scalars are needed, enqueues the work on
// the solve's stream and returns BA_WAIT; the caller resumes it once the stream has drained. One window blocks on its stream between
// calls; a batch keeps several windows in flight from one host thread (several host threads slow each other down inside the HIP runtime).
enum { BA_WAIT = 0, BA_DONE = 1, BA_PIN_ABORT = 56 };              // pinned[56]: the mirrored stop flag (a 64-bit word; 0..39 LM scalars, 48 the batch's done counter)
struct BaSolve {
    BaCtxLease lease;
    BaDev D; hipStream_t st = nullptr; double* h = nullptr;       // h: 64 page-locked doubles (0..7 scal, 8..39 ctl)
    int model = 0; const volatile int* stop = nullptr; uint8_t* d_erase = nullptr;
    double* kfs_out = nullptr; double* points_out = nullptr; uint8_t* erase = nullptr; double* info = nullptr;
    // LM state
    enum State { ST_OPT_BEGIN, ST_ITER_BEGIN, ST_AFTER_CHI, ST_AFTER_DIAG, ST_TRIAL_ENQ, ST_AFTER_TRIAL, ST_FINISH, ST_AFTER_FINAL, ST_DONE,
                 ST_FAST_CHUNK, ST_FAST_WAIT } state = ST_OPT_BEGIN;
    // Device-side LM control (k_ba_decide_body): iterations are enqueued FAST_CHUNK at a time with no host round trip in between (8 launches
    // each, k_ba_f_*); the host looks at the control block once per chunk and only takes the per-trial path below after a trial was
    // rejected. The caller's stop flag, which g2o polls once per iteration, is polled on the device through its page-locked mirror.
    // VIORB_LBA_HOST_LM=1 forces the per-trial path (tests compare both).
    enum { FAST_CHUNK = 10 };          // a whole optimize() call per round trip (5 and 10 iterations); the stop flag is polled on the device (k_ba_decide_body)
    bool fast_phase = false; int fast_enq = 0;
    int phase = 0, iterations = 5, it = 0, nBad = 0, qmax = 0, mono_kernel = 1;
    int its[2] = {0, 0}; double chi[2] = {0, 0};
    double lambda = 0, ni = 2, currentChi = 0, iniChi = 0, rho = 0;
    bool terminate() const { return stop && *stop; }
    int advance() {
        const int nk = D.NK, npts = D.NP, ne = D.NE, n_local = D.W;
        const size_t n2 = (size_t)D.np * D.np, nl2 = (size_t)D.ld * D.ld;
        const int TB = 256, gE = (ne + TB - 1) / TB, gP = (npts + TB - 1) / TB;
        auto enqueue_errors = [&]() -> int {
            VIORB_HIP_TRY(hipMemsetAsync(D.scal, 0, sizeof(double), st));
            if (model == 0) hipLaunchKernelGGL(k_ba_errors, dim3(gE), dim3(TB), 0, st, D, mono_kernel);
            else hipLaunchKernelGGL(k_ba_se3_errors, dim3(gE), dim3(TB), 0, st, D, mono_kernel);
            return VIORB_OK;
        };
        int rc;
        for (;;) {
            switch (state) {
            case ST_OPT_BEGIN: {
                it = 0; currentChi = 0; nBad = 0;
                static const bool host_lm = getenv("VIORB_LBA_HOST_LM") != nullptr;
                fast_phase = !host_lm;
                D.use_ctl = 0;
                if (!fast_phase) { state = ST_ITER_BEGIN; break; }
                // chi2 at the phase's starting point (computeActiveErrors of the first iteration) and a clean control block
                VIORB_HIP_TRY(hipMemsetAsync(D.ctl, 0, BA_CTL_N * sizeof(double), st));
                if ((rc = enqueue_errors()) != VIORB_OK) return rc;
                fast_enq = 0; state = ST_FAST_CHUNK;
                break;
            }
            case ST_FAST_CHUNK: {
                if (terminate() || fast_enq >= iterations) { state = ST_FINISH; break; }
                D.use_ctl = 1;
                const int n = std::min((int)FAST_CHUNK, iterations - fast_enq);
                const unsigned gC = (unsigned)std::min<size_t>((n2 + TB - 1) / TB, 256);
                for (int i = 0; i < n; i++, fast_enq++) {
                    const int first = fast_enq == 0;
                    const unsigned gR = (unsigned)((nl2 + TB - 1) / TB);
                    if (model == 0) {
                        hipLaunchKernelGGL(k_ba_f_line_clear, dim3(gE + gC), dim3(TB), 0, st, D, mono_kernel, first, (int)gE, (int)gC);
                        hipLaunchKernelGGL(k_ba_f_hpp_imu_hll, dim3(2 * n_local + gP), dim3(256), 0, st, D);
                    } else {
                        hipLaunchKernelGGL(k_ba_f_lin_clear, dim3(gP + gC), dim3(TB), 0, st, D, mono_kernel, first, (int)gP, (int)gC);
                        hipLaunchKernelGGL(k_ba_f_hpp_imu, dim3(n_local), dim3(256), 0, st, D);
                    }
                    if (first) {
                        hipLaunchKernelGGL(k_ba_max_diag, dim3(32), dim3(256), 0, st, D);
                        hipLaunchKernelGGL(k_ba_lambda0, dim3(1), dim3(64), 0, st, D);
                    }
                    hipLaunchKernelGGL(k_ba_f_init_dinv, dim3(gR + gP), dim3(TB), 0, st, D, (int)gR);
                    hipLaunchKernelGGL(k_ba_schur, dim3(n_local * (n_local + 1) / 2), dim3(64), 0, st, D);
                    (void)raise_dynamic_lds(reinterpret_cast<const void*>(k_ba_chol_solve), BA_CHOL_LDS_BYTES);
                    hipLaunchKernelGGL(k_ba_chol_solve, dim3(1), dim3(BA_CHOL_THREADS), BA_CHOL_LDS_BYTES, st, D);
                    hipLaunchKernelGGL(k_ba_f_backsub8_update, dim3((unsigned)std::max((8 * npts + TB - 1) / TB, 1)), dim3(TB), 0, st, D);
                    hipLaunchKernelGGL(k_ba_f_errors_decide, dim3(gE), dim3(TB), 0, st, D, mono_kernel, 0);
                }
                D.use_ctl = 0;
                VIORB_HIP_TRY(hipMemcpyAsync(h, D.scal, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
                VIORB_HIP_TRY(hipMemcpyAsync(h + 8, D.ctl, BA_CTL_N * sizeof(double), hipMemcpyDeviceToHost, st));
                state = ST_FAST_WAIT;
                return BA_WAIT;
            }
            case ST_FAST_WAIT: {
                const double* c = h + 8;
                its[phase] = (int)c[BA_CTL_ITS]; chi[phase] = c[BA_CTL_CHI]; it = (int)c[BA_CTL_IT];
                currentChi = c[BA_CTL_CHI]; iniChi = c[BA_CTL_INICHI]; lambda = c[BA_CTL_LAMBDA]; ni = c[BA_CTL_NI]; nBad = (int)c[BA_CTL_NBAD];
                const int halt = (int)c[BA_CTL_HALT];
                if (halt == 1) { qmax = 0; rho = 0; fast_phase = false; state = ST_AFTER_TRIAL; break; }      // a rejected trial: h[0..2] hold its scalars
                if (halt == 2 || it >= iterations) { state = ST_FINISH; break; }
                state = ST_FAST_CHUNK;
                break;
            }
            case ST_ITER_BEGIN:
                if (it >= iterations || terminate()) { state = ST_FINISH; break; }
                if ((rc = enqueue_errors()) != VIORB_OK) return rc;
                VIORB_HIP_TRY(hipMemcpyAsync(h + 8, D.scal, sizeof(double), hipMemcpyDeviceToHost, st));
                state = ST_AFTER_CHI;
                return BA_WAIT;
            case ST_AFTER_CHI:
                currentChi = h[8]; iniChi = currentChi;
                VIORB_HIP_TRY(hipMemsetAsync(D.Hpp, 0, n2 * sizeof(double), st));
                VIORB_HIP_TRY(hipMemsetAsync(D.bp, 0, D.np * sizeof(double), st));
                if (model == 0) hipLaunchKernelGGL(k_ba_lin_points, dim3(gP), dim3(TB), 0, st, D, mono_kernel);
                else hipLaunchKernelGGL(k_ba_se3_lin_points, dim3(gP), dim3(TB), 0, st, D, mono_kernel);
                hipLaunchKernelGGL(k_ba_hpp, dim3(n_local), dim3(256), 0, st, D);
                if (model == 0) hipLaunchKernelGGL(k_ba_imu, dim3(n_local), dim3(256), 0, st, D);
                rho = 0; qmax = 0;
                if (it == 0) {
                    VIORB_HIP_TRY(hipMemsetAsync(D.scal + 3, 0, sizeof(double), st));
                    hipLaunchKernelGGL(k_ba_max_diag, dim3(32), dim3(256), 0, st, D);
                    VIORB_HIP_TRY(hipMemcpyAsync(h, D.scal, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
                    state = ST_AFTER_DIAG;
                    return BA_WAIT;
                }
                state = ST_TRIAL_ENQ;
                break;
            case ST_AFTER_DIAG:
                lambda = 1e-5 * h[3]; ni = 2; nBad = 0; state = ST_TRIAL_ENQ;
                break;
            case ST_TRIAL_ENQ:
                hipLaunchKernelGGL(k_ba_init_reduced, dim3((unsigned)((nl2 + TB - 1) / TB)), dim3(TB), 0, st, D, lambda);
                hipLaunchKernelGGL(k_ba_dinv, dim3(gP), dim3(TB), 0, st, D, lambda);
                hipLaunchKernelGGL(k_ba_schur, dim3(n_local * (n_local + 1) / 2), dim3(64), 0, st, D);
                (void)raise_dynamic_lds(reinterpret_cast<const void*>(k_ba_chol_solve), BA_CHOL_LDS_BYTES);
                    hipLaunchKernelGGL(k_ba_chol_solve, dim3(1), dim3(BA_CHOL_THREADS), BA_CHOL_LDS_BYTES, st, D);
                hipLaunchKernelGGL(k_ba_backsub, dim3(gP), dim3(TB), 0, st, D, lambda);
                if (model == 0) hipLaunchKernelGGL(k_ba_update, dim3(std::max(gP, 1)), dim3(TB), 0, st, D);
                else hipLaunchKernelGGL(k_ba_se3_update, dim3(std::max(gP, 1)), dim3(TB), 0, st, D);
                // chi2 of the trial state, the solver's flag and the gain denominator come back in one copy / one synchronisation
                // (the factorisation kernel has zeroed both accumulators; the state backup is part of the update kernel)
                if (model == 0) hipLaunchKernelGGL(k_ba_errors, dim3(gE), dim3(TB), 0, st, D, mono_kernel);
                else hipLaunchKernelGGL(k_ba_se3_errors, dim3(gE), dim3(TB), 0, st, D, mono_kernel);
                VIORB_HIP_TRY(hipMemcpyAsync(h, D.scal, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
                state = ST_AFTER_TRIAL;
                return BA_WAIT;
            case ST_AFTER_TRIAL: {
                double tempChi = h[0];
                const bool ok2 = h[2] > 0.5;
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                const double scale = (ok2 ? h[1] : 0.0) + 1e-3;
                rho = (currentChi - tempChi) / scale;
                if (rho > 0 && std::isfinite(tempChi)) { double alpha = 1. - std::pow((2 * rho - 1), 3); alpha = std::min(alpha, 2. / 3.); lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi; }
                else {
                    lambda *= ni; ni *= 2;
                    hipLaunchKernelGGL(k_ba_restore, dim3(std::max(gP, 1)), dim3(TB), 0, st, D);
                }
                qmax++;
                if (rho < 0 && qmax < 10 && !terminate()) { state = ST_TRIAL_ENQ; break; }
                its[phase]++; chi[phase] = currentChi;
                bool stop_opt = (qmax == 10 || rho == 0);
                if (!stop_opt) { if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0; if (nBad >= 3) stop_opt = true; }
                it++;
                if (stop_opt) it = iterations;                   // leaves the iteration loop through ST_ITER_BEGIN
                state = ST_ITER_BEGIN;
                break;
            }
            case ST_FINISH:
                if (phase == 0 && !terminate()) {
                    if (model == 0) hipLaunchKernelGGL(k_ba_gate, dim3(gE), dim3(TB), 0, st, D, d_erase, 1);
                    else hipLaunchKernelGGL(k_ba_se3_gate, dim3(gE), dim3(TB), 0, st, D, d_erase, 1);
                    mono_kernel = 0; phase = 1; iterations = 10; state = ST_OPT_BEGIN;
                    break;
                }
                if (model == 0) hipLaunchKernelGGL(k_ba_gate, dim3(gE), dim3(TB), 0, st, D, d_erase, 0);
                else hipLaunchKernelGGL(k_ba_se3_gate, dim3(gE), dim3(TB), 0, st, D, d_erase, 0);
                VIORB_HIP_TRY(hipMemcpyAsync(kfs_out, D.kf, (size_t)n_local * D.kf_stride * sizeof(double), hipMemcpyDeviceToHost, st));
                VIORB_HIP_TRY(hipMemcpyAsync(points_out, D.pt, (size_t)npts * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
                VIORB_HIP_TRY(hipMemcpyAsync(erase, d_erase, ne, hipMemcpyDeviceToHost, st));
                state = ST_AFTER_FINAL;
                return BA_WAIT;
            case ST_AFTER_FINAL:
                info[0] = chi[0]; info[1] = chi[1]; info[2] = its[0]; info[3] = its[1];
                state = ST_DONE;
                return BA_DONE;
            case ST_DONE:
                return BA_DONE;
            }
        }
    }
    // one window: block on the stream between the steps
    int run() {
        if (state != ST_DONE) { const int rc = ba_launch_pairs(D, st); if (rc != VIORB_OK) return rc; }
        for (;;) {
            const int r = advance();
            if (r < 0 || r == BA_DONE) return r < 0 ? r : VIORB_OK;
            BaSolve* self = this;
            VIORB_HIP_TRY(ba_wait(st, &self, 1));
        }
    }
};

static void ba_mirror_stop_flags(BaSolve* const* S, int n) {
    for (int i = 0; i < n; i++)
        if (S[i] && S[i]->h && S[i]->D.abort_host && S[i]->terminate())
            reinterpret_cast<volatile unsigned long long*>(S[i]->h)[BA_PIN_ABORT] = 1ull;
}
// the page-locked mirror of the caller's stop flag, mapped for the device (called by ba_prepare_* once S.h and S.stop are set)
static hipError_t ba_map_stop_flag(BaSolve& S) {
    S.D.abort_host = nullptr;
    if (!S.stop) return hipSuccess;
    reinterpret_cast<volatile unsigned long long*>(S.h)[BA_PIN_ABORT] = 0ull;
    void* dp = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&dp, S.h + BA_PIN_ABORT, 0);
    if (e == hipSuccess) S.D.abort_host = reinterpret_cast<const unsigned long long*>(dp);
    return e;
}


// The gather lists of k_ba_schur, built on the device when a window is prepared (the graph is fixed over the solve): (edge of a, edge of b)
// for every pair of local key frames a >= b and every point both observe; the diagonal pairs list every local observation once. One
// 1024-thread workgroup per window on the window's stream: histogram of the pairs in LDS, exclusive scan, fill through LDS cursors (the
// order inside a pair's list is not fixed: it only permutes the terms of that block's sum). Building the lists on the host cost 0.1-0.2 ms
// per window, counting with global atomics 0.1 ms (30 k increments of 210 counters) — more than a batched solve gains from them.
#define BA_MAX_PAIRS 820                                               // 40 local key frames (the SE3 window's limit)
template <class F> __device__ __forceinline__ void ba_pairs_walk(const BaDev& D, F f) {
    for (int p = threadIdx.x; p < D.NP; p += blockDim.x) {
        const int k0 = D.pt_start[p], k1 = D.pt_start[p + 1];
        for (int i = k0; i < k1; i++) {
            const int a = D.e_kf[i];
            if (a >= D.W) continue;
            for (int j = k0; j < k1; j++) {
                const int b = D.e_kf[j];
                if (b <= a) f(a * (a + 1) / 2 + b, i, j);            // (i, j) with kf(i) >= kf(j); i == j once
            }
        }
    }
}
__device__ __forceinline__ void k_ba_pairs_build_body(const BaDev& D) {
    __shared__ int s_h[BA_MAX_PAIRS + 1];
    const int npairs = D.W * (D.W + 1) / 2, t = threadIdx.x;
    for (int q = t; q <= npairs; q += blockDim.x) s_h[q] = 0;
    __syncthreads();
    ba_pairs_walk(D, [&](int q, int, int) { atomicAdd(&s_h[q], 1); });
    __syncthreads();
    if (t < 64) {                                                      // exclusive scan; the counters become the fill cursors
        int run = 0;
        for (int base = 0; base <= npairs; base += 64) {
            const int q = base + t, c = q < npairs ? s_h[q] : 0;
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (t >= d) incl += o; }
            if (q <= npairs) { D.pr_start[q] = run + incl - c; s_h[q] = run + incl - c; }
            run += __shfl(incl, 63);
        }
    }
    __syncthreads();
    ba_pairs_walk(D, [&](int q, int i, int j) { const int at = atomicAdd(&s_h[q], 1); D.pr_ent[2 * (size_t)at] = i; D.pr_ent[2 * (size_t)at + 1] = j; });
}
__global__ __launch_bounds__(1024) void k_ba_pairs_build(BaDev D) { k_ba_pairs_build_body(D); }
__global__ __launch_bounds__(1024) void k_bab_pairs_build(const BaDev* __restrict__ Dv) { k_ba_pairs_build_body(Dv[blockIdx.x]); }   // a lock-step group's windows in one launch
static int ba_launch_pairs(const BaDev& D, hipStream_t st) {
    hipLaunchKernelGGL(k_ba_pairs_build, dim3(1), dim3(1024), 0, st, D);
    VIORB_HIP_TRY(hipGetLastError());
    return VIORB_OK;
}
// upper bound of the pair lists' length: every ordered pair of a point's local observations
static size_t ba_pairs_bound(const std::vector<int>& e_kf, const std::vector<int>& pt_start, int npts, int n_local) {
    size_t tot = 0;
    for (int p = 0; p < npts; p++) { size_t k = 0; for (int i = pt_start[p]; i < pt_start[p + 1]; i++) k += e_kf[i] < n_local; tot += k * k; }
    return tot;
}

// Argument checks, host-side graph bookkeeping and the upload of one NavState window; leaves `S` ready for advance() (or already done
// when the stop flag was set on entry).
static int ba_prepare_navstate(BaSolve& S, const double* kfs, int nk, int n_local, int prev_kf, const double* preint, const double* points, int npts,
                               const int32_t* edge_idx, const double* edge_obs, int ne, const double gw[3], const double cam[16],
                               const volatile int* stop, double* kfs_out, double* points_out, uint8_t* erase, double info[6]) {
    VIORB_REQUIRE(kfs && preint && points && edge_idx && edge_obs && gw && cam && kfs_out && points_out && erase && info, "null array");
    VIORB_REQUIRE(n_local >= 1 && n_local <= 20 && nk >= n_local && npts >= 1 && ne >= 1, "1 <= n_local <= 20 key frames, at least one point and edge");
    VIORB_REQUIRE(prev_kf == -1 || (prev_kf >= n_local && prev_kf < nk), "prev_kf must index a fixed key frame or be -1");
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    for (int i = 0; i < 6; i++) info[i] = 0;
    for (int i = 0; i < n_local * 22; i++) kfs_out[i] = kfs[i];
    for (int i = 0; i < npts * 3; i++) points_out[i] = points[i];
    for (int k = 0; k < ne; k++) erase[k] = 0;
    auto terminate = [&]() { return stop && *stop; };
    if (terminate()) { S.state = BaSolve::ST_DONE; return VIORB_OK; }
    // ---- host-side graph bookkeeping
    std::vector<int> e_pt(ne), e_kf(ne), pt_start(npts + 1, 0);
    for (int k = 0; k < ne; k++) {
        e_pt[k] = edge_idx[2 * k]; e_kf[k] = edge_idx[2 * k + 1];
        VIORB_REQUIRE(e_pt[k] >= 0 && e_pt[k] < npts && e_kf[k] >= 0 && e_kf[k] < nk, "edge index out of range");
        VIORB_REQUIRE(k == 0 || e_pt[k] >= e_pt[k - 1], "edges must be grouped by point (ascending point index)");
        pt_start[e_pt[k] + 1]++;
    }
    for (int p = 0; p < npts; p++) pt_start[p + 1] += pt_start[p];
    std::vector<int> kf_start(n_local + 1, 0), kf_list;
    for (int k = 0; k < ne; k++) if (e_kf[k] < n_local) kf_start[e_kf[k] + 1]++;
    for (int i = 0; i < n_local; i++) kf_start[i + 1] += kf_start[i];
    kf_list.resize(kf_start[n_local]);
    { std::vector<int> pos(kf_start.begin(), kf_start.end() - 1); for (int k = 0; k < ne; k++) if (e_kf[k] < n_local) kf_list[pos[e_kf[k]]++] = k; }
    const size_t pr_bound = ba_pairs_bound(e_kf, pt_start, npts, n_local);
    std::vector<double> info_pvr((size_t)n_local * 81);
    for (int i = 0; i < n_local; i++) if (!host_inverse9(preint + (size_t)i * 142 + 60, &info_pvr[(size_t)i * 81])) { set_error("singular IMU covariance"); return VIORB_ERR_INVALID_ARG; }

    VIORB_HIP_TRY(hipSetDevice(lba_device()));
    BaCtxLease& lease = S.lease;
    if (!lease.ready()) { set_error("could not create a HIP stream"); return VIORB_ERR_HIP; }
    BaBuf B; BaDev& D = S.D;
    D.W = n_local; D.NK = nk; D.NP = npts; D.NE = ne; D.np = 12 * n_local; D.prev_kf = prev_kf; D.acc_bias_rw2 = 5e-3 * 5e-3;
    D.pose_dim = 12; D.rows = 2; D.kf_stride = 22;
    for (int i = 0; i < 16; i++) D.cam[i] = cam[i];
    for (int i = 0; i < 3; i++) D.gw[i] = gw[i];
    int *d_ept, *d_ekf, *d_pts, *d_kfs, *d_kfl; double *d_obs, *d_pre, *d_info; uint8_t* d_erase;
    D.ld = (D.np + 15) & ~15;
    const size_t n2 = (size_t)D.np * D.np, nl2 = (size_t)D.ld * D.ld;
    bool ok = B.alloc(&D.kf, (size_t)nk * 22, kfs) && B.alloc_raw(&D.kf_bak, (size_t)nk * 22) && B.alloc(&D.pt, (size_t)npts * 3, points) && B.alloc_raw(&D.pt_bak, (size_t)npts * 3) &&
              B.alloc(&d_ept, ne, e_pt.data()) && B.alloc(&d_ekf, ne, e_kf.data()) && B.alloc(&d_obs, (size_t)ne * 3, edge_obs) && B.alloc(&D.level, ne) &&
              B.alloc_raw(&D.err, (size_t)ne * 2) && B.alloc_raw(&D.Jp, (size_t)ne * 6) && B.alloc_raw(&D.Jk, (size_t)ne * 12) && B.alloc_raw(&D.wgt, ne) && B.alloc_raw(&D.We, (size_t)ne * 18) &&
              B.alloc(&D.pr_start, (size_t)n_local * (n_local + 1) / 2 + 1) && B.alloc_raw(&D.pr_ent, 2 * pr_bound + 2) &&
              B.alloc(&d_pts, npts + 1, pt_start.data()) && B.alloc(&d_kfs, n_local + 1, kf_start.data()) && B.alloc(&d_kfl, kf_list.size(), kf_list.data()) &&
              B.alloc_raw(&D.Hll, (size_t)npts * 9) && B.alloc_raw(&D.bl, (size_t)npts * 3) && B.alloc_raw(&D.Dinv, (size_t)npts * 9) &&
              B.alloc_raw(&D.Hpp, n2) && B.alloc(&D.bp, D.np) && B.alloc(&D.S, nl2 + (size_t)16 * D.ld + (size_t)(D.ld / 16) * 256) && B.alloc(&D.bs, D.ld) && B.alloc(&D.xp, D.ld) && B.alloc_raw(&D.db, (size_t)npts * 3) && B.alloc_raw(&D.xl, (size_t)npts * 3) &&
              B.alloc(&d_pre, (size_t)n_local * 142, preint) && B.alloc(&d_info, info_pvr.size(), info_pvr.data()) &&
              B.alloc(&D.e_pvr, (size_t)n_local * 9) && B.alloc(&D.e_b, (size_t)n_local * 3) && B.alloc(&D.scal, 8) && B.alloc(&D.ctl, (size_t)BA_CTL_N) && B.alloc(&D.ticket, 1) && B.alloc(&d_erase, ne) && B.commit(lease.c);
    if (!ok) { set_error("device allocation / upload failed"); return VIORB_ERR_HIP; }
    D.e_pt = d_ept; D.e_kf = d_ekf; D.e_obs = d_obs; D.pt_start = d_pts; D.kf_start = d_kfs; D.kf_list = d_kfl; D.preint = d_pre; D.info_pvr = d_info;
    S.st = lease.c->st; S.h = lease.c->pinned; S.model = 0; S.stop = stop; S.d_erase = d_erase;
    VIORB_HIP_TRY(ba_map_stop_flag(S));
    S.kfs_out = kfs_out; S.points_out = points_out; S.erase = erase; S.info = info;
    return VIORB_OK;
}

extern "C" int viorb_local_ba_set_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device >= n) { set_error("no such HIP device: %d", device); return VIORB_ERR_INVALID_ARG; }
    g_lba_device.store(device < 0 ? -1 : device);
    return VIORB_OK;
}

extern "C" int viorb_local_ba_navstate(const double* kfs, int nk, int n_local, int prev_kf, const double* preint, const double* points, int npts,
                                       const int32_t* edge_idx, const double* edge_obs, int ne, const double gw[3], const double cam[16],
                                       const volatile int* stop, double* kfs_out, double* points_out, uint8_t* erase, double info[6]) {
    BaSolve S;
    const int rc = ba_prepare_navstate(S, kfs, nk, n_local, prev_kf, preint, points, npts, edge_idx, edge_obs, ne, gw, cam, stop, kfs_out, points_out, erase, info);
    return rc != VIORB_OK ? rc : S.run();
}

// One host thread keeps up to max_in_flight windows going, each on its own stream: whenever a window's stream has drained, its driver
// takes the next LM decision and enqueues the next batch of kernels. prepare(i, S) fills window i's solve and returns its status.
template <class Prepare, class SetStatus>
static int ba_run_batch(int n, int max_in_flight, Prepare prepare, SetStatus set_status) {
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    if (max_in_flight <= 0) max_in_flight = 16;
    VIORB_HIP_TRY(hipSetDevice(lba_device()));
    std::vector<std::unique_ptr<BaSolve>> live(std::min(max_in_flight, std::max(n, 1)));
    std::vector<int> which(live.size(), -1);
    int next = 0, finished = 0, first_error = VIORB_OK;
    auto start_next = [&](size_t slot) {
        while (next < n) {
            const int i = next++;
            auto S = std::make_unique<BaSolve>();
            int status = prepare(i, *S);
            if (status == VIORB_OK && S->state != BaSolve::ST_DONE) status = ba_launch_pairs(S->D, S->st);
            if (status == VIORB_OK && S->state != BaSolve::ST_DONE) {
                const int r = S->advance();                      // first batch of work
                if (r == BA_WAIT) { live[slot] = std::move(S); which[slot] = i; return; }
                status = r < 0 ? r : VIORB_OK;
            }
            set_status(i, status);
            if (status != VIORB_OK && first_error == VIORB_OK) first_error = status;
            finished++;
        }
        live[slot].reset(); which[slot] = -1;
    };
    for (size_t k = 0; k < live.size(); k++) start_next(k);
    while (finished < n) {
        bool progressed = false;
        for (size_t k = 0; k < live.size(); k++) {
            if (!live[k]) continue;
            const hipError_t e = hipStreamQuery(live[k]->st);
            if (e == hipErrorNotReady) continue;
            progressed = true;
            const int r = e == hipSuccess ? live[k]->advance() : VIORB_ERR_HIP;
            if (e != hipSuccess) set_error("hipStreamQuery failed: %s", hipGetErrorString(e));
            if (r == BA_WAIT) continue;
            set_status(which[k], r < 0 ? r : VIORB_OK);
            if (r < 0 && first_error == VIORB_OK) first_error = r;
            finished++;
            start_next(k);
        }
        if (!progressed) for (volatile int spin = 0; spin < 200; spin = spin + 1) {}
    }
    return first_error;
}

// Lock-step batch of NavState windows (kernels k_bab_*): returns the first error; every window's status through set_status.
template <class Win, class Prepare>
static int ba_run_lockstep(Win* w, int n, int max_in_flight, Prepare prepare) {
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    VIORB_HIP_TRY(hipSetDevice(lba_device()));
    int first_error = VIORB_OK;
    // windows advanced together (max_in_flight of the batch entry points; default 128): every solver step is one launch over the group
    const int GROUP = max_in_flight > 0 ? std::min(max_in_flight, 512) : 128, TB = 256, ROUNDS_PER_CHECK = 4, MAX_ROUNDS = 600;
    for (int g0 = 0; g0 < n; g0 += GROUP) {
        const int ng = std::min(GROUP, n - g0);
        std::vector<std::unique_ptr<BaSolve>> S(ng);
        std::vector<int> act;
        {   // argument checks, graph bookkeeping, input mirror and upload of every window: independent per window (own context, stream and
            // arena), ~0.1 ms of host time each — spread over a few host threads so that the device does not wait for them
            const int dev = lba_device();
            const int nthr = std::max(1, std::min(std::min(8, (int)std::thread::hardware_concurrency()), ng / 4));
            std::vector<std::string> errs(ng);
            auto work = [&](int t0) {
                (void)hipSetDevice(dev);
                for (int i = t0; i < ng; i += nthr) {
                    Win& q = w[g0 + i];
                    S[i] = std::make_unique<BaSolve>();
                    q.status = prepare(*S[i], q);
                    if (q.status != VIORB_OK) errs[i] = viorb_last_error();      // the error text is per thread
                }
            };
            std::vector<std::thread> pool;
            for (int t0 = 1; t0 < nthr; t0++) pool.emplace_back(work, t0);
            work(0);
            for (std::thread& th : pool) th.join();
            for (int i = 0; i < ng; i++) {
                const int rc = w[g0 + i].status;
                if (rc != VIORB_OK) { if (first_error == VIORB_OK) { first_error = rc; set_error("%s", errs[i].c_str()); } continue; }
                if (S[i]->state == BaSolve::ST_DONE) continue;           // stop flag already set: inputs copied to the outputs
                act.push_back(i);
            }
        }
        const int na = (int)act.size();
        if (na == 0) continue;
        // the group's solve; an early return (a HIP error, MAX_ROUNDS) must not leave status == VIORB_OK on windows whose outputs were never written
        auto solve_group = [&]() -> int {
        hipStream_t st = S[act[0]]->st;
        std::vector<BaDev> Dh(na); std::vector<uint8_t*> Eh(na);
        int gE = 1, gP = 1, Wmax = 1; size_t nl2 = 1, n2 = 1;
        for (int a = 0; a < na; a++) {
            BaSolve& B = *S[act[a]];
            VIORB_HIP_TRY(hipStreamSynchronize(B.st));                   // the window's inputs are uploaded
            Dh[a] = B.D; Dh[a].use_ctl = 1; Eh[a] = B.d_erase;
            gE = std::max(gE, (B.D.NE + TB - 1) / TB); gP = std::max(gP, (B.D.NP + TB - 1) / TB); Wmax = std::max(Wmax, B.D.W);
            nl2 = std::max(nl2, (size_t)B.D.ld * B.D.ld); n2 = std::max(n2, (size_t)B.D.np * B.D.np);
        }
        BaDev* Dv = nullptr; uint8_t** Ev = nullptr; int* d_done = nullptr;
        // the group's descriptor arrays: grow-only buffers of the calling thread (hipMalloc / hipFree synchronise the whole device)
        struct GroupBufs {
            void* p[3] = {nullptr, nullptr, nullptr}; int cap = 0, device = -1;
            void drop() { for (void*& q : p) { if (q) (void)hipFree(q); q = nullptr; } cap = 0; }
            ~GroupBufs() { drop(); }
        };
        static thread_local GroupBufs gb;
        if (gb.device != lba_device() || gb.cap < na) {
            gb.drop(); gb.device = lba_device();
            const int want = std::max(na, 128);
            VIORB_HIP_TRY(hipMalloc(&gb.p[0], sizeof(BaDev) * want)); VIORB_HIP_TRY(hipMalloc(&gb.p[1], sizeof(uint8_t*) * want)); VIORB_HIP_TRY(hipMalloc(&gb.p[2], sizeof(int)));
            gb.cap = want;
        }
        Dv = (BaDev*)gb.p[0]; Ev = (uint8_t**)gb.p[1]; d_done = (int*)gb.p[2];
        VIORB_HIP_TRY(hipMemcpyAsync(Dv, Dh.data(), sizeof(BaDev) * na, hipMemcpyHostToDevice, st));
        VIORB_HIP_TRY(hipMemcpyAsync(Ev, Eh.data(), sizeof(uint8_t*) * na, hipMemcpyHostToDevice, st));
        VIORB_HIP_TRY(hipMemsetAsync(d_done, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_bab_ctl_init, dim3(na), dim3(64), 0, st, Dv);
        hipLaunchKernelGGL(k_bab_pairs_build, dim3(na), dim3(1024), 0, st, Dv);            // the Schur complement's gather lists of every window
        double* pin = S[act[0]]->h;                                      // page-locked scratch of the first window's context
        int* h_done = reinterpret_cast<int*>(pin + 48);
        const unsigned gw = (unsigned)((na + 63) / 64), gR = (unsigned)((nl2 + TB - 1) / TB), gC = (unsigned)std::min<size_t>((n2 + TB - 1) / TB, 64);
        const dim3 Y1(1, na), YE(gE, na), YP(std::max(gP, 1), na);
        const int gL = std::max(gE, gP), g8 = std::max(8 * gP, 1);      // blocks of the linearisation (edges or points, by the kind of window) and of the 8-lane back-substitution
        int rounds = 0, done = 0;
        std::vector<char> aborted(na, 0);
        std::vector<BaSolve*> Sp(na);
        for (int a = 0; a < na; a++) Sp[a] = S[act[a]].get();
        while (done < na && rounds < MAX_ROUNDS) {
            for (int r = 0; r < ROUNDS_PER_CHECK; r++, rounds++) {
                // (ProfScope: HIP-event pairs around the heavy launches when viorb_profile_enable is on — bench.py --config local_ba; free otherwise)
                hipLaunchKernelGGL(k_bab_round_begin, dim3(gw), dim3(64), 0, st, Dv, na);
                { ProfScope ps("k_bab_errors_chi", st); hipLaunchKernelGGL(k_bab_errors_chi, YE, dim3(TB), 0, st, Dv); }
                hipLaunchKernelGGL(k_bab_take_chi, dim3(gw), dim3(64), 0, st, Dv, na);
                { ProfScope ps("k_bab_f_lin_clear", st); hipLaunchKernelGGL(k_bab_f_lin_clear, dim3((unsigned)gL + gC, na), dim3(TB), 0, st, Dv, gL, (int)gC); }
                { ProfScope ps("k_bab_f_hpp_imu_hll", st); hipLaunchKernelGGL(k_bab_f_hpp_imu_hll, dim3((unsigned)(2 * Wmax + gP), na), dim3(256), 0, st, Dv, Wmax); }
                hipLaunchKernelGGL(k_bab_max_diag, dim3(8, na), dim3(256), 0, st, Dv);
                hipLaunchKernelGGL(k_bab_lambda0, dim3(gw), dim3(64), 0, st, Dv, na);
                { ProfScope ps("k_bab_f_init_dinv", st); hipLaunchKernelGGL(k_bab_f_init_dinv, dim3(gR + (unsigned)gP, na), dim3(TB), 0, st, Dv, (int)gR); }
                { ProfScope ps("k_bab_schur", st);
                  hipLaunchKernelGGL(k_bab_schur, dim3((unsigned)(Wmax * (Wmax + 1) / 2) * 8u * (unsigned)((na + 7) / 8)), dim3(64), 0, st, Dv, Wmax * (Wmax + 1) / 2, na); }
                (void)raise_dynamic_lds(reinterpret_cast<const void*>(k_bab_chol_solve), BA_CHOL_LDS_BYTES);
                { ProfScope ps("k_bab_chol_solve", st); hipLaunchKernelGGL(k_bab_chol_solve, Y1, dim3(BA_CHOL_THREADS), BA_CHOL_LDS_BYTES, st, Dv); }
                { ProfScope ps("k_bab_f_backsub8_update", st); hipLaunchKernelGGL(k_bab_f_backsub8_update, dim3((unsigned)g8, na), dim3(TB), 0, st, Dv); }
                { ProfScope ps("k_bab_errors", st); hipLaunchKernelGGL(k_bab_errors, YE, dim3(TB), 0, st, Dv); }
                hipLaunchKernelGGL(k_bab_decide, dim3(gw), dim3(64), 0, st, Dv, na);
                hipLaunchKernelGGL(k_bab_restore, YP, dim3(TB), 0, st, Dv);
                hipLaunchKernelGGL(k_bab_gate, YE, dim3(TB), 0, st, Dv, Ev);
                hipLaunchKernelGGL(k_bab_phase, dim3(gw), dim3(64), 0, st, Dv, na, d_done);
            }
            VIORB_HIP_TRY(hipGetLastError());
            VIORB_HIP_TRY(hipMemcpyAsync(h_done, d_done, sizeof(int), hipMemcpyDeviceToHost, st));
            VIORB_HIP_TRY(ba_wait(st, Sp.data(), na));
            done = *h_done;
            // the callers' stop flags: g2o polls pbStopFlag once per iteration and per trial; k_bab_decide does the same through the mirrored
            // words ba_wait keeps up to date, and the host marks the window here so that the next rounds skip its kernels altogether
            for (int a = 0; a < na; a++) {
                BaSolve& B = *S[act[a]];
                if (!aborted[a] && B.terminate()) {
                    aborted[a] = 1;
                    static const double one = 1.0;                 // static storage: outlives the asynchronous copy
                    VIORB_HIP_TRY(hipMemcpyAsync(B.D.ctl + BA_B_ABORT, &one, sizeof(double), hipMemcpyHostToDevice, st));
                }
            }
        }
        if (done < na) { set_error("window solve did not finish in %d rounds", MAX_ROUNDS); return VIORB_ERR_HIP; }
        // results: packed by one kernel into a group buffer on the device, ONE copy into page-locked memory, then plain memcpy into the callers'
        // (pageable) arrays
        {
            std::vector<unsigned long long> offs(4 * (size_t)na);
            size_t total = 0, max_ne = 1;
            auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
            for (int a = 0; a < na; a++) {
                const BaDev& Da = S[act[a]]->D;
                offs[4 * a] = total; total += up((size_t)Da.W * Da.kf_stride * sizeof(double));
                offs[4 * a + 1] = total; total += up((size_t)Da.NP * 3 * sizeof(double));
                offs[4 * a + 2] = total; total += up((size_t)Da.NE);
                offs[4 * a + 3] = total; total += up(sizeof(double) * BA_B_N);
                max_ne = std::max(max_ne, (size_t)std::max(Da.NE, std::max(Da.NP * 3, Da.W * Da.kf_stride)));
            }
            // grow-only buffers of the calling thread (hipMalloc / hipHostMalloc synchronise the device and take longer than a round)
            struct Bufs {
                void* dev = nullptr; void* host = nullptr; void* doffs = nullptr; size_t bytes = 0, noffs = 0; int device = -1;
                void drop() { if (dev) (void)hipFree(dev); if (host) (void)hipHostFree(host); if (doffs) (void)hipFree(doffs); dev = host = doffs = nullptr; bytes = noffs = 0; }
                ~Bufs() { drop(); }
            };
            static thread_local Bufs bufs;
            if (bufs.device != lba_device() || bufs.bytes < total || bufs.noffs < offs.size()) {
                bufs.drop(); bufs.device = lba_device();
                const size_t want = total + total / 4, wo = offs.size() + 64;
                VIORB_HIP_TRY(hipMalloc(&bufs.dev, want)); VIORB_HIP_TRY(hipHostMalloc(&bufs.host, want)); VIORB_HIP_TRY(hipMalloc(&bufs.doffs, wo * sizeof(unsigned long long)));
                bufs.bytes = want; bufs.noffs = wo;
            }
            VIORB_HIP_TRY(hipMemcpyAsync(bufs.doffs, offs.data(), offs.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_bab_pack, dim3((unsigned)std::min<size_t>((max_ne + TB - 1) / TB, 64), na), dim3(TB), 0, st, Dv, Ev, static_cast<uint8_t*>(bufs.dev),
                               static_cast<const unsigned long long*>(bufs.doffs));
            VIORB_HIP_TRY(hipGetLastError());
            VIORB_HIP_TRY(hipMemcpyAsync(bufs.host, bufs.dev, total, hipMemcpyDeviceToHost, st));
            VIORB_HIP_TRY(hipStreamSynchronize(st));
            const uint8_t* hb = static_cast<const uint8_t*>(bufs.host);
            for (int a = 0; a < na; a++) {
                BaSolve& B = *S[act[a]];
                memcpy(B.kfs_out, hb + offs[4 * a], (size_t)B.D.W * B.D.kf_stride * sizeof(double));
                memcpy(B.points_out, hb + offs[4 * a + 1], (size_t)B.D.NP * 3 * sizeof(double));
                memcpy(B.erase, hb + offs[4 * a + 2], (size_t)B.D.NE);
                const double* c = reinterpret_cast<const double*>(hb + offs[4 * a + 3]);
                double* info = B.info;
                info[0] = c[BA_B_CHI0]; info[1] = c[BA_B_CHI1]; info[2] = c[BA_B_ITS0]; info[3] = c[BA_B_ITS1];
            }
        }
        return VIORB_OK;
        };
        const int grc = solve_group();
        if (grc != VIORB_OK) {
            for (int i : act) w[g0 + i].status = grc;
            for (int j = g0 + ng; j < n; j++) w[j].status = grc;
            return grc;
        }
    }
    return first_error;
}

extern "C" int viorb_local_ba_navstate_batch(viorb_lba_window* w, int n, int max_in_flight) {
    VIORB_REQUIRE(w && n >= 0, "null windows");
    static const bool per_stream = getenv("VIORB_LBA_STREAMS") != nullptr;      // the round-1 driver: one stream and host LM loop per window
    if (!per_stream) return ba_run_lockstep(w, n, max_in_flight, [](BaSolve& S, viorb_lba_window& q) {
        return ba_prepare_navstate(S, q.kfs, q.nk, q.n_local, q.prev_kf, q.preint, q.points, q.np, q.edge_idx, q.edge_obs, q.ne, q.gw, q.cam, q.stop,
                                   q.kfs_out, q.points_out, q.erase, q.info);
    });
    return ba_run_batch(n, max_in_flight,
        [&](int i, BaSolve& S) {
            viorb_lba_window& q = w[i];
            return ba_prepare_navstate(S, q.kfs, q.nk, q.n_local, q.prev_kf, q.preint, q.points, q.np, q.edge_idx, q.edge_obs, q.ne, q.gw, q.cam, q.stop,
                                       q.kfs_out, q.points_out, q.erase, q.info);
        },
        [&](int i, int status) { w[i].status = status; });
}

static int ba_prepare_se3(BaSolve& S, const double* kfs, int nk, int n_local, const double* points, int npts, const int32_t* edge_idx,
                          const double* edge_obs, int ne, const double intr5[5], const volatile int* stop, double* kfs_out,
                          double* points_out, uint8_t* erase, double info[6]) {
    VIORB_REQUIRE(kfs && points && edge_idx && edge_obs && intr5 && kfs_out && points_out && erase && info, "null array");
    VIORB_REQUIRE(n_local >= 1 && n_local <= 40 && nk >= n_local && npts >= 1 && ne >= 1, "1 <= n_local <= 40 key frames, at least one point and edge");
    if (viorb_device_count() < 1) { set_error("no HIP device: libviorb_hip has no CPU fallback"); return VIORB_ERR_NO_DEVICE; }
    for (int i = 0; i < 6; i++) info[i] = 0;
    for (int i = 0; i < n_local * 7; i++) kfs_out[i] = kfs[i];
    for (int i = 0; i < npts * 3; i++) points_out[i] = points[i];
    for (int k = 0; k < ne; k++) erase[k] = 0;
    if (stop && *stop) { S.state = BaSolve::ST_DONE; return VIORB_OK; }
    std::vector<int> e_pt(ne), e_kf(ne), pt_start(npts + 1, 0);
    for (int k = 0; k < ne; k++) {
        e_pt[k] = edge_idx[2 * k]; e_kf[k] = edge_idx[2 * k + 1];
        VIORB_REQUIRE(e_pt[k] >= 0 && e_pt[k] < npts && e_kf[k] >= 0 && e_kf[k] < nk, "edge index out of range");
        VIORB_REQUIRE(k == 0 || e_pt[k] >= e_pt[k - 1], "edges must be grouped by point (ascending point index)");
        pt_start[e_pt[k] + 1]++;
    }
    for (int p = 0; p < npts; p++) pt_start[p + 1] += pt_start[p];
    std::vector<int> kf_start(n_local + 1, 0), kf_list;
    for (int k = 0; k < ne; k++) if (e_kf[k] < n_local) kf_start[e_kf[k] + 1]++;
    for (int i = 0; i < n_local; i++) kf_start[i + 1] += kf_start[i];
    kf_list.resize(kf_start[n_local]);
    { std::vector<int> pos(kf_start.begin(), kf_start.end() - 1); for (int k = 0; k < ne; k++) if (e_kf[k] < n_local) kf_list[pos[e_kf[k]]++] = k; }
    const size_t pr_bound = ba_pairs_bound(e_kf, pt_start, npts, n_local);
    VIORB_HIP_TRY(hipSetDevice(lba_device()));
    BaCtxLease& lease = S.lease;
    if (!lease.ready()) { set_error("could not create a HIP stream"); return VIORB_ERR_HIP; }
    BaBuf B; BaDev& D = S.D;
    D.W = n_local; D.NK = nk; D.NP = npts; D.NE = ne; D.np = 6 * n_local; D.prev_kf = -1; D.acc_bias_rw2 = 0;
    D.pose_dim = 6; D.rows = 3; D.kf_stride = 7;
    for (int i = 0; i < 16; i++) D.cam[i] = i < 5 ? intr5[i] : 0.0;
    for (int i = 0; i < 3; i++) D.gw[i] = 0;
    D.ld = (D.np + 15) & ~15;
    const size_t n2 = (size_t)D.np * D.np, nl2 = (size_t)D.ld * D.ld;
    int *d_ept, *d_ekf, *d_pts, *d_kfs, *d_kfl; double* d_obs; uint8_t* d_erase;
    D.preint = nullptr; D.info_pvr = nullptr; D.e_pvr = nullptr; D.e_b = nullptr;
    bool ok = B.alloc(&D.kf, (size_t)nk * 7, kfs) && B.alloc_raw(&D.kf_bak, (size_t)nk * 7) && B.alloc(&D.pt, (size_t)npts * 3, points) && B.alloc_raw(&D.pt_bak, (size_t)npts * 3) &&
              B.alloc(&d_ept, ne, e_pt.data()) && B.alloc(&d_ekf, ne, e_kf.data()) && B.alloc(&d_obs, (size_t)ne * 4, edge_obs) && B.alloc(&D.level, ne) &&
              B.alloc_raw(&D.err, (size_t)ne * 3) && B.alloc_raw(&D.Jp, (size_t)ne * 9) && B.alloc_raw(&D.Jk, (size_t)ne * 18) && B.alloc_raw(&D.wgt, ne) && B.alloc_raw(&D.We, (size_t)ne * 18) &&
              B.alloc(&D.pr_start, (size_t)n_local * (n_local + 1) / 2 + 1) && B.alloc_raw(&D.pr_ent, 2 * pr_bound + 2) &&
              B.alloc(&d_pts, npts + 1, pt_start.data()) && B.alloc(&d_kfs, n_local + 1, kf_start.data()) && B.alloc(&d_kfl, kf_list.size(), kf_list.data()) &&
              B.alloc_raw(&D.Hll, (size_t)npts * 9) && B.alloc_raw(&D.bl, (size_t)npts * 3) && B.alloc_raw(&D.Dinv, (size_t)npts * 9) &&
              B.alloc_raw(&D.Hpp, n2) && B.alloc(&D.bp, D.np) && B.alloc(&D.S, nl2 + (size_t)16 * D.ld + (size_t)(D.ld / 16) * 256) && B.alloc(&D.bs, D.ld) && B.alloc(&D.xp, D.ld) &&
              B.alloc_raw(&D.db, (size_t)npts * 3) && B.alloc_raw(&D.xl, (size_t)npts * 3) && B.alloc(&D.scal, 8) && B.alloc(&D.ctl, (size_t)BA_CTL_N) && B.alloc(&D.ticket, 1) && B.alloc(&d_erase, ne) && B.commit(lease.c);
    if (!ok) { set_error("device allocation / upload failed"); return VIORB_ERR_HIP; }
    D.e_pt = d_ept; D.e_kf = d_ekf; D.e_obs = d_obs; D.pt_start = d_pts; D.kf_start = d_kfs; D.kf_list = d_kfl;
    S.st = lease.c->st; S.h = lease.c->pinned; S.model = 1; S.stop = stop; S.d_erase = d_erase;
    VIORB_HIP_TRY(ba_map_stop_flag(S));
    S.kfs_out = kfs_out; S.points_out = points_out; S.erase = erase; S.info = info;
    return VIORB_OK;
}

extern "C" int viorb_local_ba_se3(const double* kfs, int nk, int n_local, const double* points, int npts, const int32_t* edge_idx,
                                  const double* edge_obs, int ne, const double intr5[5], const volatile int* stop, double* kfs_out,
                                  double* points_out, uint8_t* erase, double info[6]) {
    BaSolve S;
    const int rc = ba_prepare_se3(S, kfs, nk, n_local, points, npts, edge_idx, edge_obs, ne, intr5, stop, kfs_out, points_out, erase, info);
    return rc != VIORB_OK ? rc : S.run();
}

extern "C" int viorb_local_ba_se3_batch(viorb_lba_se3_window* w, int n, int max_in_flight) {
    VIORB_REQUIRE(w && n >= 0, "null windows");
    static const bool per_stream = getenv("VIORB_LBA_STREAMS") != nullptr;      // the round-1 driver: one stream and host LM loop per window
    if (!per_stream) return ba_run_lockstep(w, n, max_in_flight, [](BaSolve& S, viorb_lba_se3_window& q) {
        return ba_prepare_se3(S, q.kfs, q.nk, q.n_local, q.points, q.np, q.edge_idx, q.edge_obs, q.ne, q.intr5, q.stop, q.kfs_out, q.points_out, q.erase, q.info);
    });
    return ba_run_batch(n, max_in_flight,
        [&](int i, BaSolve& S) {
            viorb_lba_se3_window& q = w[i];
            return ba_prepare_se3(S, q.kfs, q.nk, q.n_local, q.points, q.np, q.edge_idx, q.edge_obs, q.ne, q.intr5, q.stop, q.kfs_out, q.points_out, q.erase, q.info);
        },
        [&](int i, int status) { w[i].status = status; });
}
