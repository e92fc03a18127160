// VALU issue-rate micro-benchmark (dev aid): SIMD cycles per wave64 instruction for every opcode class the hot kernels issue, by
// number of waves resident on the SIMD. Settles what "vector-issue bound" means on gfx950: does a wave64
// instruction occupy its SIMD for 4 cycles (SIMD-16 pass rate) or 2 (SIMD-32), and for which opcodes?
//
// Method: ONE workgroup of 256 x W threads on one CU (W waves on each of the 4 SIMDs), every wave runs N_ITER x REP independent
// instances of the opcode over 8 register chains (no dependent-issue stalls at W >= 1: 8 chains cover the ~8-cycle ALU latency) between two
// s_memtime reads; cycles per instruction per SIMD = (latest end - earliest start of the workgroup) / (instructions per wave x W).
// A second figure is chip-wide wall clock (2048 workgroups of 256 threads = 8 waves per SIMD everywhere).
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#define REP 256
enum { OP_MIN_I32, OP_ADD_U32, OP_AND, OP_LSHL_ADD, OP_MAD_U24, OP_MUL_LO, OP_MUL_HI_U24, OP_MAX3_I32, OP_PERM, OP_ALIGNBYTE, OP_BFE, OP_BCNT, OP_CNDMASK,
       OP_CMP, OP_PK_MIN_I16, OP_PK_SUB_I16, OP_PK_MAD_U16, OP_DOT4_U8, OP_DOT2_U16, OP_SAD_U8, OP_ADD_F32, OP_FMA_F32, OP_PK_FMA_F32, OP_MAX3_F32, OP_CVT_UBYTE, OP_RCP_F32,
       OP_FMA_F64, OP_MUL_F64, OP_ADD_F64, OP_RCP_F64, OP_RSQ_F64, OP_MOV_DPP, OP_READLANE,
       OP_SUB_U32, OP_OR, OP_XOR, OP_LSHLREV, OP_LSHRREV, OP_MOV, OP_MUL_F32, OP_MAX_I32, OP_MIN_U32, OP_AND_OR, OP_ADD3, OP_MUL_U24, OP_PK_ADD_U16, OP_ADD_U16, OP_MAX_I16, OP_BFI, OP_XAD, OP_ADD_CO,
       OP_CNDMASK_SGPR, OP_CNDMASK_AFTER_CMP, OP_CMP_SDST, OP_CNDMASK_ZERO, OP_SUB_F32, OP_MAX_F32, OP_ASHRREV, OP_LSHL_OR, OP_OR3, OP_PERMLANE32_SWAP, OP_PERMLANE16_SWAP, OP_MOV_DPP_BCAST,
       OP_CMP_CND2, OP_CMP_CND4, OP_CMP_CND8, OP_CMP_CND4_SGPR, OP_SMOV_CND4, OP_FMAC_F32, OP_FMAC_F64, OP_MIN_I16, OP_SUB_U16, OP_LSHLREV_B16, OP_ADD_U32_SGPR, OP_ADD_U32_LIT, OP_AND_LIT, OP_SUBREV_U32, OP_MUL_LO_U16, OP_CND_VCC_NOP, OP_N };
static const char* kNames[OP_N] = {"v_min_i32", "v_add_u32", "v_and_b32", "v_lshl_add_u32", "v_mad_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32_u24", "v_max3_i32", "v_perm_b32",
       "v_alignbyte_b32", "v_bfe_u32", "v_bcnt_u32_b32", "v_cndmask_b32", "v_cmp_lt_i32 (vcc)", "v_pk_min_i16", "v_pk_sub_i16", "v_pk_mad_u16", "v_dot4_u32_u8", "v_dot2_u32_u16",
       "v_sad_u8", "v_add_f32", "v_fma_f32", "v_pk_fma_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_rcp_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64",
       "v_mov_b32 dpp row_shr:1", "v_readlane_b32",
       "v_sub_u32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_mov_b32", "v_mul_f32", "v_max_i32", "v_min_u32", "v_and_or_b32", "v_add3_u32", "v_mul_u32_u24", "v_pk_add_u16",
       "v_add_u16", "v_max_i16", "v_bfi_b32", "v_xad_u32", "v_add_co_u32 (vcc)", "v_cndmask_b32 (sgpr pair)", "v_cmp + v_cndmask (pair)", "v_cmp_lt_i32 (sgpr dst)",
       "v_cndmask_b32 (vcc = 0)", "v_sub_f32", "v_max_f32", "v_ashrrev_i32", "v_lshl_or_b32", "v_or3_b32", "v_permlane32_swap_b32", "v_permlane16_swap_b32", "v_mov_b32 dpp row_bcast:15",
       "v_cmp + 2 v_cndmask vcc (3)", "v_cmp + 4 v_cndmask vcc (5)", "v_cmp + 8 v_cndmask vcc (9)", "v_cmp sdst + 4 cndmask s (5)", "s_mov vcc + 4 v_cndmask (5)", "v_fmac_f32 (VOP2)", "v_fmac_f64 (VOP2)",
       "v_min_i16", "v_sub_u16", "v_lshlrev_b16", "v_add_u32 v, s, v", "v_add_u32 v, literal, v", "v_and_b32 v, literal, v", "v_subrev_u32", "v_mul_lo_u16", "v_cndmask vcc + s_nop 4"};

template <int OP> __device__ __forceinline__ void body(int (&a)[8], float (&fa)[8], double (&da)[8], int b, float fb, double db) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == OP_MIN_I32) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MUL_HI_U24) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MAX3_I32) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[i]));
            if (OP == OP_BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            if (OP == OP_CMP) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (OP == OP_PK_MIN_I16) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_PK_SUB_I16) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_PK_MAD_U16) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OP_DOT4_U8) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OP_DOT2_U16) asm volatile("v_dot2_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OP_SAD_U8) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OP_ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
            if (OP == OP_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fa[i]) : "v"(fb));
            if (OP == OP_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(da[i]) : "v"(db));
            if (OP == OP_MAX3_F32) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(fa[i]) : "v"(fb));
            if (OP == OP_CVT_UBYTE) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "+v"(fa[i]) : "v"(b));
            if (OP == OP_RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(fa[i]));
            if (OP == OP_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(da[i]) : "v"(db));
            if (OP == OP_MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(da[i]) : "v"(db));
            if (OP == OP_ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(da[i]) : "v"(db));
            if (OP == OP_RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(da[i]));
            if (OP == OP_RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(da[i]));
            if (OP == OP_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (OP == OP_SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == OP_LSHRREV) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
            if (OP == OP_MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_ADD_U16) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MAX_I16) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_XAD) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_ADD_CO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == OP_CNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
            if (OP == OP_CNDMASK_AFTER_CMP) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == OP_CMP_SDST) asm volatile("v_cmp_lt_i32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
            if (OP == OP_CNDMASK_ZERO) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
            if (OP == OP_SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
            if (OP == OP_MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
            if (OP == OP_ASHRREV) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i]));
            if (OP == OP_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_OR3) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b));
            if (OP == OP_PERMLANE16_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b));
            if (OP == OP_MOV_DPP_BCAST) asm volatile("v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]));
            if (OP == OP_CMP_CND2) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == OP_CMP_CND4) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == OP_CMP_CND8) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == OP_CMP_CND4_SGPR) asm volatile("v_cmp_lt_i32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21");
            if (OP == OP_SMOV_CND4) asm volatile("s_mov_b64 vcc, 0x0f0f0f0f\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == OP_FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(fa[i]) : "v"(fb));
            if (OP == OP_FMAC_F64) asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(da[i]) : "v"(db));
            if (OP == OP_MIN_I16) asm volatile("v_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_SUB_U16) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_LSHLREV_B16) asm volatile("v_lshlrev_b16 %0, 1, %0" : "+v"(a[i]));
            if (OP == OP_ADD_U32_SGPR) asm volatile("v_add_u32 %0, s22, %0" : "+v"(a[i]));
            if (OP == OP_ADD_U32_LIT) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i]));
            if (OP == OP_AND_LIT) asm volatile("v_and_b32 %0, 0x00ff00ff, %0" : "+v"(a[i]));
            if (OP == OP_SUBREV_U32) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MUL_LO_U16) asm volatile("v_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_CND_VCC_NOP) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n s_nop 4" : "+v"(a[i]) : "v"(b));
            if (OP == OP_READLANE) { int s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a[i])); asm volatile("" :: "s"(s)); }
        }
    }
}

template <int OP> __global__ __launch_bounds__(1024) void k(int* out, unsigned long long* stamps, int n_iter, int a0, int b0) {
    int a[8], b = b0 + threadIdx.x;
    float fa[8], fb = 1.0f + 1e-7f * (float)b; double da[8], db = 1.0 + 1e-9 * (double)b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = a0 + i + threadIdx.x; fa[i] = 1.0f + (float)a[i] * 1e-6f; da[i] = fa[i]; }
    if (OP == OP_CNDMASK_ZERO) asm volatile("s_mov_b64 vcc, 0" ::: "vcc");
    if (OP == OP_CNDMASK) asm volatile("s_mov_b64 vcc, 0x0f0f0f0f" ::: "vcc");
    if (OP == OP_CNDMASK_SGPR) asm volatile("s_mov_b64 s[20:21], 0x0f0f0f0f" ::: "s20", "s21");
    if (OP == OP_ADD_U32_SGPR) asm volatile("s_mov_b32 s22, 77" ::: "s22");
    if (OP == OP_CND_VCC_NOP) asm volatile("s_mov_b64 vcc, 0x0f0f0f0f" ::: "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n_iter; it++) body<OP>(a, fa, da, b, fb, db);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0; float fs = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { s += a[i]; fs += fa[i] + (float)da[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (int)fs;
    if (stamps && (threadIdx.x & 63) == 0) { stamps[2 * (threadIdx.x >> 6)] = t0; stamps[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int OP> void run(int* d, unsigned long long* d_st) {
    double cyc[4]; const int ws[4] = {1, 2, 4, 8 > 4 ? 4 : 4};
    const int iters = 200;
    int col = 0;
    for (int W : {1, 2, 3, 4}) {                                         // waves per SIMD on one CU (1024 threads = 16 waves = 4 per SIMD at most)
        const int threads = 256 * W;
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, d_st, 2, 1, 2);
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, d_st, iters, 1, 2);
        hipDeviceSynchronize();
        unsigned long long st[32]; hipMemcpy(st, d_st, sizeof(unsigned long long) * 2 * (threads / 64), hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < threads / 64; w++) { lo = std::min(lo, st[2 * w]); hi = std::max(hi, st[2 * w + 1]); }
        cyc[col++] = (double)(hi - lo) / ((double)iters * REP * W);
    }
    (void)ws;
    // chip-wide wall clock, 8 waves per SIMD
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, (unsigned long long*)nullptr, 2, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, (unsigned long long*)nullptr, iters, 1, 2);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * REP;
    printf("%-26s %6.2f %6.2f %6.2f %6.2f   | chip-wide %7.3f ms = %7.1f G wave-instr/s = %.2f ns per instruction per SIMD\n", kNames[OP], cyc[0], cyc[1], cyc[2], cyc[3], ms,
           winstr / ms / 1e6, ms * 1e6 * 1024 / winstr);
    hipEventDestroy(e0); hipEventDestroy(e1);
}
template <int OP> void run_all(int* d, unsigned long long* st) { run<OP>(d, st); if constexpr (OP + 1 < OP_N) run_all<OP + 1>(d, st); }
int main() {
    int* d; hipMalloc(&d, 256 * 8 * 256 * 4); unsigned long long* st; hipMalloc(&st, 64 * 8);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clock %d MHz; s_memtime ticks (shader cycles? see the ns column: cycles = ns x GHz) per wave64 instruction per SIMD\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    printf("%-26s %6s %6s %6s %6s\n", "opcode", "W=1", "W=2", "W=3", "W=4");
    run_all<0>(d, st);
    return 0;
}
