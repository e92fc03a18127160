// v_cndmask_b32 reading VCC: how many SIMD cycles by encoding (VOP2 e32 with the implicit vcc / VOP3 e64 with vcc or another SGPR pair named explicitly)
// and by distance from the instruction that wrote the mask. One workgroup of 256 x W threads on one CU, s_memtime around N x REP
// repetitions of the sequence, 8 register chains; reported: cycles per SEQUENCE per SIMD (divide by the instruction count in brackets).
// Build: hipcc --offload-arch=gfx950 -O2 cndmask_vcc.hip -o cndmask_vcc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#define REP 128
#define F "v_add_u32 %2, %2, %1\n"
enum { S_E32_VCC, S_E64_VCC, S_E64_SGPR, S_CMP_E32, S_CMP_1F_E32, S_CMP_2F_E32, S_CMP_4F_E32, S_CMP_8F_E32, S_CMP_E32X2, S_CMP_E32X4, S_CMP_E64VCC_X4, S_CMP_SDST_E64X4, S_SAND_E32, S_SAND_E64,
       S_ADDC, S_CMP_E32_E32_FILL, S_N };
static const char* kNames[S_N] = {
    "v_cndmask e32 (vcc const)                 [1]", "v_cndmask e64 ..., vcc (vcc const)        [1]", "v_cndmask e64 ..., s[20:21] (const)       [1]",
    "v_cmp vcc; v_cndmask e32                  [2]", "v_cmp vcc; 1 add; v_cndmask e32           [3]", "v_cmp vcc; 2 add; v_cndmask e32           [4]",
    "v_cmp vcc; 4 add; v_cndmask e32           [6]", "v_cmp vcc; 8 add; v_cndmask e32          [10]", "v_cmp vcc; 2 x v_cndmask e32              [3]",
    "v_cmp vcc; 4 x v_cndmask e32              [5]", "v_cmp vcc; 4 x v_cndmask e64 vcc          [5]", "v_cmp s[20:21]; 4 x v_cndmask e64 s[20:21][5]",
    "s_and_b64 vcc; v_cndmask e32              [2]", "s_and_b64 vcc; v_cndmask e64 vcc          [2]", "v_add_co_u32 vcc; v_addc_co_u32 vcc       [2]",
    "v_cmp vcc; e32; add; e32; add; e32        [6]"};
template <int SEQ> __device__ __forceinline__ void body(int (&a)[8], int b, int (&c)[8]) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (SEQ == S_E32_VCC) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]));
            if (SEQ == S_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]));
            if (SEQ == S_E64_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c[i]));
            if (SEQ == S_CMP_E32) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_CMP_1F_E32) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n" F "v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_CMP_2F_E32) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n" F F "v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_CMP_4F_E32) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n" F F F F "v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_CMP_8F_E32) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n" F F F F F F F F "v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_CMP_E32X2) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_CMP_E32X4) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %1, vcc\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_CMP_E64VCC_X4) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %2, %2, %1, vcc\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %2, %2, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_CMP_SDST_E64X4) asm volatile("v_cmp_lt_i32 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %1, s[20:21]\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %1, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "s20", "s21");
            if (SEQ == S_SAND_E32) asm volatile("s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_SAND_E64) asm volatile("s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
            if (SEQ == S_ADDC) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[i]), "+v"(c[i]) : "v"(b) : "vcc");
            if (SEQ == S_CMP_E32_E32_FILL) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n" F "v_cndmask_b32_e32 %0, %0, %1, vcc\n" F "v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b), "v"(c[i]) : "vcc");
        }
    }
}
template <int SEQ> __global__ __launch_bounds__(1024) void k(int* out, unsigned long long* stamps, int n_iter, int a0, int b0) {
    int a[8], c[8], b = b0 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = a0 + i + threadIdx.x; c[i] = a0 * 3 + i; }
    asm volatile("s_mov_b64 vcc, 0x0f0f0f0f\n s_mov_b64 s[20:21], 0x0f0f0f0f\n s_mov_b64 s[22:23], 0x3c3c3c3c" ::: "vcc", "s20", "s21", "s22", "s23");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n_iter; it++) body<SEQ>(a, b, c);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { stamps[2 * (threadIdx.x >> 6)] = t0; stamps[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int SEQ> void run(int* d, unsigned long long* d_st) {
    double cyc[4]; int col = 0; const int iters = 100;
    for (int W : {1, 2, 3, 4}) {
        const int threads = 256 * W;
        hipLaunchKernelGGL(k<SEQ>, dim3(1), dim3(threads), 0, 0, d, d_st, 2, 1, 2);
        hipLaunchKernelGGL(k<SEQ>, dim3(1), dim3(threads), 0, 0, d, d_st, iters, 1, 2);
        (void)hipDeviceSynchronize();
        unsigned long long st[32]; (void)hipMemcpy(st, d_st, sizeof(unsigned long long) * 2 * (threads / 64), hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < threads / 64; w++) { lo = std::min(lo, st[2 * w]); hi = std::max(hi, st[2 * w + 1]); }
        cyc[col++] = (double)(hi - lo) / ((double)iters * REP * W);
    }
    printf("%-52s W=1 %7.2f  W=2 %7.2f  W=3 %7.2f  W=4 %7.2f\n", kNames[SEQ], cyc[0], cyc[1], cyc[2], cyc[3]);
}
template <int SEQ> void run_all(int* d, unsigned long long* st) { run<SEQ>(d, st); if constexpr (SEQ + 1 < S_N) run_all<SEQ + 1>(d, st); }
int main() {
    int* d; (void)hipMalloc(&d, 1024 * 4); unsigned long long* st; (void)hipMalloc(&st, 64 * 8);
    printf("SIMD cycles per sequence [instructions in the sequence]; an ordinary VALU instruction costs ~4.2 (add / and / or ... ~2.3 at W >= 2)\n");
    run_all<0>(d, st);
    return 0;
}
