// One wave per SIMD: cycles per v_fma_f64 / v_mul_f64 / v_add_f64 when 1, 2, 4 or 8 independent dependency chains are in flight
// (chains = 1 is the back-to-back dependent latency, chains = 8 the issue rate). Answers how much instruction-level parallelism a
// single-wave f64 loop (the pose solver's edge loop) needs. Build: hipcc --offload-arch=gfx950 -O2 f64_latency.hip -o f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void k_fma(double* out, unsigned long long* cyc, int n) {
    double a[CH];
    for (int c = 0; c < CH; c++) a[c] = 1.0 + threadIdx.x * 1e-3 + c;
    const double m = 1.0000001, d = 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int c = 0; c < CH; c++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(d));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int c = 0; c < CH; c++) s += a[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH>
__global__ void k_rcp(double* out, unsigned long long* cyc, int n) {
    double a[CH];
    for (int c = 0; c < CH; c++) a[c] = 1.0 + threadIdx.x * 1e-3 + c;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int c = 0; c < CH; c++) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[c]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int c = 0; c < CH; c++) s += a[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// same chains with only lane 0 (or lanes 0..15) active: does the SIMD skip the passes of a wave64 instruction whose lanes are all off?
template <int CH, int ACTIVE>
__global__ void k_fma_masked(double* out, unsigned long long* cyc, int n) {
    double a[CH];
    for (int c = 0; c < CH; c++) a[c] = 1.0 + threadIdx.x * 1e-3 + c;
    const double m = 1.0000001, d = 1e-9;
    unsigned long long t0 = 0, t1 = 0;
    if (threadIdx.x < ACTIVE) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int c = 0; c < CH; c++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(d));
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    double s = 0; for (int c = 0; c < CH; c++) s += a[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename K> void run(const char* name, K k, int ch) {
    double* out; unsigned long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int n = 2000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, n); hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, n); hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s chains=%d  %.2f s_memtime ticks per instruction\n", name, ch, (double)c / ((double)n * 8 * ch));
    hipFree(out); hipFree(cyc);
}
int main() {
    run("v_fma_f64", k_fma<1>, 1); run("v_fma_f64", k_fma<2>, 2); run("v_fma_f64", k_fma<4>, 4); run("v_fma_f64", k_fma<8>, 8);
    run("v_rcp_f64", k_rcp<1>, 1); run("v_rcp_f64", k_rcp<4>, 4);
    run("fma 1 lane", k_fma_masked<8, 1>, 8); run("fma 16 lanes", k_fma_masked<8, 16>, 8); run("fma 32 lanes", k_fma_masked<8, 32>, 8);
    return 0;
}
