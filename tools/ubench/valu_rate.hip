// VALU issue-rate micro-benchmark (dev aid): wave64 instructions per cycle per SIMD for a few integer / float / packed ops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 256
template <int OP> __global__ __launch_bounds__(256) void k(int* out, int n_iter, int a0, int b0) {
    int a[8], b = b0 + threadIdx.x;
    float fa[8], fb = (float)b; double da[8], db = (double)b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = a0 + i + threadIdx.x; fa[i] = (float)a[i]; da[i] = fa[i]; }
    for (int it = 0; it < n_iter; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
                if (OP == 2) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fb));
                if (OP == 4) asm volatile("v_dot4_i32_i8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 5) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 6) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 8) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 9) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(fa[i]) : "v"(fb));
                if (OP == 10) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "+v"(fa[i]) : "v"(b));
                if (OP == 11) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 12) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 13) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 14) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&da[i]) : "v"(*(double*)&db));
            }
        }
    }
    int s = 0; float fs = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { s += a[i]; fs += fa[i] + (float)da[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (int)fs;
}
template <int OP> void run(const char* name, int* d) {
    const int blocks = 256 * 8, iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2, 1, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1, 2);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * REP;           // wave-instructions of the op under test
    printf("%-28s %8.3f ms  %7.1f G wave-instr/s  = %.3f per SIMD-cycle at 2.4 GHz (1024 SIMDs)\n", name, ms, winstr / ms / 1e6, winstr / (ms * 1e-3) / (1024 * 2.4e9));
}
int main() {
    int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_min_i32", d); run<1>("v_min_f32", d); run<2>("v_sub_u32", d); run<3>("v_sub_f32", d);
    run<4>("v_dot4_i32_i8", d); run<5>("v_pk_min_i16", d); run<6>("v_max3_i32", d); run<7>("v_mul_lo_u32", d);
    run<8>("v_pk_sub_i16", d); run<9>("v_max3_f32", d); run<10>("v_cvt_f32_ubyte1", d); run<11>("v_mad_u32_u24", d);
    run<12>("v_and_b32", d); run<13>("v_lshl_add_u32", d); run<14>("v_pk_fma_f32", d);
    return 0;
}
