// Probe (dev aid): operand / result lane layout of v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* A /*16x4*/, const double* B /*4x16*/, double* out /*64 lanes x 4*/) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];       // A[i][k], i = l % 16, k = l / 16
    const double b = B[(l / 16) * 16 + l % 16];      // B[k][j], k = l / 16, j = l % 16
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
int main() {
    double A[64], B[64], D[256], out[256];
    for (int i = 0; i < 64; i++) { A[i] = 1 + i * 0.37; B[i] = 2 - i * 0.11; }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j]; D[i * 16 + j] = s; }
    double *dA, *dB, *dO; (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dO, 2048);
    (void)hipMemcpy(dA, A, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    (void)hipMemcpy(out, dO, 2048, hipMemcpyDeviceToHost);
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        const double v = out[l * 4 + r];
        if (fabs(v - D[(4 * (l / 16) + r) * 16 + l % 16]) > 1e-9) okA = 0;      // layout A: row = 4*(lane/16) + r, col = lane % 16
        if (fabs(v - D[((l / 16) + 4 * r) * 16 + l % 16]) > 1e-9) okB = 0;      // layout B: row = lane/16 + 4*r
    }
    printf("operands A[i=l%%16][k=l/16], B[k=l/16][j=l%%16]; D layout row=4*(l/16)+r: %d   row=l/16+4*r: %d\n", okA, okB);
    return 0;
}
