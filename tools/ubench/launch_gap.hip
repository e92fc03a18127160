// Dependent-launch cost on one stream: a chain of N short kernels launched one by one vs. the same chain captured into a hipGraph.
// Build: hipcc --offload-arch=gfx950 -O2 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int n = 1 << 20, N = 40, reps = 20;
    float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto chain = [&]() { for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, st, p, n); };
    chain(); CK(hipStreamSynchronize(st));
    // single kernel time
    CK(hipEventRecord(a, st)); hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, st, p, n); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float one = 0; CK(hipEventElapsedTime(&one, a, b));
    CK(hipEventRecord(a, st)); for (int r = 0; r < reps; r++) chain(); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    printf("stream launches: %.2f us per kernel (one kernel alone, event to event: %.2f us)\n", ms * 1e3 / (reps * N), one * 1e3);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); chain(); CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st)); for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("graph launches:  %.2f us per kernel\n", ms * 1e3 / (reps * N));
    return 0;
}
