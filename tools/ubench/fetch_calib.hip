// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the extractor kernels use (the guide's
// "x2" rule is for 16 bytes per lane; other widths are uncalibrated): every kernel streams the same 1 GiB buffer once (well past the
// 256 MiB Infinity Cache), so the true HBM read is 1 GiB = 1048576 KiB for each of them.
//   k_read_b32   one dword per lane, consecutive lanes consecutive dwords (256 B per wave instruction)
//   k_read_b128  16 bytes per lane (1 KiB per wave instruction)
//   k_read_rows72 18 consecutive dwords per 128-byte-aligned row segment start... (a 72-byte row segment per 18 lanes, rows 768 B apart:
//                 the blur kernel's staging pattern before the XCD-aware placement; reads 72 of every 128 bytes of a 64-px tile column)
//   k_write_b32  one dword per lane streaming store
// Build: hipcc --offload-arch=gfx950 -O2 fetch_calib.hip -o fetch_calib ; run under rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_read_b32(const uint32_t* __restrict__ p, size_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_read_b128(const uint4* __restrict__ p, size_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// image-like: rows of 768 bytes; a block reads a 64-row x 72-byte tile (18 dwords per row), tiles 64 bytes apart in x: every 128-byte
// line is touched by two or three different tiles (different blocks, hence different XCDs under round-robin placement)
__global__ void k_read_rows72(const uint8_t* __restrict__ p, int rows, uint32_t* __restrict__ out) {
    const int tiles_x = 11, tile = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x);
    if (ty * 64 >= rows) return;
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < 64 * 18; i += blockDim.x) {
        const int r = i / 18, c = i - r * 18;
        acc ^= *reinterpret_cast<const uint32_t*>(p + (size_t)(ty * 64 + r) * 768 + tile * 64 + 4 * c);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_write_b32(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
int main() {
    const size_t bytes = 1ull << 30;
    uint8_t* buf; uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_read_b32, dim3(4096), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, out);
        hipLaunchKernelGGL(k_read_b128, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
        const int rows = (int)(bytes / 768);
        hipLaunchKernelGGL(k_read_rows72, dim3(11 * ((rows + 63) / 64)), dim3(256), 0, 0, buf, rows - 64, out);
        hipLaunchKernelGGL(k_write_b32, dim3(4096), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4);
    }
    (void)hipDeviceSynchronize();
    printf("done: each kernel touched %zu KiB (rows72: 11 x 72 of every 768-byte row = %zu KiB distinct bytes, all 128-byte lines)\n", bytes >> 10, (bytes / 768 * 768) >> 10);
    return 0;
}
