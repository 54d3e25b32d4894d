// What does FETCH_SIZE report for the access patterns of megastep_hip.hip?  Four kernels with KNOWN useful byte counts over a
// 1 GiB array (four times the Infinity Cache, so nothing is served twice): a 16 B/lane coalesced stream (the pattern the
// guide's "x2" was calibrated on) and random gathers of 4 B, 12 B (three dwords, as texels are read) and 16 B rows - one
// element per lane, N lanes.  tools/calib/fetch_calibration.sh runs it under rocprofv3 (FETCH_SIZE in one pass, the L2's
// fabric read requests by size in another) and prints counter bytes / useful bytes per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void stream16(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    if (i < n) { const float4 v = a[i]; if (v.x == 12345.f) out[0] = v.y; }
}
__global__ void gather4(const float* __restrict__ a, float* __restrict__ out, size_t n, uint32_t rows) {
    const size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    if (i < n) { const float v = a[mix((uint32_t)i) % rows]; if (v == 12345.f) out[0] = v; }
}
__global__ void gather12(const float* __restrict__ a, float* __restrict__ out, size_t n, uint32_t rows) {
    const size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    if (i < n) { const float* p = a + 3*(size_t)(mix((uint32_t)i) % rows); const float v = p[0] + p[1] + p[2]; if (v == 12345.f) out[0] = v; }
}
__global__ void gather16(const float4* __restrict__ a, float* __restrict__ out, size_t n, uint32_t rows) {
    const size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    if (i < n) { const float4 v = a[mix((uint32_t)i) % rows]; if (v.x == 12345.f) out[0] = v.y; }
}

int main() {
    const size_t bytes = 1ull << 30, lanes = 1ull << 24;           // 1 GiB; 16 M lanes per kernel
    float* a; float* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess || hipMemset(a, 0, bytes) != hipSuccess) return 1;
    const dim3 block(256), grid((unsigned)(lanes/256));
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const float4*)a, out, lanes);                       // 256 MiB useful
        hipLaunchKernelGGL(gather4, grid, block, 0, 0, a, out, lanes, (uint32_t)(bytes/4));                  // 64 MiB useful
        hipLaunchKernelGGL(gather12, grid, block, 0, 0, a, out, lanes, (uint32_t)(bytes/12));                // 192 MiB useful
        hipLaunchKernelGGL(gather16, grid, block, 0, 0, (const float4*)a, out, lanes, (uint32_t)(bytes/16)); // 256 MiB useful
    }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    printf("useful bytes per launch: stream16 %zu gather4 %zu gather12 %zu gather16 %zu\n", lanes*16, lanes*4, lanes*12, lanes*16);
    return 0;
}
