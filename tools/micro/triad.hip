// HBM bandwidth calibration: stream triad a[i] = b[i] + s * c[i] and a device-to-device copy on 1 GiB arrays.
// hipcc --offload-arch=gfx950 -O3 triad.hip -o triad ; prints GB/s (bytes moved / time), best of 20.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void triad(float4 *__restrict__ a, const float4 *__restrict__ b, const float4 *__restrict__ c, float s, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(float4);
    float4 *a, *b, *c;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 1, bytes); hipMemset(c, 2, bytes);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, %.0f MiB L2, memory clock %d kHz, bus %d bit\n", p.name, p.multiProcessorCount, p.l2CacheSize / 1048576.0, p.memoryClockRate, p.memoryBusWidth);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, ms;
    for (int it = 0; it < 20; it++) {
        hipEventRecord(e0); hipLaunchKernelGGL(triad, dim3(256 * 16), dim3(256), 0, 0, a, b, c, 3.0f, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("triad  : %.0f GB/s (3 x 1 GiB in %.3f ms)\n", 3.0 * bytes / (best * 1e-3) / 1e9, best);
    best = 1e9f;
    for (int it = 0; it < 20; it++) {
        hipEventRecord(e0); hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("d2d copy: %.0f GB/s (2 x 1 GiB in %.3f ms)\n", 2.0 * bytes / (best * 1e-3) / 1e9, best);
    return 0;
}
