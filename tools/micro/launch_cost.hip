// Host-side cost of hipLaunchKernelGGL by kernel-argument size / LDS / grid (diagnosis helper): hipcc --offload-arch=gfx950 -O2 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Small { int a; };
struct Big { float v[360]; };       // 1440 bytes, like FrameParams + ViewImages
__global__ void k_small(Small s, int *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && s.a == -1) *out = 1; }
__global__ void k_big(Big b, int *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && b.v[7] == -1.0f) *out = 1; }
__global__ void k_big_lds(Big b, int *out) { __shared__ int lds[46336 / 4]; lds[threadIdx.x] = (int)b.v[3]; __syncthreads(); if (lds[0] == -1) *out = 1; }
template <class F> static void timeit(const char *name, hipStream_t s, F f) {
    for (int i = 0; i < 50; i++) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; i++) f();
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    auto t2 = std::chrono::steady_clock::now();
    printf("%-34s host %.2f us/launch, drained %.2f us/launch\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000, std::chrono::duration<double, std::micro>(t2 - t0).count() / 2000);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *out; hipMalloc(&out, 4);
    Small sm = { 1 }; Big bg = {};
    timeit("small args, 1 block", s, [&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, sm, out); });
    timeit("small args, 2048 blocks", s, [&] { hipLaunchKernelGGL(k_small, dim3(2048), dim3(256), 0, s, sm, out); });
    timeit("1440-B args, 1 block", s, [&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, bg, out); });
    timeit("1440-B args, 2048 blocks", s, [&] { hipLaunchKernelGGL(k_big, dim3(2048), dim3(256), 0, s, bg, out); });
    timeit("1440-B args + 46 KB LDS, 2048", s, [&] { hipLaunchKernelGGL(k_big_lds, dim3(2048), dim3(256), 0, s, bg, out); });
    hipEvent_t e; hipEventCreate(&e);
    timeit("hipEventRecord", s, [&] { hipEventRecord(e, s); });
    return 0;
}
