// gather.hip -- device side of the multi-GPU framebuffer gather behind the C ABI (RT64_CreateGather / RT64_SubmitGather, rt64_host.cpp):
// rank 0 reassembles the frame from the packed row buffers the ranks sent.
//
// The reference is single-GPU (NodeMask 0, rt64_device.cpp:753); SURVEY 8(e) partitions the frame's rows over the GPUs of a node and
// gathers the composited back buffer with ONE RCCL exchange per frame.  Each rank's buffer holds its owned rows packed in ascending
// order (the layout RT64_CopyDeviceImage / RT64_SetDeviceGatherTarget produce); gather_row_owner() in rt64_gpu.h is the one
// definition of which rank owns a frame row and where it sits in that rank's buffer -- used here, by the host code and (through
// RT64_GatherRowOwner) by the Python harness's CPU tests.
#include "kernels.h"

namespace {

// One workgroup per frame row: 16-byte copies of the row from the owner's packed buffer (rank 0's own rows come from its send buffer).
__global__ __launch_bounds__(256) void gather_assemble_kernel(const uint8_t *__restrict__ own, const uint8_t *__restrict__ bucket, size_t slotBytes,
                                                              uint8_t *__restrict__ frame, int width, GatherLayout L) {
    const int y = blockIdx.x, height = L.height;
    if (y >= height) return;
    int packed;
    const int owner = gather_row_owner(L, y, &packed);
    const uint8_t *src = (owner == 0 ? own : bucket + (size_t)owner * slotBytes) + (size_t)packed * (size_t)width * 4;
    uint8_t *dst = frame + (size_t)y * (size_t)width * 4;
    const int bytes = width * 4;
    if ((bytes & 15) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src); uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (int i = threadIdx.x; i < bytes / 16; i += 256) d4[i] = s4[i];
    }
    else {
        const uint32_t *s1 = reinterpret_cast<const uint32_t *>(src); uint32_t *d1 = reinterpret_cast<uint32_t *>(dst);
        for (int i = threadIdx.x; i < width; i += 256) d1[i] = s1[i];
    }
}

}  // namespace

hipError_t launch_gather_assemble(const uint8_t *own, const uint8_t *bucket, size_t slotBytes, uint8_t *frame, int width, const GatherLayout &L, hipStream_t s) {
    hipLaunchKernelGGL(gather_assemble_kernel, dim3((unsigned)L.height), dim3(256), 0, s, own, bucket, slotBytes, frame, width, L);
    return hipGetLastError();
}

// Pixels per row whose primary ray hit geometry (the cost model of the cost-balanced bands): one workgroup per row.
namespace {
__global__ __launch_bounds__(256) void row_hit_count_kernel(const int32_t *__restrict__ hitInstance, uint32_t *__restrict__ counts, int width) {
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    uint32_t n = 0;
    for (int x = threadIdx.x; x < width; x += 256) n += hitInstance[(size_t)blockIdx.x * width + x] >= 0 ? 1u : 0u;
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, n);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
}  // namespace
hipError_t launch_row_hit_count(const int32_t *hitInstance, uint32_t *counts, int width, int height, hipStream_t s) {
    hipLaunchKernelGGL(row_hit_count_kernel, dim3((unsigned)height), dim3(256), 0, s, hitInstance, counts, width);
    return hipGetLastError();
}
