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
                                                              uint8_t *__restrict__ frame, int width, int height, int count, int bands) {
    const int y = blockIdx.x;
    if (y >= height) return;
    int packed;
    const int owner = gather_row_owner(height, count, bands, y, &packed);
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

hipError_t launch_gather_assemble(const uint8_t *own, const uint8_t *bucket, size_t slotBytes, uint8_t *frame, int width, int height, int count, int bands, hipStream_t s) {
    hipLaunchKernelGGL(gather_assemble_kernel, dim3((unsigned)height), dim3(256), 0, s, own, bucket, slotBytes, frame, width, height, count, bands);
    return hipGetLastError();
}
