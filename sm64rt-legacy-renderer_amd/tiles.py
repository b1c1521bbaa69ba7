"""Image-tile partition of one frame across the GPUs of a node + gather of the composited framebuffer.

The reference is single-GPU (NodeMask 0, rt64_device.cpp:753); pixels are independent in every pass of the C2
configuration, so the frame shards naturally: the scene (BVH, textures, lights) is replicated on every rank, rank r
renders the 16-row strips r, r+N, r+2N, ... (RT64_SetDeviceInterleave: interleaving balances the sky/geometry split of
the sample scene), and ONE collective per frame -- a gather of the packed RGBA8 strips to rank 0 over RCCL/xGMI --
reassembles the back buffer.  No other data-path communication exists.
"""
import numpy as np

STRIP = 16


def strip_ranges(height, rank, count, y0=0, y1=None):
    """Row ranges [(a, b), ...] rendered by `rank` of `count` (same rule as Device::forEachOwnedStrip)."""
    y1 = height if y1 is None else y1
    out = []
    y = y0 + rank * STRIP
    while y < y1:
        out.append((y, min(y + STRIP, y1)))
        y += count * STRIP
    return out


def owned_rows(height, rank, count):
    return sum(b - a for a, b in strip_ranges(height, rank, count))


def max_owned_rows(height, count):
    return max(owned_rows(height, r, count) for r in range(count))


def assemble(packed, height, width, count, channels=4):
    """packed: [count, max_rows * width * channels] per-rank packed strips (padded).  Returns the full frame
    [height, width, channels].  Works on numpy arrays and torch tensors (uses only indexing / reshape)."""
    if isinstance(packed, np.ndarray):
        frame = np.empty((height, width, channels), dtype=packed.dtype)
    else:
        import torch
        frame = torch.empty((height, width, channels), dtype=packed.dtype, device=packed.device)
    for r in range(count):
        row = 0
        flat = packed[r]
        for a, b in strip_ranges(height, r, count):
            n = (b - a) * width * channels
            frame[a:b] = flat[row:row + n].reshape(b - a, width, channels)
            row += n
    return frame


def assemble_fast(packed, height, width, count, channels=4):
    """Same as assemble() for torch tensors: one strided copy per rank for the full strips, one for a ragged tail."""
    import torch
    frame = torch.empty((height, width, channels), dtype=packed.dtype, device=packed.device)
    strip_elems = STRIP * width * channels
    full = height // STRIP
    view = frame[:full * STRIP].view(full, strip_elems)
    for r in range(count):
        n = len(range(r, full, count))
        if n:
            view[r::count] = packed[r][:n * strip_elems].view(n, strip_elems)
    if height % STRIP:
        r = full % count
        n = len(range(r, full, count))
        rem = (height - full * STRIP) * width * channels
        frame[full * STRIP:] = packed[r][n * strip_elems:n * strip_elems + rem].view(height - full * STRIP, width, channels)
    return frame


def gather_frame(local, height, width, rank, count, group=None, fast=True):
    """Gather every rank's packed strips (1-D uint8 tensor padded to max_owned_rows) on rank 0 and reassemble.
    Returns the full frame on rank 0, None elsewhere.  One collective (dist.gather == ncclGather over send/recv)."""
    import torch
    import torch.distributed as dist
    if count == 1:
        return local[:height * width * 4].view(height, width, 4)
    if rank == 0:
        bucket = torch.empty((count, local.numel()), dtype=local.dtype, device=local.device)
        dist.gather(local, list(bucket.unbind(0)), dst=0, group=group)
        return assemble_fast(bucket, height, width, count) if fast else assemble(bucket, height, width, count)
    dist.gather(local, None, dst=0, group=group)
    return None
