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


def band_rows(height, count):
    """Rows of one contiguous band when the frame is cut into `count` bands (the last one may be shorter)."""
    return (height + count - 1) // count


def band_range(height, rank, count):
    b = band_rows(height, count)
    return min(rank * b, height), min((rank + 1) * b, height)


def strips_per_rank(height, count):
    """Largest number of 16-row strips one rank owns."""
    full = (height + STRIP - 1) // STRIP
    return (full + count - 1) // count


class FrameGatherer:
    """Pipelined gather of the composited framebuffer: rank r fills `local(slot)` with its packed strips, `submit(slot)` starts ONE
    asynchronous collective (dist.gather to rank 0 == RCCL send/recv over xGMI) and, on rank 0, one strided copy that de-interleaves
    the strips into the frame; `wait(slot)` orders the caller's stream behind both before the slot is reused.  With two slots the
    gather of frame k overlaps the rendering of frame k+1.

    Layout: every rank's buffer is [K][16 rows][W][4] with K = strips_per_rank (unused strips stay empty), so the bucket on rank 0 is
    [N][K][S] and the frame, padded to N*K strips, is bucket.transpose(0, 1): strip k of rank r is frame strip k*N + r.
    `stream` (a torch.cuda.Stream, e.g. the renderer's stream wrapped in torch.cuda.ExternalStream) is the stream the local buffer
    is produced on; None = the current stream / CPU tensors (gloo rehearsal)."""

    def __init__(self, height, width, rank, count, device, group=None, slots=2, stream=None, bands=False):
        import torch
        self.height, self.width, self.rank, self.count, self.group, self.stream = height, width, rank, count, group, stream
        self.bands = bands                          # contiguous bands of band_rows(height, count) rows instead of interleaved strips
        self.k = strips_per_rank(height, count)
        self.strip_elems = STRIP * width * 4
        n = self.k * self.strip_elems
        if bands:
            n = band_rows(height, count) * width * 4
        self.locals = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(slots)]
        self.work = [None] * slots
        self.frames = [None] * slots
        if rank == 0:
            self.buckets = [torch.empty((count, n), dtype=torch.uint8, device=device) for _ in range(slots)]
            self.padded = [None if bands else torch.empty((self.k * count, self.strip_elems), dtype=torch.uint8, device=device) for _ in range(slots)]
        self.side = torch.cuda.Stream(device=device) if (stream is not None) else None      # assembly runs beside the renderer's stream

    def local(self, slot):
        return self.locals[slot]

    def owned_bytes(self):
        if self.bands:
            a, b = band_range(self.height, self.rank, self.count)
            return (b - a) * self.width * 4
        return owned_rows(self.height, self.rank, self.count) * self.width * 4

    def submit(self, slot):
        import torch
        import torch.distributed as dist
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _Null()
        with ctx:                                   # the collective is ordered behind whatever produced locals[slot] on this stream
            if self.rank == 0:
                self.work[slot] = dist.gather(self.locals[slot], list(self.buckets[slot].unbind(0)), dst=0, group=self.group, async_op=True)
            else:
                self.work[slot] = dist.gather(self.locals[slot], None, dst=0, group=self.group, async_op=True)
        if self.rank == 0:
            ctx = torch.cuda.stream(self.side) if self.side is not None else _Null()
            with ctx:
                self.work[slot].wait()              # stream-level wait on CUDA tensors, blocking on CPU tensors
                if self.bands:                      # rank r's band is rows [r*B, (r+1)*B): the bucket, flattened, IS the frame
                    self.frames[slot] = self.buckets[slot].view(-1)[:self.height * self.width * 4].view(self.height, self.width, 4)
                else:
                    self.padded[slot].view(self.k, self.count, self.strip_elems).copy_(self.buckets[slot].view(self.count, self.k, self.strip_elems).transpose(0, 1))
                    self.frames[slot] = self.padded[slot].view(-1)[:self.height * self.width * 4].view(self.height, self.width, 4)

    def wait(self, slot):
        """Before refilling locals[slot]: the collective that reads it (and rank 0's assembly of it) must have run."""
        import torch
        w = self.work[slot]
        if w is None:
            return
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _Null()
        with ctx:
            w.wait()
            if self.side is not None:
                torch.cuda.current_stream().wait_stream(self.side)
        self.work[slot] = None

    def frame(self, slot):
        """Rank 0: the assembled frame of the last submit(slot) (valid after wait(slot) / a device synchronise)."""
        return self.frames[slot]


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
