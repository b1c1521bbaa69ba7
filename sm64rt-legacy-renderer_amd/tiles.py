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


def assemble_by_library_layout(lib, packed, height, width, count, bands=False, channels=4):
    """Reassemble the frame the way rank 0 of the in-library gather does (RT64_CreateGather / gather_assemble_kernel): row y comes from
    row RT64_GatherRowOwner(..., y) -> (rank, packed row) of that rank's packed buffer.  `lib` is rt64.Library(); the layout exports are
    pure host functions, so this runs without a GPU.  packed: [count, >= RT64_GatherSlotRows * width * channels]."""
    import ctypes as C
    frame = np.empty((height, width, channels), dtype=packed.dtype)
    p = C.c_int()
    row = width * channels
    for y in range(height):
        r = lib.GatherRowOwner(height, count, int(bands), y, C.byref(p))
        frame[y] = np.asarray(packed[r][p.value * row:(p.value + 1) * row]).reshape(width, channels)
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
    asynchronous collective (dist.gather to rank 0 == RCCL send/recv over xGMI); `wait(slot)` -- called before the slot is refilled, two
    frames later, or before rank 0 reads `frame(slot)` -- orders the caller's stream behind the collective and, on rank 0, runs the one
    strided copy that de-interleaves the strips into the frame.  With two slots the gather of frame k overlaps the rendering of frame
    k+1 (RCCL works on its own stream; by the time the slot comes round again the collective has long finished, so the wait is free).

    Layout: every rank's buffer is [K][16 rows][W][4] with K = strips_per_rank (unused strips stay empty), so the bucket on rank 0 is
    [N][K][S] and the frame, padded to N*K strips, is bucket.transpose(0, 1): strip k of rank r is frame strip k*N + r.
    `stream` (a torch.cuda.Stream, e.g. the renderer's stream wrapped in torch.cuda.ExternalStream) is the stream the local buffer
    is produced on: it becomes this process's current torch stream, once, so that no per-frame call has to switch streams (a
    `with torch.cuda.stream(...)` block costs ~25 us of host time per frame -- a third of what a 1/8 share of the frame takes to render).
    None = the current stream / CPU tensors (gloo rehearsal)."""

    def __init__(self, height, width, rank, count, device, group=None, slots=2, stream=None, bands=False):
        import torch
        self.height, self.width, self.rank, self.count, self.group, self.stream = height, width, rank, count, group, stream
        self.bands = bands                          # contiguous bands of band_rows(height, count) rows instead of interleaved strips
        self.k = strips_per_rank(height, count)
        self.strip_elems = STRIP * width * 4
        n = self.k * self.strip_elems
        if bands:
            n = band_rows(height, count) * width * 4
        self._previous_stream = None
        if stream is not None:
            self._previous_stream = torch.cuda.current_stream()
            torch.cuda.set_stream(stream)
        self.locals = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(slots)]
        self.work = [None] * slots
        self.frames = [None] * slots
        self.outputs = [None] * slots
        if rank == 0:
            self.buckets = [torch.empty((count, n), dtype=torch.uint8, device=device) for _ in range(slots)]
            self.outputs = [list(b.unbind(0)) for b in self.buckets]            # built once: the per-frame call only passes them on
            self.padded = [None if bands else torch.empty((self.k * count, self.strip_elems), dtype=torch.uint8, device=device) for _ in range(slots)]
            for sl in range(slots):                 # the views every frame is read through
                flat = self.buckets[sl] if bands else self.padded[sl]
                self.frames[sl] = flat.view(-1)[:height * width * 4].view(height, width, 4)
            if not bands:
                self._dst = [p.view(self.k, count, self.strip_elems) for p in self.padded]
                self._src = [b.view(count, self.k, self.strip_elems).transpose(0, 1) for b in self.buckets]

    def local(self, slot):
        return self.locals[slot]

    def owned_bytes(self):
        if self.bands:
            a, b = band_range(self.height, self.rank, self.count)
            return (b - a) * self.width * 4
        return owned_rows(self.height, self.rank, self.count) * self.width * 4

    def submit(self, slot):
        """Start the collective of `slot`: ordered behind whatever produced locals[slot] on the current stream; returns at once."""
        if self._pg is None:
            self._bind()
        if self._pg is not False:       # the process group's own entry point: same collective as dist.gather without ~20 us of per-call argument checking
            self.work[slot] = self._pg.gather(self._out_arg[slot], self._in_arg[slot], self._opts)
        else:
            import torch.distributed as dist
            self.work[slot] = dist.gather(self.locals[slot], self.outputs[slot], dst=0, group=self.group, async_op=True)

    _pg = None

    def _bind(self):
        import torch.distributed as dist
        try:
            pg = self.group if self.group is not None else dist.distributed_c10d._get_default_group()
            opts = dist.GatherOptions(); opts.rootRank = 0
            self._out_arg = [[o] if self.rank == 0 else [] for o in (self.outputs if self.rank == 0 else [None] * len(self.locals))]
            self._in_arg = [[t] for t in self.locals]
            pg.gather; self._pg, self._opts = pg, opts
        except Exception:                # another torch version: the public wrapper does the same thing
            self._pg = False

    def wait(self, slot):
        """Before refilling locals[slot] / reading frame(slot): the collective that used the slot must have run; rank 0 assembles."""
        w = self.work[slot]
        if w is None:
            return
        w.wait()                                    # stream-level wait on CUDA tensors, blocking on CPU tensors
        if self.rank == 0 and not self.bands:       # rank r's band is rows [r*B, (r+1)*B): with bands the bucket, flattened, IS the frame
            self._dst[slot].copy_(self._src[slot])
        self.work[slot] = None

    def close(self):
        """Finish the outstanding collectives and hand torch's current stream back (the renderer's stream dies with its device)."""
        for slot in range(len(self.work)):
            self.wait(slot)
        if self._previous_stream is not None:
            import torch
            torch.cuda.current_stream().synchronize()
            torch.cuda.set_stream(self._previous_stream)
            self._previous_stream = None

    def frame(self, slot):
        """Rank 0: the assembled frame of the last submit(slot), valid after wait(slot) (on the current stream for CUDA tensors)."""
        return self.frames[slot]


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def halo_exchange_gloo(regions, count):
    """Transport of the SVGF halo exchange over torch.distributed (the body of a RT64_HALO_EXCHANGE callback: RT64_SetDeviceHaloExchange): every region's
    host buffer is sent to / filled from its peer.  Non-blocking sends and receives, so the order of the regions cannot deadlock; two messages between
    the same pair of ranks are matched in issue order, which the schedule keeps identical on both sides (RT64_HaloPlan lists regions by peer)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    reqs, keep = [], []
    for k in range(count):
        g = regions[k]
        if g.bytes == 0:
            continue
        t = torch.frombuffer((C.c_uint8 * g.bytes).from_address(g.host), dtype=torch.uint8)
        keep.append(t)
        reqs.append(dist.isend(t, g.peer) if g.send else dist.irecv(t, g.peer))
    for r in reqs:
        r.wait()
