"""Multi-rank path on CPU: strip partition + gather of the composited framebuffer with torch.distributed (gloo, world_size 2)."""
import os
import sys
import tempfile

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # spawned workers import this module without conftest
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from sm64rt_legacy_renderer_amd import tiles  # noqa: E402


def test_strip_partition_covers_every_row_once():
    for h in (1, 15, 16, 17, 270, 1080, 2160):
        for n in (1, 2, 3, 4, 8):
            rows = np.zeros(h, dtype=np.int32)
            for r in range(n):
                for a, b in tiles.strip_ranges(h, r, n):
                    assert a % 16 == 0 and b - a <= 16
                    rows[a:b] += 1
                assert tiles.owned_rows(h, r, n) == sum(b - a for a, b in tiles.strip_ranges(h, r, n))
            assert (rows == 1).all()
            assert tiles.max_owned_rows(h, n) >= -(-h // n) - 16


def test_assemble_numpy_roundtrip():
    rng = np.random.default_rng(5)
    for h, w, n in ((1080, 64, 8), (270, 48, 3), (33, 16, 2)):
        frame = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        mx = tiles.max_owned_rows(h, n) * w * 4
        packed = np.zeros((n, mx), dtype=np.uint8)
        for r in range(n):
            flat = np.concatenate([frame[a:b].reshape(-1) for a, b in tiles.strip_ranges(h, r, n)]) if tiles.strip_ranges(h, r, n) else np.zeros(0, np.uint8)
            packed[r, :flat.size] = flat
        assert np.array_equal(tiles.assemble(packed, h, w, n), frame)
        import torch
        assert np.array_equal(tiles.assemble_fast(torch.from_numpy(packed), h, w, n).numpy(), frame)


def test_library_layout_exports_agree_with_the_python_partition():
    """RT64_GatherRowOwner / RT64_GatherOwnedRows / RT64_GatherSlotRows (the one layout definition behind RT64_CreateGather, shared with the
    device-side reassembly kernel) against tiles.strip_ranges / band_range; pure host functions, no GPU."""
    import ctypes as C
    from sm64rt_legacy_renderer_amd import rt64
    lib = rt64.Library()
    p = C.c_int()
    for h in (1, 15, 16, 17, 40, 270, 1080):
        for n in (1, 2, 3, 8):
            for bands in (0, 1):
                owner = np.full(h, -1); packed_row = np.full(h, -1)
                for r in range(n):
                    ranges = [tiles.band_range(h, r, n)] if bands else tiles.strip_ranges(h, r, n)
                    ranges = [(a, b) for a, b in ranges if b > a]
                    k = 0
                    for a, b in ranges:
                        for y in range(a, b):
                            owner[y] = r; packed_row[y] = k; k += 1
                    assert lib.GatherOwnedRows(h, n, bands, r) == k
                    assert lib.GatherSlotRows(h, n, bands) >= k
                for y in range(h):
                    assert lib.GatherRowOwner(h, n, bands, y, C.byref(p)) == owner[y] and p.value == packed_row[y], (h, n, bands, y)


def _worker(rank, world, init_file, h, w, out_file):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    rng = np.random.default_rng(11)
    frame = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)            # every rank knows the expected frame
    mx = tiles.max_owned_rows(h, world) * w * 4
    local = torch.zeros(mx, dtype=torch.uint8)
    mine = np.concatenate([frame[a:b].reshape(-1) for a, b in tiles.strip_ranges(h, rank, world)])
    local[:mine.size] = torch.from_numpy(mine)
    for fast in (True, False):
        full = tiles.gather_frame(local, h, w, rank, world, fast=fast)
        if rank == 0:
            assert np.array_equal(full.numpy(), frame)
        else:
            assert full is None
    # the same bucket reassembled by the layout the in-library RCCL gather uses on rank 0 (RT64_GatherRowOwner)
    from sm64rt_legacy_renderer_amd import rt64
    lib = rt64.Library()
    assert lib.GatherSlotRows(h, world, 0) * w * 4 >= mx and lib.GatherOwnedRows(h, world, 0, rank) * w * 4 == mine.size
    if rank == 0:
        bucket = torch.empty((world, mx), dtype=torch.uint8)
        dist.gather(local, list(bucket.unbind(0)), dst=0)
        assert np.array_equal(tiles.assemble_by_library_layout(lib, bucket.numpy(), h, w, world), frame)
    else:
        dist.gather(local, None, dst=0)
    # pipelined gatherer (bench.py's N > 1 path): two slots, frames that differ per step, submit / wait in flight order
    g = tiles.FrameGatherer(h, w, rank, world, "cpu")
    assert g.local(0).numel() == tiles.strips_per_rank(h, world) * 16 * w * 4 >= mine.size == g.owned_bytes()
    for step in range(5):
        slot = step % 2
        g.wait(slot)
        shifted = np.roll(frame, step, axis=1)
        part = np.concatenate([shifted[a:b].reshape(-1) for a, b in tiles.strip_ranges(h, rank, world)])
        g.local(slot)[:part.size] = torch.from_numpy(part)
        g.submit(slot)
        if rank == 0:
            g.wait(slot)                      # CPU tensors: the collective has completed here
            assert np.array_equal(g.frame(slot).numpy(), shifted)
    # contiguous bands (frames with GI + denoiser): rank r owns rows [r*B, (r+1)*B), the gathered bucket is the frame itself
    gb = tiles.FrameGatherer(h, w, rank, world, "cpu", bands=True)
    a, b = tiles.band_range(h, rank, world)
    assert gb.owned_bytes() == (b - a) * w * 4 and gb.local(0).numel() == tiles.band_rows(h, world) * w * 4
    for step in range(3):
        slot = step % 2
        gb.wait(slot)
        shifted = np.roll(frame, step, axis=0)
        gb.local(slot)[:(b - a) * w * 4] = torch.from_numpy(np.ascontiguousarray(shifted[a:b]).reshape(-1))
        gb.submit(slot)
        if rank == 0:
            gb.wait(slot)
            assert np.array_equal(gb.frame(slot).numpy(), shifted)
    dist.barrier()
    if rank == 0:
        open(out_file, "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("h,w", [(270, 32), (1080, 8), (40, 8)])
def test_gather_frame_gloo_world2(h, w):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "init"); out_file = os.path.join(d, "out")
        mp.spawn(_worker, args=(2, init_file, h, w, out_file), nprocs=2, join=True)
        assert open(out_file).read() == "ok"


def test_cost_balanced_bands_are_balanced_deterministic_and_cover_the_frame():
    """RT64_BalanceGatherBands (the cut behind RT64_CreateGather(bands = 2)): contiguous bands whose modelled cost -- a row's pixels, the ones
    that hit geometry weighted 6x -- is about equal, on a frame whose top half is sky (the sample scene's shape)."""
    import ctypes as C
    from sm64rt_legacy_renderer_amd import rt64
    lib = rt64.Library()
    H, W = 2160, 3840
    hits = np.zeros(H, dtype=np.uint32)
    hits[980:] = (W * np.clip(np.linspace(0.3, 1.0, H - 980), 0, 1)).astype(np.uint32)          # geometry only below the horizon
    cost = W + 6.0 * hits.astype(np.float64)
    for n in (1, 2, 3, 8):
        starts = (C.c_int * (n + 1))()
        lib.BalanceGatherBands(hits.ctypes.data_as(C.POINTER(C.c_uint)), W, H, n, starts)
        s = list(starts)
        assert s[0] == 0 and s[-1] == H and all(b - a >= 16 for a, b in zip(s, s[1:]))
        band_cost = np.array([cost[a:b].sum() for a, b in zip(s, s[1:])])
        assert band_cost.max() <= 1.05 * cost.sum() / n + cost.max()                           # within a row of the ideal share
        equal = np.array([cost[a:b].sum() for a, b in (tiles.band_range(H, r, n) for r in range(n))])
        if n >= 2:
            assert band_cost.max() < 0.75 * equal.max()                                         # equal heights leave the sky bands idle
        again = (C.c_int * (n + 1))()
        lib.BalanceGatherBands(hits.ctypes.data_as(C.POINTER(C.c_uint)), W, H, n, again)
        assert list(again) == s
    # degenerate inputs: nothing hit (equal heights), fewer rows than 16 per band
    starts = (C.c_int * 9)()
    zero = np.zeros(100, dtype=np.uint32)
    lib.BalanceGatherBands(zero.ctypes.data_as(C.POINTER(C.c_uint)), 64, 100, 8, starts)
    s = list(starts)
    assert s[0] == 0 and s[-1] == 100 and all(b > a for a, b in zip(s, s[1:]))
