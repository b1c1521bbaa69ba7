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


def test_layout_exports_for_cost_balanced_bands_take_the_boundaries():
    """bands = 2: the boundaries come from a frame, so the (height, count, bands) forms refuse it (-1) and the *Of forms answer from
    starts[] -- the table RT64_GetGatherBands / RT64_BalanceGatherBands produce.  Pure host functions, no GPU."""
    import ctypes as C
    from sm64rt_legacy_renderer_amd import rt64
    lib = rt64.Library()
    p = C.c_int()
    assert lib.GatherRowOwner(1080, 4, 2, 10, C.byref(p)) == -1 and lib.GatherOwnedRows(1080, 4, 2, 0) == -1 and lib.GatherSlotRows(1080, 4, 2) == -1
    rng = np.random.default_rng(5)
    for h, w, n in ((270, 480, 3), (1080, 1920, 8), (64, 64, 4), (2160, 3840, 8)):
        hits = np.zeros(h, dtype=np.uint32)
        hits[h // 2:] = rng.integers(w // 2, w, size=h - h // 2)            # sky above, geometry below
        starts = (C.c_int * (n + 1))()
        lib.BalanceGatherBands(hits.ctypes.data_as(C.POINTER(C.c_uint)), w, h, n, starts)
        st = list(starts)
        assert st[0] == 0 and st[n] == h and all(b > a for a, b in zip(st, st[1:]))
        cost = (w + 6.0 * hits.astype(np.float64))
        per = [cost[a:b].sum() for a, b in zip(st, st[1:])]
        assert max(per) <= 1.35 * cost.sum() / n or max(b - a for a, b in zip(st, st[1:])) <= 16       # balanced unless the 16-row minimum binds
        if h >= 32 * n:
            assert st != [min(r * ((h + n - 1) // n), h) for r in range(n + 1)]                          # ... and not the equal-height cut
        assert lib.GatherSlotRowsOf(h, n, starts) == max(b - a for a, b in zip(st, st[1:]))
        for r in range(n):
            assert lib.GatherOwnedRowsOf(h, n, starts, r) == st[r + 1] - st[r]
        for y in range(h):
            r = lib.GatherRowOwnerOf(h, n, starts, y, C.byref(p))
            assert st[r] <= y < st[r + 1] and p.value == y - st[r], (h, n, y, r)
        assert lib.GatherRowOwnerOf(h, n, starts, h, C.byref(p)) == -1
        bad = (C.c_int * (n + 1))(*([0] + [h] * n)); bad[n] = h - 1
        assert lib.GatherRowOwnerOf(h, n, bad, 0, C.byref(p)) == -1                                      # starts[count] must be the height


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


def _rebalance_worker(rank, world, init_file, out_file):
    """bench.py's rebalancing round between two processes (control plane only, gloo): each rank "measures" the cost of its own band of a profile neither knows as a
    whole, the figures are all-gathered, RT64_RebalanceGatherBands runs on every rank: both arrive at the same boundaries, and these are more even."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from sm64rt_legacy_renderer_amd import rt64
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    lib = rt64.Library()
    H = 1080
    true = np.full(H, 0.2 / 600.0); true[600:] = 0.9 / 480.0; true[800:900] *= 2.5          # ms per row
    starts = (C.c_int * (world + 1))(0, 760, H)                                             # some first cut
    history = []
    for _ in range(3):
        a, b = starts[rank], starts[rank + 1]
        mine = torch.tensor([0.1 + float(true[a:b].sum())], dtype=torch.float32)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ms = (C.c_float * world)(*[float(t.item()) for t in every])
        nxt = (C.c_int * (world + 1))()
        assert lib.RebalanceGatherBands(H, world, starts, ms, nxt) == 1
        history.append((max(ms) / (sum(ms) / world), list(nxt)))
        starts = nxt
    mineT = torch.tensor(list(starts), dtype=torch.int32)
    both = [torch.zeros_like(mineT) for _ in range(world)]
    dist.all_gather(both, mineT)
    assert all(torch.equal(both[0], t) for t in both)               # the same boundaries everywhere
    assert history[-1][0] < 1.03 < history[0][0] and history[1][0] < history[0][0]      # an even split where the first cut was 24 % off, without overshooting on the way
    dist.barrier()
    if rank == 0:
        open(out_file, "w").write("ok")
    dist.destroy_process_group()


def test_band_rebalance_round_gloo_world2():
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "init"); out_file = os.path.join(d, "out")
        mp.spawn(_rebalance_worker, args=(2, init_file, out_file), nprocs=2, join=True)
        assert open(out_file).read() == "ok"


def test_measured_cost_feedback_levels_the_bands():
    """RT64_RebalanceGatherBands (the feedback step behind RT64_SetGatherBands): bands cut by the model, costed by a "true" profile the model does not know -- rows
    over the sphere cost twice the floor's, every band carries a fixed 0.2 ms of launches and tails -- and re-cut from the measured figures: after three rounds
    the slowest band is within 6 % of the mean (it starts 45 % above), boundaries stay ordered, 16 rows minimum, a fixed point once the costs are equal,
    invalid boundaries are refused."""
    import ctypes as C
    from sm64rt_legacy_renderer_amd import rt64
    lib = rt64.Library()
    H, W, N = 2160, 3840, 8
    hits = np.zeros(H, dtype=np.uint32); hits[980:] = W
    true = np.full(H, 0.15 / 980.0); true[980:] = 2.4 / 1180.0; true[1600:1880] *= 2.0          # ms per row: sky, floor, the rows under the sphere
    starts = (C.c_int * (N + 1))()
    lib.BalanceGatherBands(hits.ctypes.data_as(C.POINTER(C.c_uint)), W, H, N, starts)

    def measure(st):
        return np.array([0.2 + true[a:b].sum() for a, b in zip(st, st[1:])], dtype=np.float32)
    first = measure(list(starts))
    assert first.max() > 1.25 * first.mean()
    for _ in range(3):
        ms = measure(list(starts))
        out = (C.c_int * (N + 1))()
        assert lib.RebalanceGatherBands(H, N, starts, ms.ctypes.data_as(C.POINTER(C.c_float)), out) == 1
        s = list(out)
        assert s[0] == 0 and s[-1] == H and all(b - a >= 16 for a, b in zip(s, s[1:]))
        starts = out
    last = measure(list(starts))
    assert last.max() < 1.06 * last.mean() and last.max() < 0.75 * first.max()
    equal = np.full(N, 0.5, dtype=np.float32)
    out = (C.c_int * (N + 1))()
    assert lib.RebalanceGatherBands(H, N, starts, equal.ctypes.data_as(C.POINTER(C.c_float)), out) == 1 and list(out) == list(starts)
    bad = (C.c_int * (N + 1))(*([0] * N + [H - 1]))
    assert lib.RebalanceGatherBands(H, N, bad, equal.ctypes.data_as(C.POINTER(C.c_float)), out) == 0


def _halo_regions(lib, h, starts, rank, halo):
    import ctypes as C
    from sm64rt_legacy_renderer_amd import rt64
    n = len(starts) - 1
    arr = (C.c_int * (n + 1))(*starts)
    regs = (rt64.HALO_REGION * (2 * n))()
    got = lib.HaloPlan(h, n, arr, rank, halo, C.cast(regs, C.c_void_p), 2 * n)
    assert 0 <= got <= 2 * n
    assert lib.HaloPlan(h, n, arr, rank, halo, None, 0) == got                     # capacity 0: the count alone
    return [(g.peer, g.send, g.y0, g.y1) for g in regs[:got]]


def test_halo_plan_pairs_every_receive_with_a_send_and_covers_the_halo():
    """RT64_HaloPlan (the schedule of the SVGF halo exchange, pure host function): what rank r expects from q is what q sends to r; a rank's receives
    are exactly the rows within `halo` of its band that other bands own; its sends lie inside its own band.  Equal bands, cost-balanced (ragged) bands,
    bands thinner than the halo (rows from several neighbours), empty bands, a world of one."""
    from sm64rt_legacy_renderer_amd import rt64
    lib = rt64.Library()
    cases = [(180, [0, 90, 180], 62), (180, [0, 70, 131, 180], 62), (2160, [0, 300, 700, 1100, 1259, 1418, 1577, 1800, 2160], 62), (100, [0, 20, 40, 60, 80, 100], 62),
             (64, [0, 16, 16, 40, 64], 5), (50, [0, 50], 62), (1080, [0, 540, 1080], 0)]
    for h, starts, halo in cases:
        n = len(starts) - 1
        plans = [_halo_regions(lib, h, starts, r, halo) for r in range(n)]
        for r, plan in enumerate(plans):
            a, b = starts[r], starts[r + 1]
            need = np.zeros(h, dtype=np.int32)
            for peer, send, y0, y1 in plan:
                assert peer != r and 0 <= y0 < y1 <= h
                if send:
                    assert a <= y0 and y1 <= b                                        # rows of my own band
                    assert (r, 0, y0, y1) in plans[peer]                              # ... that the peer expects from me
                else:
                    assert starts[peer] <= y0 and y1 <= starts[peer + 1]              # rows of the peer's band
                    assert (r, 1, y0, y1) in plans[peer]                              # ... that the peer sends me
                    need[y0:y1] += 1
            want = np.zeros(h, dtype=np.int32)
            if b > a:
                want[max(0, a - halo):a] = 1; want[b:min(h, b + halo)] = 1
            assert np.array_equal(need, want), (h, starts, r)
    import ctypes as C
    bad = (C.c_int * 3)(0, 90, 179)
    assert lib.HaloPlan(180, 2, bad, 0, 62, None, 0) == -1


def _halo_worker(rank, world, init_file, h, w, starts, out_file):
    """The schedule carried out over gloo: the callback a host hands to RT64_SetDeviceHaloExchange, here fed with synthetic rows."""
    import ctypes as C
    import torch.distributed as dist
    from sm64rt_legacy_renderer_amd import rt64
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    lib = rt64.Library()
    arr = (C.c_int * (world + 1))(*starts)
    regs = (rt64.HALO_REGION * (2 * world))()
    n = lib.HaloPlan(h, world, arr, rank, rt64.HALO_ROWS, C.cast(regs, C.c_void_p), 2 * world)
    row_bytes = w * rt64.HALO_BYTES_PER_PIXEL

    def row_pattern(owner, y):                      # the bytes rank `owner` holds for row y
        return ((np.arange(row_bytes, dtype=np.int64) * 7 + y * 13 + owner * 101) % 251).astype(np.uint8)
    bufs = []
    for k in range(n):
        g = regs[k]
        buf = np.zeros((g.y1 - g.y0) * row_bytes, dtype=np.uint8)
        if g.send:
            buf[:] = np.concatenate([row_pattern(rank, y) for y in range(g.y0, g.y1)])
        g.host = buf.ctypes.data; g.bytes = buf.nbytes
        bufs.append(buf)
    tiles.halo_exchange_gloo(regs, n)
    for k in range(n):
        g = regs[k]
        if not g.send:
            assert np.array_equal(bufs[k], np.concatenate([row_pattern(g.peer, y) for y in range(g.y0, g.y1)])), (rank, g.peer, g.y0, g.y1)
    dist.barrier()
    if rank == 0:
        open(out_file, "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,starts", [(2, 180, [0, 90, 180]), (3, 180, [0, 70, 131, 180])])
def test_halo_exchange_schedule_over_gloo(world, h, starts):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "init"); out_file = os.path.join(d, "out")
        mp.spawn(_halo_worker, args=(world, init_file, h, 40, starts, out_file), nprocs=world, join=True)
        assert open(out_file).read() == "ok"


def test_direct_gather_flow_control_never_overwrites_a_frame_rank0_still_holds():
    """The slot discipline of the direct gather (RT64_SetGatherDirect; Gather::Direct in rt64_host.cpp) as an event simulation: `ranks` renderers of random, changing speeds
    store frame j into slot j mod 6 of rank 0's memory as soon as their own token of frame j - 3 has been taken; rank 0 takes the tokens of frame m in one group, after its
    own frame m and in frame order.  Asserted at every store: the slot's previous tenant (frame j - 6) was complete on rank 0 -- every rank's token taken -- and is not one of
    the three most recent frames rank 0 has gathered (the window the interface promises to leave intact)."""
    import random
    SLOTS, LAG = 6, 3
    for seed in range(20):
        rng = random.Random(seed)
        ranks, frames = rng.choice([2, 3, 8]), 60
        speed = [rng.uniform(0.2, 3.0) for _ in range(ranks)]
        rendered = [[None] * frames for _ in range(ranks)]      # time rank r finished storing frame j
        started = [[None] * frames for _ in range(ranks)]
        taken = [None] * frames                                  # time rank 0's group of frame m completed (every token taken)
        clock = [0.0] * ranks
        for j in range(frames):
            for r in range(ranks):
                t = clock[r]
                if j - LAG >= 0:
                    t = max(t, taken[j - LAG])                   # this rank's token of frame j - LAG has been taken (the group completes for every rank together)
                started[r][j] = t
                if rng.random() < 0.1:
                    speed[r] = rng.uniform(0.2, 3.0)
                rendered[r][j] = t + speed[r] * rng.uniform(0.8, 1.2)
                clock[r] = rendered[r][j]
            prev = taken[j - 1] if j else 0.0
            taken[j] = max(max(rendered[r][j] for r in range(ranks)), prev) + 0.05      # rank 0's groups run in order, each after every rank's frame
        for j in range(SLOTS, frames):
            first_store = min(started[r][j] for r in range(ranks))
            assert taken[j - SLOTS] <= first_store, (seed, j)                            # the previous tenant was complete before anyone stores over it
            newest = max((m for m in range(frames) if taken[m] <= first_store), default=-1)      # the newest frame rank 0 has gathered when the first store lands
            assert j - SLOTS <= newest - 3 or newest < 3, (seed, j, newest)              # ... and it is older than the three most recent gathered frames
