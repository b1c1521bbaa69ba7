"""Halo EXCHANGE between the bands of a partitioned GI + SVGF frame ("renders its rows plus a halo ..., or exchanges halos").

One device per band, all in this process and on the one GPU of the box, each driven by its own host thread; the transport is the host callback of
RT64_SetDeviceHaloExchange (an in-process mailbox here; RCCL between processes in production, tests/test_tiles_gloo.py rehearses the schedule over gloo).
In the middle of RT64_DrawDevice every band ships the filter input (variance image + guide records) of its edge rows to its neighbours and receives
theirs, so that it renders only its own rows (+ 4) instead of its rows + 66 on each side.  The bands put together are bit-identical to the frame
one device renders alone."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 320, 180


class Mailbox:
    """Transport between the bands' callbacks: post what you send, meet, take what was sent to you, meet again."""

    def __init__(self, n):
        self.barrier = threading.Barrier(n, timeout=60)
        self.box = {}
        self.calls = [0] * n
        self.bytes_sent = [0] * n

    def callback(self, rt64, rank):
        def cb(user, regions, count):
            regs = [regions[k] for k in range(count)]
            for g in regs:
                assert g.bytes == (g.y1 - g.y0) * W * rt64.HALO_BYTES_PER_PIXEL
                if g.send:
                    self.box[(rank, g.peer, g.y0, g.y1)] = C.string_at(g.host, g.bytes)
                    self.bytes_sent[rank] += g.bytes
            self.barrier.wait()
            for g in regs:
                if not g.send:
                    data = self.box[(g.peer, rank, g.y0, g.y1)]
                    C.memmove(g.host, data, g.bytes)
            self.barrier.wait()
            self.calls[rank] += 1
        return rt64.HALO_EXCHANGE(cb)


def _render_bands(rt64_lib, sample_data, bands, frames, gi_samples, exchange, margin=None):
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    n = len(bands)
    parts = [sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0) for _ in bands]
    box = Mailbox(n)
    keep = []
    try:
        starts = (C.c_int * (n + 1))(*([b[0] for b in bands] + [H]))
        for r, (s, (a, b)) in enumerate(zip(parts, bands)):
            s.set_view_description(gi_samples=gi_samples, denoiser=True)
            assert s.option("denoiser_mode", 1) and s.option("count_traversal", 1)
            s.set_tile(a, b)
            if exchange:
                cb = box.callback(rt64, r); keep.append(cb)
                assert rt64_lib.SetDeviceHaloExchange(s.device, C.cast(cb, C.c_void_p), None, starts, r, n) == 1
                if margin:
                    assert s.option("halo_margin", margin)
        errors = []

        def run(s):
            try:
                for _ in range(frames):
                    s.draw()
            except Exception as e:      # noqa: BLE001
                errors.append(e)
                box.barrier.abort()
        threads = [threading.Thread(target=run, args=(s,)) for s in parts]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        images = {im: np.concatenate([s.readback(im) for s in parts], axis=0)
                  for im in (rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_INDIRECT_LIGHT_FILTERED)}
        return images, [s.stats() for s in parts], box
    finally:
        for s in parts:
            s.close()


@pytest.mark.parametrize("bands,gi_samples", [([(0, 90), (90, H)], 1), ([(0, 70), (70, 131), (131, H)], 2)])
def test_bands_with_halo_exchange_equal_the_whole_frame(rt64_lib, sample_data, bands, gi_samples):
    """Two bands, and three ragged ones of which the middle one (61 rows) is thinner than the halo, so that the outer bands take rows from BOTH
    other bands.  Four frames (temporal accumulation + SVGF history).  Bit-identical to the single-device frame, and a band traces its rows + 4 on
    each side instead of + 66."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    whole = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    try:
        whole.set_view_description(gi_samples=gi_samples, denoiser=True)
        assert whole.option("denoiser_mode", 1)
        for _ in range(4):
            whole.draw()
        full = {im: whole.readback(im) for im in (rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_INDIRECT_LIGHT_FILTERED)}
    finally:
        whole.close()
    images, stats, box = _render_bands(rt64_lib, sample_data, bands, 4, gi_samples, exchange=True)
    for im, ref in full.items():
        assert images[im].shape == ref.shape and np.array_equal(images[im], ref), im
    assert box.calls == [4] * len(bands)
    for (a, b), st in zip(bands, stats):
        assert st.primaryRays == (min(H, b + 4) - max(0, a - 4)) * W                     # the band + the 4 rows its filter input needs, not + 66
    # what travelled: every rank sends each neighbour the rows of its own band that lie within 62 rows of that neighbour's band
    expect = [0] * len(bands)
    for r, (a, b) in enumerate(bands):
        for q, (qa, qb) in enumerate(bands):
            if q != r:
                above = max(0, min(b, qa) - max(a, qa - rt64.HALO_ROWS)); below = max(0, min(b, qb + rt64.HALO_ROWS) - max(a, qb))
                expect[r] += (above + below) * W * rt64.HALO_BYTES_PER_PIXEL * 4          # (4 frames)
    assert box.bytes_sent == expect
    # the same bands with the halo re-rendered (the default) give the same frame: both modes agree with the single device
    images2, stats2, _ = _render_bands(rt64_lib, sample_data, bands, 4, gi_samples, exchange=False)
    for im, ref in full.items():
        assert np.array_equal(images2[im], ref), im
    assert stats2[0].primaryRays == (min(H, bands[0][1] + 66)) * W


def test_halo_exchange_is_refused_when_the_layout_does_not_match_the_tile(rt64_lib, sample_data):
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    try:
        s.set_view_description(gi_samples=1, denoiser=True)
        cb = rt64.HALO_EXCHANGE(lambda user, regions, count: None)
        bad = (C.c_int * 3)(0, 90, H - 1)
        assert rt64_lib.SetDeviceHaloExchange(s.device, C.cast(cb, C.c_void_p), None, bad, 0, 2) == 0          # starts[count] must be the frame height
        good = (C.c_int * 3)(0, 90, H)
        assert rt64_lib.SetDeviceHaloExchange(s.device, C.cast(cb, C.c_void_p), None, good, 0, 2) == 1
        s.set_tile(0, 80)                                                                                        # not the band the layout gives rank 0
        s.draw()                                                                                                 # RT64_DrawDevice swallows the exception like the reference's (rt64_device.cpp:1234-1241) and keeps the message
        assert "halo exchange" in rt64_lib.last_error()
        assert rt64_lib.SetDeviceHaloExchange(s.device, None, None, None, 0, 0) == 1                             # off again: the band re-renders its halo
        s.draw()
    finally:
        s.close()
