#!/usr/bin/env python3
"""Host-side cost of one enqueued frame (sync_present = 0), split by call: tools/host_overhead.py  (diagnosis helper, 1 GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene, tiles

W, H, K = int(os.environ.get("RT64_W", 1920)), int(os.environ.get("RT64_H", 1080)), 300
lib = rt64.Library()
data = sample_scene.make_sample_scene()
scene = sample_scene.Rt64Scene(lib, data, W, H, hip_device=0)
for _ in range(20):
    scene.draw()
scene.option("sync_present", 0)
dst = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")


def timed(fn, label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-28s host %.1f us/step   incl. drain %.1f us/step" % (label, (t1 - t0) * 1e6 / K, (t2 - t0) * 1e6 / K))


timed(lambda: lib.DrawDevice(scene.device, 1, 16.0), "DrawDevice only")
timed(scene.draw, "scene.draw (4 ctypes calls)")
timed(lambda: (scene.draw(), lib.CopyDeviceImage(scene.device, rt64.IMAGE_FINAL_RGBA8, dst.data_ptr(), dst.numel())), "draw + CopyDeviceImage")
# the N > 1 loop on a world of one: RT64_DrawDevice + RT64_SubmitGather, rows through RCCL / stored directly into the frame slots
import ctypes as C
uid = (C.c_uint8 * rt64.GATHER_ID_BYTES)(); lib.GetGatherUniqueId(uid, len(uid))
g = lib.CreateGather(scene.device, uid, len(uid), 0, 1, 0)
timed(lambda: (lib.DrawDevice(scene.device, 1, 16.0), lib.SubmitGather(g)), "Draw + SubmitGather (rows)")
timed(lambda: (scene.draw(), lib.SubmitGather(g)), "scene.draw + SubmitGather")
h = (C.c_uint8 * 64)(); lib.GetGatherDirectHandle(g, h, 64); lib.SetGatherDirect(g, h, 64, 1)
timed(lambda: (lib.DrawDevice(scene.device, 1, 16.0), lib.SubmitGather(g)), "Draw + SubmitGather (direct)")
timed(lambda: (scene.draw(), lib.SubmitGather(g)), "scene.draw + Submit (direct)")
lib.SetGatherDirect(g, None, 0, 0); lib.DestroyGather(g)
scene.option("profile_passes", 0)
timed(lambda: lib.DrawDevice(scene.device, 1, 16.0), "DrawDevice, no pass events")
scene.option("lean_frames", 0)
timed(lambda: lib.DrawDevice(scene.device, 1, 16.0), "  ... full (non-lean) frame")
scene.close()
