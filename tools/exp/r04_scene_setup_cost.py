#!/usr/bin/env python3
"""Where does the time of one small GPU test go?  (scene set-up through the C ABI, one frame, tear-down)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene
from oracle import oracle_py
lib = rt64.Library()
data = sample_scene.make_sample_scene()
for rep in range(3):
    T = [time.perf_counter()]
    def lap(name):
        T.append(time.perf_counter()); print("  %-28s %.1f ms" % (name, (T[-1] - T[-2]) * 1e3))
    dev = lib.CreateDeviceHeadless(320, 180, 0); lap("CreateDeviceHeadless")
    sc = lib.CreateScene(dev); lib.SetSceneDescription(sc, data.desc); view = lib.CreateView(sc); lap("CreateScene + CreateView")
    texs = []
    for t in data.textures:
        d = rt64.TEXTURE_DESC(); buf, pitch = t.upload_buffer(); d.bytes = buf.ctypes.data; d.byteCount = buf.nbytes; d.format = t.format
        if t.format == rt64.TEXTURE_FORMAT_RGBA8: d.width, d.height, d.rowPitch = t.width, t.height, pitch
        else: d.width = d.height = d.rowPitch = -1
        texs.append(lib.CreateTexture(dev, d)); lap("CreateTexture " + t.name)
    for h in texs: lib.DestroyTexture(h)
    lap("DestroyTexture x%d" % len(texs))
    lib.DestroyDevice(dev); lap("DestroyDevice")
    s = sample_scene.Rt64Scene(lib, data, 320, 180, hip_device=0); lap("Rt64Scene()")
    s.draw(); lap("first draw")
    s.draw(); lap("second draw")
    s.readback(rt64.IMAGE_PRIMARY_HIT); lap("readback PRIMARY_HIT")
    s.close(); lap("close")
    o = oracle_py.OracleScene(data); lap("OracleScene()")
    o.render(320, 180); lap("oracle render 1"); o.render(320, 180); lap("oracle render 2"); o.close(); lap("oracle close")
    print("---")
