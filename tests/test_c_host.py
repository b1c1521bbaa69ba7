"""tools/sample_host.c -- the reference's sample application as a plain C host of librt64.so (dlopen + the RT64_LIBRARY table).
CPU: it builds from include/rt64.h alone and its asset readers (PNG via zlib, OBJ) produce the bytes the Python harness loads.
GPU: the frame it renders through the C ABI is byte-identical to the frame the Python/ctypes harness renders."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tools", "sample_host")


def build_host():
    src = os.path.join(ROOT, "tools", "sample_host.c")
    if not os.path.exists(HOST) or os.path.getmtime(HOST) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", HOST, "-ldl", "-lz", "-lm"])
    return HOST


def fnv1a(b):
    h = 1469598103934665603
    for x in memoryview(b).cast("B"):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def test_c_host_builds_and_reads_the_sample_assets_like_the_python_harness(sample_data):
    host = build_host()
    out = json.loads(subprocess.check_output([host, "--selftest", "--assets", os.path.join(ROOT, "assets", "sample")]))
    by_name = {t.name: t for t in sample_data.textures}
    for name in ("grass_nrm.png", "grass_spc.png", "clouds.png", "tiles_dif.png", "tiles_nrm.png", "tiles_spc.png"):
        t = by_name[name]
        w, h, s, _fnv = out[name]
        flat = t.data.reshape(-1, 4).astype(np.uint64)
        assert (w, h) == (t.width, t.height) and s == int((flat * np.array([1, 2, 3, 4], dtype=np.uint64)).sum()), name
    n, h, _pn = out["sphere.obj"]
    v = np.ascontiguousarray(sample_data.meshes[0].vertices)
    assert n == len(v) == 960 and h == fnv1a(v.tobytes())


def test_c_host_reports_a_missing_library_or_device():
    host = build_host()
    r = subprocess.run([host, "--width", "64", "--height", "36", "--frames", "1"], env=dict(os.environ, RT64_LIBRARY_PATH="/nonexistent/librt64.so"), capture_output=True, text=True)
    assert r.returncode == 2 and "failed to load the library" in r.stderr


@pytest.mark.gpu
def test_c_host_frame_equals_the_python_hosts_frame(rt64_lib, sample_data):
    """RT64_LoadLibrary -> setupRT64Scene (main.cpp:201-412) -> 3 x the WM_PAINT calls -> RT64_ReadbackDevice, all from C: same back
    buffer, byte for byte, as the ctypes harness driving the same exports; the centre pixel picks the sphere (main.cpp:76-83)."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    host = build_host()
    W, H = 640, 360
    env = dict(os.environ, RT64_LIBRARY_PATH=rt64_lib.path)
    r = subprocess.run([host, "--width", str(W), "--height", str(H), "--frames", "3", "--assets", os.path.join(ROOT, "assets", "sample")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = next(l for l in r.stdout.splitlines() if l.startswith("{"))
    got = json.loads(line)
    s = sample_scene.Rt64Scene(rt64_lib, sample_data, W, H, hip_device=0)
    try:
        for _ in range(3):
            s.draw()
        frame = s.readback(rt64.IMAGE_FINAL_RGBA8)
    finally:
        s.close()
    assert got["checksum"] == int(frame.astype(np.uint64).sum()) and got["fnv1a"] == fnv1a(np.ascontiguousarray(frame).tobytes())
    assert got["picked_center"] == "sphere" and got["gpu_ms_last_frame"] > 0.0
