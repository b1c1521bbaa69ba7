"""ctypes binding of the CPU oracle (oracle/liboracle_rt64.so).  TEST INFRASTRUCTURE.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product.
`OracleScene(scene_data)` consumes the same neutral SceneData the HIP library is driven with.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle_rt64.so")


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class V4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class M4(C.Structure):
    _fields_ = [("m", (C.c_float * 4) * 4)]

    @staticmethod
    def from_rows(rows):
        m = M4()
        for r in range(4):
            for c in range(4):
                m.m[r][c] = float(rows[r][c])
        return m

    def to_numpy(self):
        return np.array([[self.m[r][c] for c in range(4)] for r in range(4)], dtype=np.float32)


class OInstanceDesc(C.Structure):
    _fields_ = [("mesh", C.c_void_p), ("transform", M4), ("previousTransform", M4),
                ("diffuse", C.c_void_p), ("normal", C.c_void_p), ("specular", C.c_void_p),
                ("shaderId", C.c_uint32), ("filter", C.c_uint32), ("hAddr", C.c_uint32), ("vAddr", C.c_uint32),
                ("shaderFlags", C.c_int), ("material", C.c_byte * 132), ("flags", C.c_uint),
                ("scissorRect", C.c_int * 4), ("viewportRect", C.c_int * 4)]


class OFrameParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("tileY0", C.c_int), ("tileY1", C.c_int),
                ("view", M4), ("fovRadians", C.c_float), ("nearDist", C.c_float), ("farDist", C.c_float),
                ("canReproject", C.c_int),
                ("diSamples", C.c_uint), ("giSamples", C.c_uint), ("maxLights", C.c_uint),
                ("denoiserEnabled", C.c_int), ("denoiserMode", C.c_int),
                ("motionBlurStrength", C.c_float), ("motionBlurSamples", C.c_uint), ("maxReflections", C.c_int),
                ("bruteForce", C.c_int), ("cullBehindOpaque", C.c_int), ("threads", C.c_int), ("resolutionScale", C.c_float),
                ("upscaler", C.c_int), ("upscalerMode", C.c_int), ("giBounces", C.c_int), ("primarySpp", C.c_int)]


_FP = C.POINTER(C.c_float)


class OFrameResult(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int),
                ("finalRGBA8", C.POINTER(C.c_uint8)), ("outputRGBA32F", _FP),
                ("shadingPosition", _FP), ("shadingNormal", _FP), ("shadingSpecular", _FP), ("diffuse", _FP),
                ("instanceId", C.POINTER(C.c_int32)),
                ("directLight", _FP), ("indirectLight", _FP), ("filteredDirect", _FP), ("filteredIndirect", _FP),
                ("reflection", _FP), ("refraction", _FP), ("transparent", _FP), ("viewDirection", _FP), ("normal", _FP),
                ("flow", _FP), ("reactiveMask", _FP), ("lockMask", _FP), ("depth", _FP),
                ("primaryHit", C.POINTER(C.c_uint32)),
                ("primaryRays", C.c_uint64), ("shadowRays", C.c_uint64), ("indirectRays", C.c_uint64),
                ("reflectionRays", C.c_uint64), ("refractionRays", C.c_uint64),
                ("nodesVisited", C.c_uint64), ("trianglesTested", C.c_uint64),
                ("nodesVisitedPrimary", C.c_uint64), ("trianglesTestedPrimary", C.c_uint64),
                ("nodesVisitedShadow", C.c_uint64), ("trianglesTestedShadow", C.c_uint64),
                ("secondsBuild", C.c_double), ("secondsRender", C.c_double), ("screenWidth", C.c_int), ("screenHeight", C.c_int),
                ("backgroundRGBA8", C.POINTER(C.c_uint8)), ("upscaledRGBA32F", _FP), ("pixelJitter", C.c_float * 2)]


class ONode(C.Structure):
    _fields_ = [("lmin", C.c_float * 3), ("lmax", C.c_float * 3), ("rmin", C.c_float * 3), ("rmax", C.c_float * 3),
                ("left", C.c_uint32), ("right", C.c_uint32), ("parent", C.c_uint32), ("pad", C.c_uint32)]


class OBvh(C.Structure):
    _fields_ = [("count", C.c_uint32), ("nodes", C.POINTER(ONode)), ("sortedIndex", C.POINTER(C.c_uint32)),
                ("morton", C.POINTER(C.c_uint32)), ("bmin", C.c_float * 3), ("bmax", C.c_float * 3)]


NODE_DTYPE = np.dtype([("lmin", "<f4", 3), ("lmax", "<f4", 3), ("rmin", "<f4", 3), ("rmax", "<f4", 3),
                       ("left", "<u4"), ("right", "<u4"), ("parent", "<u4"), ("pad", "<u4")])
TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("prim", "<u4"), ("v1", "<f4", 3), ("pad1", "<u4"), ("v2", "<f4", 3), ("pad2", "<u4")])

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    P = C.c_void_p
    sig = {
        "oracle_scene_create": (P, []), "oracle_scene_destroy": (None, [P]),
        "oracle_scene_set_desc": (None, [P, P]), "oracle_scene_set_lights": (None, [P, P, C.c_int]),
        "oracle_scene_set_bluenoise": (None, [P, P]), "oracle_scene_set_sky": (None, [P, P]),
        "oracle_scene_add_instance": (C.c_int, [P, C.POINTER(OInstanceDesc)]),
        "oracle_scene_set_instance": (None, [P, C.c_int, C.POINTER(OInstanceDesc)]),
        "oracle_texture_create_rgba8": (P, [P, C.c_int, C.c_int, C.c_int]),
        "oracle_texture_create_dds": (P, [P, C.c_size_t]), "oracle_texture_destroy": (None, [P]),
        "oracle_texture_info": (C.c_int, [P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "oracle_texture_mip": (C.POINTER(C.c_uint8), [P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "oracle_texture_sample": (None, [P] + [C.c_float] * 6 + [C.c_int] * 3 + [_FP]),
        "oracle_mesh_create": (P, [C.c_int]), "oracle_mesh_set": (None, [P, P, C.c_int, C.c_int, P, C.c_int]),
        "oracle_mesh_destroy": (None, [P]), "oracle_mesh_bvh": (C.POINTER(OBvh), [P]), "oracle_mesh_tris": (P, [P]),
        "oracle_render": (C.c_int, [P, C.POINTER(OFrameParams), C.POINTER(OFrameResult)]),
        "oracle_scene_tlas": (C.POINTER(OBvh), [P]), "oracle_scene_frame_count": (C.c_uint32, [P]),
        "oracle_init_rand": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
        "oracle_next_rand": (C.c_float, [C.POINTER(C.c_uint32)]),
        "oracle_halton": (C.c_float, [C.c_int, C.c_int]),
        "oracle_upscaler_info": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "oracle_rgb_to_hsl": (None, [_FP, _FP]), "oracle_hsl_to_rgb": (None, [_FP, _FP]),
        "oracle_fake_envmap_uv": (None, [_FP, C.c_float, _FP]),
        "oracle_perspective_fov_rh": (None, [C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(M4)]),
        "oracle_matrix_inverse": (C.c_int, [C.POINTER(M4), C.POINTER(M4)]),
        "oracle_morton30": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
        "oracle_f32_to_f16": (C.c_uint16, [C.c_float]), "oracle_f16_to_f32": (C.c_float, [C.c_uint16]),
        "oracle_decode_bc7_block": (None, [P, P]),
        "oracle_decode_combiner": (None, [C.c_uint32, C.POINTER(C.c_int)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def bvh_to_numpy(bvh_ptr):
    b = bvh_ptr.contents
    n = b.count
    inner = max(n - 1, 1)
    nodes = np.ctypeslib.as_array(C.cast(b.nodes, C.POINTER(C.c_uint8)), shape=(inner * 64,)).view(NODE_DTYPE).copy()
    sorted_index = np.ctypeslib.as_array(b.sortedIndex, shape=(n,)).copy()
    morton = np.ctypeslib.as_array(b.morton, shape=(n,)).copy()
    return {"count": n, "nodes": nodes, "sortedIndex": sorted_index, "morton": morton,
            "bmin": np.array(list(b.bmin), dtype=np.float32), "bmax": np.array(list(b.bmax), dtype=np.float32)}


def host_threads():
    """Threads the oracle may really use: the affinity mask, cut by the cgroup's CPU quota when there is one.  (OpenMP's own default is one thread per
    LOGICAL CPU of the machine -- 256 on the GPU box, whose jobs get 16 cores: every parallel region then spends its time in oversubscribed barriers,
    1.2 s for a 320 x 180 frame that takes 12 ms.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(path).read().split()
            if path.endswith("cpu.max"):
                if t[0] != "max":
                    n = min(n, max(1, int(float(t[0]) / float(t[1]) + 0.5)))
            else:
                q = int(t[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, min(n, 64))


class OracleScene:
    """Feeds a sample_scene.SceneData to the oracle and renders frames."""

    def __init__(self, data):
        L = lib()
        self.L, self.data = L, data
        self.scene = L.oracle_scene_create()
        self._keep = []
        L.oracle_scene_set_desc(self.scene, C.addressof(data.desc))
        if data.lights:
            arr = (type(data.lights[0]) * len(data.lights))(*data.lights)
            L.oracle_scene_set_lights(self.scene, C.addressof(arr), len(data.lights))
        bn = np.ascontiguousarray(data.bluenoise)
        L.oracle_scene_set_bluenoise(self.scene, bn.ctypes.data)
        self.textures = []
        for t in data.textures:
            buf, pitch = t.upload_buffer() if hasattr(t, "upload_buffer") else (np.ascontiguousarray(t.data), t.width * 4)
            if t.format == 1:
                h = L.oracle_texture_create_rgba8(buf.ctypes.data, t.width, t.height, pitch)
            else:
                h = L.oracle_texture_create_dds(buf.ctypes.data, buf.nbytes)
            assert h, t.name
            self.textures.append(h)
        if data.sky is not None:
            L.oracle_scene_set_sky(self.scene, self.textures[data.sky])
        self.meshes = []
        for m in data.meshes:
            h = L.oracle_mesh_create(m.flags)
            self.set_mesh(h, m.vertices, m.indices)
            self.meshes.append(h)
        for k, inst in enumerate(data.instances):
            d = self._desc(inst)
            L.oracle_scene_add_instance(self.scene, C.byref(d))
        self.params = dict(diSamples=0, giSamples=0, maxLights=12, denoiserEnabled=0, denoiserMode=0,
                           motionBlurStrength=0.0, motionBlurSamples=32, maxReflections=2, upscaler=0, upscalerMode=0, giBounces=1, primarySpp=1)

    def set_mesh(self, handle, vertices, indices):
        v = np.ascontiguousarray(vertices); i = np.ascontiguousarray(indices, dtype=np.uint32)
        self.L.oracle_mesh_set(handle, v.ctypes.data, len(v), v.dtype.itemsize, i.ctypes.data, len(i))

    def _desc(self, inst):
        d = OInstanceDesc()
        d.mesh = self.meshes[inst.mesh]
        d.transform = M4.from_rows(inst.transform); d.previousTransform = M4.from_rows(inst.previous_transform)
        d.diffuse = self.textures[inst.diffuse]
        d.normal = self.textures[inst.normal] if inst.normal is not None else None
        d.specular = self.textures[inst.specular] if inst.specular is not None else None
        s = self.data
        d.shaderId, d.filter, d.hAddr, d.vAddr, d.shaderFlags = s.shader_id, s.shader_filter, s.shader_haddr, s.shader_vaddr, s.shader_flags
        C.memmove(d.material, C.byref(inst.material), 132)
        d.flags = inst.flags
        for k, v in enumerate(getattr(inst, "scissor", None) or (0, 0, 0, 0)):
            d.scissorRect[k] = int(v)
        for k, v in enumerate(getattr(inst, "viewport", None) or (0, 0, 0, 0)):
            d.viewportRect[k] = int(v)
        return d

    def set_instance(self, k, inst):
        d = self._desc(inst)
        self.L.oracle_scene_set_instance(self.scene, k, C.byref(d))

    def render(self, width, height, brute_force=False, cull_behind_opaque=True, threads=0, can_reproject=True, tile=None, images=True, **over):
        """One frame.  images=False renders it (history, frame count and counters advance) and returns the counters only -- the intermediate frames of a
        multi-frame test skip the copies of two dozen images per frame."""
        p = OFrameParams()
        p.width, p.height = width, height
        p.tileY0, p.tileY1 = tile if tile else (0, height)
        p.view = M4.from_rows(self.data.view)
        p.fovRadians, p.nearDist, p.farDist = self.data.fov, self.data.near, self.data.far
        p.canReproject = int(can_reproject)
        q = dict(self.params); q.update(over)
        for k, v in q.items():
            setattr(p, k, v)
        p.bruteForce = int(brute_force); p.cullBehindOpaque = int(cull_behind_opaque); p.threads = threads if threads > 0 else host_threads()
        r = OFrameResult()
        ok = self.L.oracle_render(self.scene, C.byref(p), C.byref(r))
        assert ok
        counters = {k: getattr(r, k) for k in ("primaryRays", "shadowRays", "indirectRays", "reflectionRays", "refractionRays",
                                               "nodesVisited", "trianglesTested", "nodesVisitedPrimary", "trianglesTestedPrimary",
                                               "nodesVisitedShadow", "trianglesTestedShadow", "secondsBuild", "secondsRender")}
        if not images:
            return {"counters": counters}
        screen_w, screen_h = width, height
        width, height = r.width, r.height            # render size (differs from the screen size with resolutionScale)
        n = width * height

        def img(ptr, ch, dt):
            a = np.ctypeslib.as_array(ptr, shape=(n * ch,)).copy()
            a = a.reshape(height, width, ch) if ch > 1 else a.reshape(height, width)
            return a.astype(dt, copy=False)
        out = {
            "final": np.ctypeslib.as_array(r.finalRGBA8, shape=(screen_h * screen_w * 4,)).copy().reshape(screen_h, screen_w, 4),
            "output": img(r.outputRGBA32F, 4, np.float32),
            "shadingPosition": img(r.shadingPosition, 4, np.float32), "shadingNormal": img(r.shadingNormal, 4, np.float32),
            "shadingSpecular": img(r.shadingSpecular, 4, np.float32), "diffuse": img(r.diffuse, 4, np.float32),
            "instanceId": img(r.instanceId, 1, np.int32), "directLight": img(r.directLight, 4, np.float32),
            "indirectLight": img(r.indirectLight, 4, np.float32), "filteredDirect": img(r.filteredDirect, 4, np.float32),
            "filteredIndirect": img(r.filteredIndirect, 4, np.float32), "reflection": img(r.reflection, 4, np.float32),
            "refraction": img(r.refraction, 4, np.float32), "transparent": img(r.transparent, 4, np.float32),
            "viewDirection": img(r.viewDirection, 4, np.float32), "normal": img(r.normal, 4, np.float32),
            "flow": img(r.flow, 2, np.float32), "reactiveMask": img(r.reactiveMask, 1, np.float32),
            "lockMask": img(r.lockMask, 1, np.float32), "depth": img(r.depth, 1, np.float32),
            "primaryHit": img(r.primaryHit, 4, np.uint32),
        }
        out["background"] = (np.ctypeslib.as_array(r.backgroundRGBA8, shape=(screen_h * screen_w * 4,)).copy().reshape(screen_h, screen_w, 4)
                             if r.backgroundRGBA8 else None)
        out["upscaled"] = (np.ctypeslib.as_array(r.upscaledRGBA32F, shape=(screen_h * screen_w * 4,)).copy().reshape(screen_h, screen_w, 4)
                           if r.upscaledRGBA32F else None)
        out["pixelJitter"] = (float(r.pixelJitter[0]), float(r.pixelJitter[1]))
        out["counters"] = counters
        return out

    def mesh_bvh(self, k):
        return bvh_to_numpy(self.L.oracle_mesh_bvh(self.meshes[k]))

    def mesh_tris(self, k):
        n = self.L.oracle_mesh_bvh(self.meshes[k]).contents.count
        p = self.L.oracle_mesh_tris(self.meshes[k])
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n * 48,)).view(TRI_DTYPE).copy()

    def tlas(self):
        return bvh_to_numpy(self.L.oracle_scene_tlas(self.scene))

    def close(self):
        if self.scene:
            self.L.oracle_scene_destroy(self.scene)
            for h in self.meshes:
                self.L.oracle_mesh_destroy(h)
            for h in self.textures:
                self.L.oracle_texture_destroy(h)
            self.scene = None
