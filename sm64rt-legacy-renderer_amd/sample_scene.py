"""The reference's sample scene, re-created call for call through the RT64 C ABI.

Follows /root/reference/src/sample/main.cpp:201-412 (setupRT64Scene) and :97-134 (per-frame calls): scene description,
shader 0x01200a00, one light, view, 7 textures, sky plane, sphere mesh from sphere.obj (unrolled, uv = acos(n.xy)),
material, two raster-only HUD triangles, the ray-traced sphere instance and the floor quad scaled x10.

`SceneData` is a neutral description (numpy arrays + ctypes PODs).  `Rt64Scene` feeds it to librt64.so through the
function table; oracle/oracle_py.py feeds the same description to the CPU oracle in tests.  Assets come from assets/
(imported by tools/import_sample_assets.py) -- never from /root/reference at run time.

Optional stress variant: `subdiv`/`floor_grid` replace the 320-triangle sphere / 2-triangle floor by finer
tessellations of the same shapes so that the BVH no longer fits in cache.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import rt64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")

# VERTEX of main.cpp:36-41: float4 position, float3 normal, float2 uv, float4 input1 = 52 bytes
VERTEX_DTYPE = np.dtype([("position", "<f4", 4), ("normal", "<f4", 3), ("uv", "<f4", 2), ("input1", "<f4", 4)])
assert VERTEX_DTYPE.itemsize == 52


@dataclass
class TextureData:
    name: str
    format: int                      # rt64.TEXTURE_FORMAT_*
    data: np.ndarray                 # RGBA8: [h, w, 4] uint8 ; DDS: raw file bytes (uint8 1-D)
    width: int = -1
    height: int = -1
    row_pitch: int = 0               # RGBA8 only: bytes between rows as handed to RT64_CreateTexture (0 = 4 * width, tightly packed)

    def upload_buffer(self):
        """(bytes, rowPitch) as the host passes them: `data` itself, or -- with a row_pitch -- rows padded with bytes that must never be sampled."""
        if self.format != rt64.TEXTURE_FORMAT_RGBA8 or self.row_pitch in (0, self.width * 4):
            return np.ascontiguousarray(self.data), self.width * 4
        buf = np.full((self.height, self.row_pitch), 0xAB, dtype=np.uint8)
        buf[:, :self.width * 4] = np.ascontiguousarray(self.data).reshape(self.height, self.width * 4)
        return buf, self.row_pitch


@dataclass
class MeshData:
    name: str
    flags: int
    vertices: np.ndarray             # structured VERTEX_DTYPE
    indices: np.ndarray              # uint32


@dataclass
class InstanceData:
    name: str
    mesh: int
    transform: np.ndarray            # [4,4] float32 row-major, row-vector convention
    previous_transform: np.ndarray
    diffuse: int
    normal: Optional[int]
    specular: Optional[int]
    material: rt64.MATERIAL
    flags: int = 0
    scissor: Optional[tuple] = None  # RT64_RECT (x, y, w, h), origin bottom-left; None = unset
    viewport: Optional[tuple] = None


@dataclass
class SceneData:
    desc: rt64.SCENE_DESC
    shader_id: int
    shader_filter: int
    shader_haddr: int
    shader_vaddr: int
    shader_flags: int
    lights: List[rt64.LIGHT]
    textures: List[TextureData]
    sky: Optional[int]
    meshes: List[MeshData]
    instances: List[InstanceData]
    view: np.ndarray                 # [4,4]
    fov: float
    near: float
    far: float
    bluenoise: np.ndarray = field(default=None)


def _load_png_rgba8(path):
    """stbi_load(path, ..., STBI_rgb_alpha) (main.cpp:165): 8-bit RGBA; 16-bit sources keep their high byte."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("I;16", "I;16B", "I"):
        a = np.asarray(im).astype(np.uint32)
        g = (a >> 8).astype(np.uint8)
        return np.ascontiguousarray(np.stack([g, g, g, np.full_like(g, 255)], axis=-1))
    return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))


def load_obj_unrolled(path):
    """tinyobj::LoadObj(..., triangulate=true) + the unrolling loop of main.cpp:269-287."""
    pos, nrm, faces = [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                pos.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "vn":
                nrm.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "f":
                corners = [tuple(int(x) if x else 0 for x in c.split("/")) for c in p[1:]]
                for k in range(1, len(corners) - 1):        # fan triangulation
                    faces.append((corners[0], corners[k], corners[k + 1]))
    pos = np.asarray(pos, dtype=np.float32)
    nrm = np.asarray(nrm, dtype=np.float32)
    verts = np.zeros(len(faces) * 3, dtype=VERTEX_DTYPE)
    k = 0
    for tri in faces:
        for (vi, _ti, ni) in tri:
            n = nrm[ni - 1]
            verts["position"][k] = (pos[vi - 1][0], pos[vi - 1][1], pos[vi - 1][2], 1.0)
            verts["normal"][k] = n
            # main.cpp:278 acos(n.x), acos(n.y): evaluated in double and rounded once (a correctly rounded float acos; numpy's
            # float32 arccos and the C library's acosf differ by an ulp on a third of the inputs, tools/sample_host.c does the same)
            verts["uv"][k] = (np.float32(math.acos(float(n[0]))), np.float32(math.acos(float(n[1]))))
            verts["input1"][k] = (1.0, 1.0, 1.0, 1.0)
            k += 1
    return verts, np.arange(len(verts), dtype=np.uint32)


def _subdivide_sphere(verts, levels, centre, radius):
    """Stress variant: split every triangle 4-ways `levels` times, re-projecting onto the sample's sphere."""
    p = verts["position"][:, :3].astype(np.float64).reshape(-1, 3, 3)
    c = np.asarray(centre, dtype=np.float64)
    for _ in range(levels):
        a, b, d = p[:, 0], p[:, 1], p[:, 2]

        def mid(x, y):
            m = (x + y) * 0.5 - c
            return c + m / np.linalg.norm(m, axis=1, keepdims=True) * radius
        ab, bd, da = mid(a, b), mid(b, d), mid(d, a)
        p = np.concatenate([np.stack([a, ab, da], 1), np.stack([ab, b, bd], 1), np.stack([da, bd, d], 1), np.stack([ab, bd, da], 1)], 0)
    flat = p.reshape(-1, 3)
    n = (flat - c) / np.linalg.norm(flat - c, axis=1, keepdims=True)
    out = np.zeros(len(flat), dtype=VERTEX_DTYPE)
    out["position"][:, :3] = flat.astype(np.float32)
    out["position"][:, 3] = 1.0
    out["normal"] = n.astype(np.float32)
    out["uv"][:, 0] = np.arccos(out["normal"][:, 0].astype(np.float64)).astype(np.float32)
    out["uv"][:, 1] = np.arccos(out["normal"][:, 1].astype(np.float64)).astype(np.float32)
    out["input1"] = 1.0
    return out, np.arange(len(out), dtype=np.uint32)


def _floor_mesh(grid):
    """main.cpp:377-392 floor quad; grid > 1 tessellates the same quad into grid x grid cells (stress variant)."""
    if grid <= 1:
        v = np.zeros(4, dtype=VERTEX_DTYPE)
        v["position"] = [(-1.5, 0.0, -1.0, 1.0), (1.0, 0.0, -1.0, 1.0), (-1.5, 0.0, 1.0, 1.0), (1.0, 0.0, 1.0, 1.0)]
        v["uv"] = [(0.0, 0.0), (1.0, 0.0), (0.0, 1.0), (1.0, 1.0)]
        v["normal"] = (0.0, 1.0, 0.0)
        v["input1"] = 1.0
        return v, np.array([2, 1, 0, 1, 2, 3], dtype=np.uint32)
    g = grid + 1
    u, w = np.meshgrid(np.linspace(0.0, 1.0, g, dtype=np.float32), np.linspace(0.0, 1.0, g, dtype=np.float32), indexing="xy")
    v = np.zeros(g * g, dtype=VERTEX_DTYPE)
    v["position"][:, 0] = (-1.5 + 2.5 * u).ravel()
    v["position"][:, 2] = (-1.0 + 2.0 * w).ravel()
    v["position"][:, 3] = 1.0
    v["uv"][:, 0] = u.ravel()
    v["uv"][:, 1] = w.ravel()
    v["normal"] = (0.0, 1.0, 0.0)
    v["input1"] = 1.0
    i, j = np.meshgrid(np.arange(grid), np.arange(grid), indexing="xy")
    v0 = (j * g + i).ravel(); v1 = v0 + 1; v2 = v0 + g; v3 = v2 + 1
    idx = np.stack([v2, v1, v0, v1, v2, v3], axis=1).astype(np.uint32).ravel()
    return v, idx


def base_material():
    """RT64.baseMaterial, main.cpp:292-310 (the struct is static storage: unset members are 0)."""
    m = rt64.MATERIAL()
    m.ignoreNormalFactor = 0.0; m.uvDetailScale = 1.0
    m.reflectionFactor = 0.0; m.reflectionFresnelFactor = 1.0; m.reflectionShineFactor = 0.0; m.refractionFactor = 0.0
    m.specularColor = rt64.VECTOR3(1.0, 1.0, 1.0); m.specularExponent = 1.0
    m.solidAlphaMultiplier = 1.0; m.shadowAlphaMultiplier = 1.0
    m.diffuseColorMix = rt64.VECTOR4(0.0, 0.0, 0.0, 0.0)
    m.selfLight = rt64.VECTOR3(0.0, 0.0, 0.0)
    m.lightGroupMaskBits = 0xFFFFFFFF
    m.fogColor = rt64.VECTOR3(0.3, 0.5, 0.7); m.fogMul = 1.0; m.fogOffset = 0.0; m.fogEnabled = 0
    m.lockMask = 0.0
    return m


def copy_material(m):
    c = rt64.MATERIAL()
    C.memmove(C.byref(c), C.byref(m), C.sizeof(rt64.MATERIAL))
    return c


def make_sample_scene(subdiv=0, floor_grid=1, assets=ASSETS) -> SceneData:
    res = os.path.join(assets, "sample")
    desc = rt64.SCENE_DESC()                                            # main.cpp:204-212
    desc.ambientBaseColor = rt64.VECTOR3(0.1, 0.1, 0.1); desc.ambientNoGIColor = rt64.VECTOR3(0.2, 0.2, 0.2)
    desc.eyeLightDiffuseColor = rt64.VECTOR3(0.08, 0.08, 0.08); desc.eyeLightSpecularColor = rt64.VECTOR3(0.04, 0.04, 0.04)
    desc.skyDiffuseMultiplier = rt64.VECTOR3(1.0, 1.0, 1.0); desc.skyHSLModifier = rt64.VECTOR3(0.0, 0.0, 0.0)
    desc.skyYawOffset = 0.0; desc.giDiffuseStrength = 0.7; desc.giSkyStrength = 0.35

    light = rt64.LIGHT()                                                # main.cpp:220-231
    light.position = rt64.VECTOR3(15000.0, 30000.0, 15000.0); light.attenuationRadius = 1e9; light.pointRadius = 5000.0
    light.diffuseColor = rt64.VECTOR3(0.8, 0.75, 0.65); light.specularColor = rt64.VECTOR3(0.8, 0.75, 0.65)
    light.shadowOffset = 0.0; light.attenuationExponent = 1.0; light.groupBits = rt64.LIGHT_GROUP_DEFAULT

    def dds(name):
        raw = np.fromfile(os.path.join(res, name), dtype=np.uint8)
        return TextureData(name, rt64.TEXTURE_FORMAT_DDS, raw)

    def png(name):
        a = _load_png_rgba8(os.path.join(res, name))
        return TextureData(name, rt64.TEXTURE_FORMAT_RGBA8, a, a.shape[1], a.shape[0])

    # creation order of main.cpp:237-241,329-331
    textures = [dds("grass_dif.dds"), png("grass_nrm.png"), png("grass_spc.png"), png("clouds.png"),
                png("tiles_dif.png"), png("tiles_nrm.png"), png("tiles_spc.png")]
    T_GRASS_DIF, T_GRASS_NRM, T_GRASS_SPC, T_CLOUDS, T_TILES_DIF, T_TILES_NRM, T_TILES_SPC = range(7)

    ident = np.eye(4, dtype=np.float32)
    view = np.eye(4, dtype=np.float32)                                  # main.cpp:250-258
    view[3, 1] = -2.0; view[3, 2] = -10.0

    sphere_v, sphere_i = load_obj_unrolled(os.path.join(res, "sphere.obj"))
    if subdiv > 0:
        sphere_v, sphere_i = _subdivide_sphere(sphere_v, subdiv, (0.0, 0.5, 0.0), 2.54558420181)
    rt_flags = rt64.MESH_RAYTRACE_ENABLED | rt64.MESH_RAYTRACE_FAST_TRACE | rt64.MESH_RAYTRACE_COMPACT   # main.cpp:289

    hud = np.zeros(3, dtype=VERTEX_DTYPE)                               # main.cpp:312-326
    hud["position"] = [(-1.0, 0.1, 0.0, 1.0), (-0.5, 0.1, 0.0, 1.0), (-0.75, 0.3, 0.0, 1.0)]
    hud["normal"] = (0.0, 1.0, 0.0)
    hud["uv"] = [(0.0, 0.0), (1.0, 0.0), (0.0, 1.0)]
    hud["input1"] = 1.0
    hud_alt = hud.copy()
    hud_alt["position"][:, 1] += np.float32(0.15)                      # main.cpp:336-338
    hud_idx = np.array([0, 1, 2], dtype=np.uint32)

    floor_v, floor_i = _floor_mesh(floor_grid)
    floor_t = np.diag([10.0, 10.0, 10.0, 1.0]).astype(np.float32)       # main.cpp:394-398

    meshes = [MeshData("sphere", rt_flags, sphere_v, sphere_i), MeshData("hudA", 0, hud, hud_idx),
              MeshData("hudB", 0, hud_alt, hud_idx), MeshData("floor", rt64.MESH_RAYTRACE_ENABLED, floor_v, floor_i)]
    M_SPHERE, M_HUD, M_HUD_ALT, M_FLOOR = range(4)
    mat = base_material()
    instances = [                                                       # creation order of main.cpp:356-411
        InstanceData("hudB", M_HUD_ALT, ident, ident, T_TILES_DIF, None, None, copy_material(mat), 0),
        InstanceData("sphere", M_SPHERE, ident, ident, T_GRASS_DIF, T_GRASS_NRM, T_GRASS_SPC, copy_material(mat), 0),
        InstanceData("hudA", M_HUD, ident, ident, T_GRASS_DIF, None, None, copy_material(mat), rt64.INSTANCE_RASTER_BACKGROUND),
        InstanceData("floor", M_FLOOR, floor_t, floor_t, T_TILES_DIF, T_TILES_NRM, T_TILES_SPC, copy_material(mat), 0),
    ]
    shader_flags = (rt64.SHADER_RASTER_ENABLED | rt64.SHADER_RAYTRACE_ENABLED | rt64.SHADER_NORMAL_MAP_ENABLED |
                    rt64.SHADER_SPECULAR_MAP_ENABLED)                   # main.cpp:216
    bn = np.fromfile(os.path.join(assets, "bluenoise_512x512_rgba8.bin"), dtype=np.uint8).reshape(512, 512, 4)
    return SceneData(desc, 0x01200a00, rt64.SHADER_FILTER_LINEAR, rt64.SHADER_ADDRESSING_WRAP, rt64.SHADER_ADDRESSING_WRAP,
                     shader_flags, [light], textures, T_CLOUDS, meshes, instances, view,
                     (45.0 * math.pi) / 180.0, 0.1, 1000.0, bn)


# ---- BASELINE.json configurations as bench.py runs them; tests/test_gpu_configs.py renders the same definitions -------

BENCH_CONFIGS = {
    "C2": dict(width=1920, height=1080, gi_samples=0, denoiser=False),
    "C3": dict(width=1920, height=1080, gi_samples=1, denoiser=True),
    "C4": dict(width=2560, height=1440, gi_samples=2, denoiser=True),      # + per-frame SetMesh refit of the UPDATABLE sphere
    "C5": dict(width=3840, height=2160, gi_samples=4, denoiser=True),      # + reflective floor
    # BASELINE.json's C4 / C5 as WORDED ("2-bounce GI 1440p 2spp", "4K 4spp full path trace"), through the library's path-tracing extensions
    # (device options primary_spp / gi_bounces; rules P1-P4 and B1-B3 in oracle/oracle_render.c): every primary sample carries one GI ray of two bounces.
    "C4-literal": dict(width=2560, height=1440, gi_samples=1, denoiser=True, primary_spp=2, gi_bounces=2),      # + per-frame refit, like C4
    "C5-literal": dict(width=3840, height=2160, gi_samples=1, denoiser=True, primary_spp=4, gi_bounces=2),      # + reflective floor, like C5
}


# How these definitions read BASELINE.json's wording where the reference has no such knob (bench.py prints it as config.deviation).
BENCH_DEVIATIONS = {
    "C4-literal": "BASELINE.json's C4 as worded: 2 primary samples per pixel (device option primary_spp = 2: two jittered sub-frames, composed outputs averaged), each with one GI ray of "
                  "TWO bounces (gi_bounces = 2), SVGF per sub-frame, per-frame refit. Both knobs are extensions of this library (the reference has neither); the oracle implements the same rules.",
    "C5-literal": "BASELINE.json's C5 as worded: 4 primary samples per pixel (primary_spp = 4), each a full path -- primary, shadow, one GI ray of two bounces (gi_bounces = 2), "
                  "reflection bounces on the floor -- SVGF per sub-frame. Both knobs are extensions of this library (the reference has neither); the oracle implements the same rules.",
    "C4": "BASELINE.json words C4 '2-bounce GI, 2 spp'; as run: giSamples = 2 (two GI rays per pixel), ONE bounce per GI ray, ONE primary sample per pixel. "
          "The reference has neither a bounce-depth nor a primary-spp knob (IndirectRayGen.hlsl:58-131 traces one bounce per sample; RT64_VIEW_DESC, rt64.h:172-182).",
    "C5": "BASELINE.json words C5 '4 spp full path trace'; as run: giSamples = 4 (four GI rays per pixel, one bounce each), ONE primary sample per pixel, reflective floor "
          "(reflectionFactor 0.3, two reflection bounces), SVGF. No primary multi-sampling, no multi-bounce paths: the reference has no such knobs.",
}


def c4_animation(data: "SceneData", frames=16):
    """C4: the sphere mesh becomes UPDATABLE and is displaced every frame, p += 0.1 n sin(frame 0.1 + p.y) (build-defined, seedless).
    Returns the vertex arrays of `frames` consecutive frames; the caller hands one to RT64_SetMesh per step (host copy + refit)."""
    m = data.meshes[0]
    m.flags |= rt64.MESH_RAYTRACE_UPDATABLE
    base = m.vertices.copy()
    out = []
    for f in range(frames):
        v = base.copy()
        v["position"][:, :3] += (0.1 * np.sin(f * 0.1 + base["position"][:, 1]))[:, None].astype(np.float32) * base["normal"]
        out.append(v)
    return out


def apply_bench_config(data: "SceneData", config: str):
    """Scene-side part of a config: C4 -> list of animated vertex arrays (else None); C5 -> floor reflectionFactor 0.3."""
    if config in ("C4", "C4-literal"):
        return c4_animation(data)
    if config in ("C5", "C5-literal"):
        for inst in data.instances:
            if inst.name == "floor":
                inst.material.reflectionFactor = 0.3
    return None


class Rt64Scene:
    """Drives librt64.so with a SceneData exactly like the sample drives rt64lib.dll."""

    def __init__(self, lib: rt64.Library, data: SceneData, width: int, height: int, hip_device: int = -1):
        self.lib, self.data, self.width, self.height = lib, data, width, height
        self.device = lib.CreateDeviceHeadless(width, height, hip_device)
        if not self.device:
            raise RuntimeError("RT64_CreateDeviceHeadless failed: " + lib.last_error())
        self.scene = lib.CreateScene(self.device)
        lib.SetSceneDescription(self.scene, data.desc)
        self.shader = lib.CreateShader(self.device, data.shader_id, data.shader_filter, data.shader_haddr, data.shader_vaddr, data.shader_flags)
        if not self.shader:
            raise RuntimeError("RT64_CreateShader failed: " + lib.last_error())
        self._lights = (rt64.LIGHT * len(data.lights))(*data.lights)
        self.view = lib.CreateView(self.scene)
        self.textures = []
        for t in data.textures:
            d = rt64.TEXTURE_DESC()
            buf, pitch = t.upload_buffer()
            d.bytes = buf.ctypes.data; d.byteCount = buf.nbytes; d.format = t.format
            if t.format == rt64.TEXTURE_FORMAT_RGBA8:
                d.width, d.height, d.rowPitch = t.width, t.height, pitch
            else:
                d.width = d.height = d.rowPitch = -1
            h = lib.CreateTexture(self.device, d)
            if not h:
                raise RuntimeError(f"RT64_CreateTexture({t.name}) failed: " + lib.last_error())
            self.textures.append(h)
        if data.sky is not None:
            lib.SetViewSkyPlane(self.view, self.textures[data.sky])
        self.meshes = []
        for m in data.meshes:
            h = lib.CreateMesh(self.device, m.flags)
            self.set_mesh(h, m.vertices, m.indices)
            self.meshes.append(h)
        self.instances = []
        self._desc_cache = {}
        for inst in data.instances:
            h = lib.CreateInstance(self.scene)
            self.instances.append(h)
            self.set_instance(len(self.instances) - 1, inst)
        self.view_desc = None

    def set_mesh(self, handle, vertices, indices):
        v = np.ascontiguousarray(vertices); i = np.ascontiguousarray(indices, dtype=np.uint32)
        self.lib.SetMesh(handle, v.ctypes.data, len(v), v.dtype.itemsize, i.ctypes.data, len(i))

    def _instance_desc(self, inst: InstanceData):
        d = rt64.INSTANCE_DESC()
        d.mesh = self.meshes[inst.mesh]
        d.transform = rt64.MATRIX4.from_rows(inst.transform); d.previousTransform = rt64.MATRIX4.from_rows(inst.previous_transform)
        d.diffuseTexture = self.textures[inst.diffuse]
        d.normalTexture = self.textures[inst.normal] if inst.normal is not None else None
        d.specularTexture = self.textures[inst.specular] if inst.specular is not None else None
        d.shader = self.shader
        d.material = inst.material
        d.flags = inst.flags
        if inst.scissor:
            d.scissorRect = rt64.RECT(*[int(v) for v in inst.scissor])
        if inst.viewport:
            d.viewportRect = rt64.RECT(*[int(v) for v in inst.viewport])
        return d

    def set_instance(self, k, inst: InstanceData):
        self._desc_cache.pop(k, None)
        self.lib.SetInstanceDescription(self.instances[k], self._instance_desc(inst))

    def set_view_description(self, di_samples=0, gi_samples=0, max_lights=12, denoiser=False, resolution_scale=1.0, motion_blur=0.0, upscaler=0, upscaler_mode=0):
        v = rt64.VIEW_DESC()
        v.resolutionScale = resolution_scale; v.motionBlurStrength = motion_blur
        v.diSamples = di_samples; v.giSamples = gi_samples; v.maxLights = max_lights
        v.upscaler = upscaler; v.upscalerMode = upscaler_mode; v.upscalerSharpness = 0.0; v.denoiserEnabled = denoiser
        self.view_desc = v
        self.lib.SetViewDescription(self.view, v)

    def draw(self, can_reproject=True):
        """One WM_PAINT of main.cpp:97-134: SetViewPerspective, SetInstanceDescription of the sphere, SetSceneLights, DrawDevice.
        The ctypes descriptors are rebuilt only when the scene description object they came from changed (host-side overhead
        of the harness, not of the library)."""
        d = self.data
        if getattr(self, "_view_src", None) is not d.view:
            self._view_src, self._view_m = d.view, rt64.MATRIX4.from_rows(d.view)
        self.lib.SetViewPerspective(self.view, self._view_m, d.fov, d.near, d.far, can_reproject)
        k = getattr(self, "_sphere_k", -2)
        if k == -2:
            k = self._sphere_k = next((i for i, inst in enumerate(d.instances) if inst.name == "sphere"), -1)
        if k >= 0:
            if self._desc_cache.get(k, (None,))[0] is not d.instances[k]:
                self._desc_cache[k] = (d.instances[k], self._instance_desc(d.instances[k]))
            self.lib.SetInstanceDescription(self.instances[k], self._desc_cache[k][1])                # main.cpp:129
        self.lib.SetSceneLights(self.scene, self._lights, len(d.lights))
        self.lib.DrawDevice(self.device, 1, 1000.0 / 60.0)

    def readback(self, image):
        dt, ch = rt64.IMAGE_FORMATS[image]
        st = rt64.FRAME_STATS(); st.structSize = C.sizeof(rt64.FRAME_STATS)
        self.lib.GetDeviceStats(self.device, C.byref(st))
        rows, width = (st.rowsRendered if st.rowsRendered else st.tileY1 - st.tileY0), st.width
        if (st.screenWidth, st.screenHeight) != (st.width, st.height):       # resolutionScale: render size != back-buffer size, whole frame
            rows, width = (st.screenHeight, st.screenWidth) if image in (rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_BACKGROUND) else (st.height, st.width)
        if image in (rt64.IMAGE_BACKGROUND, rt64.IMAGE_UPSCALED):
            rows, width = st.screenHeight, st.screenWidth                      # always the whole screen, on every device
        out = np.empty((rows, width, ch), dtype=dt)
        n = self.lib.ReadbackDevice(self.device, image, out.ctypes.data, out.nbytes)
        if n != out.nbytes:
            raise RuntimeError(f"RT64_ReadbackDevice(image={image}) returned {n}, expected {out.nbytes}: " + self.lib.last_error())
        return out[..., 0] if ch == 1 else out

    def stats(self):
        st = rt64.FRAME_STATS(); st.structSize = C.sizeof(rt64.FRAME_STATS)
        self.lib.GetDeviceStats(self.device, C.byref(st))
        return st

    def set_interleave(self, rank, count):
        self.lib.SetDeviceInterleave(self.device, rank, count)

    def set_tile(self, y0, y1):
        """Contiguous band of rows [y0, y1) (frames with GI + denoiser partition this way: the library adds the filter's halo)."""
        self.lib.SetDeviceTile(self.device, int(y0), int(y1))

    def option(self, key, value):
        return self.lib.SetDeviceOption(self.device, key.encode(), float(value))

    def close(self):
        if self.device:
            for h in self.meshes:
                self.lib.DestroyMesh(h)
            for h in self.textures:
                self.lib.DestroyTexture(h)
            self.lib.DestroyShader(self.shader)
            self.lib.DestroyDevice(self.device)        # deletes scenes -> views + instances (rt64_device.cpp:97-100)
            self.device = None
