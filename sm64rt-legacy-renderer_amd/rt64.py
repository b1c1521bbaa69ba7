"""ctypes mirror of include/rt64.h -- the RT64 C-ABI function-pointer table, bound the way a host binds it.

The product is librt64.so (HIP); this module is only the Python-side harness used by tests/, bench.py and
__graft_entry__.py to drive it through the same 33 exported symbols a C host resolves with dlsym
(/root/reference/src/rt64lib/public/rt64.h:305-402), plus the additive RT64_* extensions.
It contains no rendering code and never falls back to a CPU path: if librt64.so is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "librt64.so")

# ---- constants (include/rt64.h) -----------------------------------------------------------------------------
MESH_RAYTRACE_ENABLED, MESH_RAYTRACE_UPDATABLE, MESH_RAYTRACE_FAST_TRACE, MESH_RAYTRACE_COMPACT = 0x1, 0x2, 0x4, 0x8
SHADER_FILTER_POINT, SHADER_FILTER_LINEAR = 0, 1
SHADER_ADDRESSING_WRAP, SHADER_ADDRESSING_MIRROR, SHADER_ADDRESSING_CLAMP = 0, 1, 2
SHADER_RASTER_ENABLED, SHADER_RAYTRACE_ENABLED, SHADER_NORMAL_MAP_ENABLED, SHADER_SPECULAR_MAP_ENABLED = 0x1, 0x2, 0x4, 0x8
INSTANCE_RASTER_BACKGROUND, INSTANCE_DISABLE_BACKFACE_CULLING = 0x1, 0x2
LIGHT_GROUP_MASK_ALL, LIGHT_GROUP_DEFAULT = 0xFFFFFFFF, 0x1
TEXTURE_FORMAT_RGBA8, TEXTURE_FORMAT_DDS = 0x1, 0x2
UPSCALER_OFF, UPSCALER_AUTO, UPSCALER_DLSS, UPSCALER_FSR, UPSCALER_XESS = range(5)
(UPSCALER_MODE_AUTO, UPSCALER_MODE_ULTRA_PERFORMANCE, UPSCALER_MODE_PERFORMANCE, UPSCALER_MODE_BALANCED, UPSCALER_MODE_QUALITY,
 UPSCALER_MODE_ULTRA_QUALITY, UPSCALER_MODE_NATIVE) = range(7)
ACCEL_NODES, ACCEL_TRIANGLES, ACCEL_SORTED_INDEX, ACCEL_MORTON, ACCEL_HEADER, ACCEL_HOST_DEPTH = range(6)

(IMAGE_FINAL_RGBA8, IMAGE_SHADING_POSITION, IMAGE_SHADING_NORMAL, IMAGE_SHADING_SPECULAR, IMAGE_DIFFUSE,
 IMAGE_INSTANCE_ID, IMAGE_DIRECT_LIGHT_RAW, IMAGE_DIRECT_LIGHT_FILTERED, IMAGE_INDIRECT_LIGHT_RAW,
 IMAGE_INDIRECT_LIGHT_FILTERED, IMAGE_REFLECTION, IMAGE_REFRACTION, IMAGE_TRANSPARENT, IMAGE_FLOW,
 IMAGE_REACTIVE_MASK, IMAGE_LOCK_MASK, IMAGE_DEPTH, IMAGE_OUTPUT_RGBA32F, IMAGE_PRIMARY_HIT,
 IMAGE_VIEW_DIRECTION, IMAGE_FIRST_INSTANCE_ID, IMAGE_BACKGROUND, IMAGE_UPSCALED) = range(23)

# image id -> (numpy dtype string, channels)
IMAGE_FORMATS = {
    IMAGE_FINAL_RGBA8: ("u1", 4), IMAGE_SHADING_POSITION: ("f4", 4), IMAGE_SHADING_NORMAL: ("f4", 4),
    IMAGE_SHADING_SPECULAR: ("f4", 4), IMAGE_DIFFUSE: ("f4", 4), IMAGE_INSTANCE_ID: ("i4", 1),
    IMAGE_DIRECT_LIGHT_RAW: ("f4", 4), IMAGE_DIRECT_LIGHT_FILTERED: ("f4", 4), IMAGE_INDIRECT_LIGHT_RAW: ("f4", 4),
    IMAGE_INDIRECT_LIGHT_FILTERED: ("f4", 4), IMAGE_REFLECTION: ("f4", 4), IMAGE_REFRACTION: ("f4", 4),
    IMAGE_TRANSPARENT: ("f4", 4), IMAGE_FLOW: ("f4", 2), IMAGE_REACTIVE_MASK: ("f4", 1), IMAGE_LOCK_MASK: ("f4", 1),
    IMAGE_DEPTH: ("f4", 1), IMAGE_OUTPUT_RGBA32F: ("f4", 4), IMAGE_PRIMARY_HIT: ("u4", 4),
    IMAGE_VIEW_DIRECTION: ("f4", 4), IMAGE_FIRST_INSTANCE_ID: ("i4", 1), IMAGE_BACKGROUND: ("u1", 4), IMAGE_UPSCALED: ("f4", 4),
}


# ---- POD structs ------------------------------------------------------------------------------------------------
class VECTOR2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class VECTOR3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z)


class VECTOR4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class MATRIX4(C.Structure):
    _fields_ = [("m", (C.c_float * 4) * 4)]

    @staticmethod
    def from_rows(rows):
        m = MATRIX4()
        for r in range(4):
            for c in range(4):
                m.m[r][c] = float(rows[r][c])
        return m

    @staticmethod
    def identity():
        return MATRIX4.from_rows([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])


class RECT(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("w", C.c_int), ("h", C.c_int)]


class MATERIAL(C.Structure):
    _fields_ = [("diffuseTexIndex", C.c_int), ("normalTexIndex", C.c_int), ("specularTexIndex", C.c_int),
                ("ignoreNormalFactor", C.c_float), ("uvDetailScale", C.c_float), ("reflectionFactor", C.c_float),
                ("reflectionFresnelFactor", C.c_float), ("reflectionShineFactor", C.c_float), ("refractionFactor", C.c_float),
                ("specularColor", VECTOR3), ("specularExponent", C.c_float), ("solidAlphaMultiplier", C.c_float),
                ("shadowAlphaMultiplier", C.c_float), ("depthBias", C.c_float), ("shadowRayBias", C.c_float),
                ("selfLight", VECTOR3), ("lightGroupMaskBits", C.c_uint), ("fogColor", VECTOR3),
                ("diffuseColorMix", VECTOR4), ("fogMul", C.c_float), ("fogOffset", C.c_float), ("fogEnabled", C.c_uint),
                ("lockMask", C.c_float), ("enabledAttributes", C.c_int)]


class LIGHT(C.Structure):
    _fields_ = [("position", VECTOR3), ("diffuseColor", VECTOR3), ("attenuationRadius", C.c_float), ("pointRadius", C.c_float),
                ("specularColor", VECTOR3), ("shadowOffset", C.c_float), ("attenuationExponent", C.c_float),
                ("flickerIntensity", C.c_float), ("groupBits", C.c_uint)]


class SCENE_DESC(C.Structure):
    _fields_ = [("ambientBaseColor", VECTOR3), ("ambientNoGIColor", VECTOR3), ("eyeLightDiffuseColor", VECTOR3),
                ("eyeLightSpecularColor", VECTOR3), ("skyDiffuseMultiplier", VECTOR3), ("skyHSLModifier", VECTOR3),
                ("skyYawOffset", C.c_float), ("giDiffuseStrength", C.c_float), ("giSkyStrength", C.c_float)]


class VIEW_DESC(C.Structure):
    _fields_ = [("resolutionScale", C.c_float), ("motionBlurStrength", C.c_float), ("diSamples", C.c_uint),
                ("giSamples", C.c_uint), ("maxLights", C.c_uint), ("upscaler", C.c_ubyte), ("upscalerMode", C.c_ubyte),
                ("upscalerSharpness", C.c_float), ("denoiserEnabled", C.c_bool)]


class INSTANCE_DESC(C.Structure):
    _fields_ = [("mesh", C.c_void_p), ("transform", MATRIX4), ("previousTransform", MATRIX4),
                ("diffuseTexture", C.c_void_p), ("normalTexture", C.c_void_p), ("specularTexture", C.c_void_p),
                ("shader", C.c_void_p), ("material", MATERIAL), ("scissorRect", RECT), ("viewportRect", RECT),
                ("flags", C.c_uint)]


class TEXTURE_DESC(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("byteCount", C.c_int), ("format", C.c_int), ("width", C.c_int),
                ("height", C.c_int), ("rowPitch", C.c_int)]


class FRAME_STATS(C.Structure):
    _fields_ = [("structSize", C.c_uint), ("width", C.c_uint), ("height", C.c_uint), ("tileY0", C.c_uint), ("tileY1", C.c_uint),
                ("primaryRays", C.c_ulonglong), ("shadowRays", C.c_ulonglong), ("indirectRays", C.c_ulonglong),
                ("reflectionRays", C.c_ulonglong), ("refractionRays", C.c_ulonglong),
                ("nodesVisited", C.c_ulonglong), ("trianglesTested", C.c_ulonglong),
                ("msTotal", C.c_float), ("msBuild", C.c_float), ("msPrimary", C.c_float), ("msDirect", C.c_float),
                ("msIndirect", C.c_float), ("msReflectRefract", C.c_float), ("msDenoise", C.c_float), ("msComposePost", C.c_float),
                ("msHostWall", C.c_float),
                ("blasNodeBytes", C.c_uint), ("blasTriangleBytes", C.c_uint), ("tlasNodeBytes", C.c_uint),
                ("instanceCount", C.c_uint), ("triangleCount", C.c_uint),
                ("msPrimaryTrace", C.c_float), ("msPrimaryShade", C.c_float),
                ("stripRank", C.c_uint), ("stripCount", C.c_uint), ("rowsRendered", C.c_uint), ("leanFrame", C.c_uint),
                ("nodesPrimary", C.c_ulonglong), ("trianglesPrimary", C.c_ulonglong), ("nodesDirect", C.c_ulonglong),
                ("trianglesDirect", C.c_ulonglong), ("nodesIndirect", C.c_ulonglong), ("trianglesIndirect", C.c_ulonglong),
                ("screenWidth", C.c_uint), ("screenHeight", C.c_uint), ("accumFrames", C.c_uint),
                ("accumMsTotal", C.c_float), ("accumMsBuild", C.c_float), ("accumMsPrimaryTrace", C.c_float), ("accumMsPrimaryShade", C.c_float),
                ("accumMsDirect", C.c_float), ("accumMsIndirect", C.c_float), ("accumMsReflectRefract", C.c_float), ("accumMsDenoise", C.c_float),
                ("accumMsComposePost", C.c_float), ("fusedFrame", C.c_uint), ("packedFinal", C.c_uint),
                ("overlappedFrame", C.c_uint), ("reflectionBesideDenoiser", C.c_uint), ("traversalOverflow", C.c_uint)]


assert C.sizeof(MATERIAL) == 132 and C.sizeof(LIGHT) == 60 and C.sizeof(SCENE_DESC) == 84
assert C.sizeof(VIEW_DESC) == 32 and C.sizeof(INSTANCE_DESC) == 336 and C.sizeof(TEXTURE_DESC) == 32

_P = C.c_void_p
# (member, symbol, restype, argtypes) in RT64_LIBRARY member order (include/rt64.h RT64_API_LIST)
API = [
    ("GetLastError", "RT64_GetLastError", C.c_char_p, []),
    ("CreateDevice", "RT64_CreateDevice", _P, [_P]),
    ("DestroyDevice", "RT64_DestroyDevice", None, [_P]),
    ("DrawDevice", "RT64_DrawDevice", None, [_P, C.c_int, C.c_float]),
    ("CreateView", "RT64_CreateView", _P, [_P]),
    ("SetViewPerspective", "RT64_SetViewPerspective", None, [_P, MATRIX4, C.c_float, C.c_float, C.c_float, C.c_bool]),
    ("SetViewDescription", "RT64_SetViewDescription", None, [_P, VIEW_DESC]),
    ("SetViewSkyPlane", "RT64_SetViewSkyPlane", None, [_P, _P]),
    ("GetViewRaytracedInstanceAt", "RT64_GetViewRaytracedInstanceAt", _P, [_P, C.c_int, C.c_int]),
    ("GetViewUpscalerSupport", "RT64_GetViewUpscalerSupport", C.c_bool, [_P, C.c_char]),
    ("DestroyView", "RT64_DestroyView", None, [_P]),
    ("CreateScene", "RT64_CreateScene", _P, [_P]),
    ("SetSceneDescription", "RT64_SetSceneDescription", None, [_P, SCENE_DESC]),
    ("SetSceneLights", "RT64_SetSceneLights", None, [_P, C.POINTER(LIGHT), C.c_int]),
    ("DestroyScene", "RT64_DestroyScene", None, [_P]),
    ("CreateMesh", "RT64_CreateMesh", _P, [_P, C.c_int]),
    ("SetMesh", "RT64_SetMesh", None, [_P, _P, C.c_int, C.c_int, _P, C.c_int]),
    ("DestroyMesh", "RT64_DestroyMesh", None, [_P]),
    ("CreateShader", "RT64_CreateShader", _P, [_P, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int]),
    ("DestroyShader", "RT64_DestroyShader", None, [_P]),
    ("CreateInstance", "RT64_CreateInstance", _P, [_P]),
    ("SetInstanceDescription", "RT64_SetInstanceDescription", None, [_P, INSTANCE_DESC]),
    ("DestroyInstance", "RT64_DestroyInstance", None, [_P]),
    ("CreateTexture", "RT64_CreateTexture", _P, [_P, TEXTURE_DESC]),
    ("DestroyTexture", "RT64_DestroyTexture", None, [_P]),
    ("CreateInspector", "RT64_CreateInspector", _P, [_P]),
    ("HandleMessageInspector", "RT64_HandleMessageInspector", C.c_bool, [_P, C.c_uint, C.c_size_t, C.c_ssize_t]),
    ("PrintClearInspector", "RT64_PrintClearInspector", None, [_P]),
    ("PrintMessageInspector", "RT64_PrintMessageInspector", None, [_P, C.c_char_p]),
    ("SetSceneInspector", "RT64_SetSceneInspector", None, [_P, C.POINTER(SCENE_DESC)]),
    ("SetMaterialInspector", "RT64_SetMaterialInspector", None, [_P, C.POINTER(MATERIAL), C.c_char_p]),
    ("SetLightsInspector", "RT64_SetLightsInspector", None, [_P, C.POINTER(LIGHT), C.POINTER(C.c_int), C.c_int]),
    ("DestroyInspector", "RT64_DestroyInspector", None, [_P]),
]

EXT_API = [
    ("CreateDeviceHeadless", "RT64_CreateDeviceHeadless", _P, [C.c_int, C.c_int, C.c_int]),
    ("SetDeviceSize", "RT64_SetDeviceSize", None, [_P, C.c_int, C.c_int]),
    ("SetDeviceTile", "RT64_SetDeviceTile", None, [_P, C.c_int, C.c_int]),
    ("SetDeviceInterleave", "RT64_SetDeviceInterleave", None, [_P, C.c_int, C.c_int]),
    ("ReadbackDevice", "RT64_ReadbackDevice", C.c_size_t, [_P, C.c_int, _P, C.c_size_t]),
    ("CopyDeviceImage", "RT64_CopyDeviceImage", C.c_size_t, [_P, C.c_int, _P, C.c_size_t]),
    ("GetDeviceStats", "RT64_GetDeviceStats", C.c_int, [_P, C.POINTER(FRAME_STATS)]),
    ("SetDeviceOption", "RT64_SetDeviceOption", C.c_int, [_P, C.c_char_p, C.c_double]),
    ("ReadbackTileTiming", "RT64_ReadbackTileTiming", C.c_size_t, [_P, _P, C.c_size_t]),
    ("GetDeviceStream", "RT64_GetDeviceStream", _P, [_P]),
    ("SetDeviceGatherTarget", "RT64_SetDeviceGatherTarget", None, [_P, _P, C.c_size_t]),
    ("ReadbackMeshAccel", "RT64_ReadbackMeshAccel", C.c_size_t, [_P, C.c_int, _P, C.c_size_t]),
    ("ReadbackTexture", "RT64_ReadbackTexture", C.c_size_t, [_P, C.c_int, _P, C.c_size_t]),
    ("ReadbackViewAccel", "RT64_ReadbackViewAccel", C.c_size_t, [_P, C.c_int, _P, C.c_size_t]),
    ("MeshTreeDepth", "RT64_MeshTreeDepth", C.c_uint, [_P, C.c_int, C.c_int, _P, C.c_int]),
    ("GetGatherUniqueId", "RT64_GetGatherUniqueId", C.c_int, [_P, C.c_size_t]),
    ("CreateGather", "RT64_CreateGather", _P, [_P, _P, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    ("SubmitGather", "RT64_SubmitGather", C.c_int, [_P]),
    ("ReadbackGather", "RT64_ReadbackGather", C.c_size_t, [_P, C.c_int, _P, C.c_size_t, C.c_int]),
    ("GetGatherFrame", "RT64_GetGatherFrame", _P, [_P, C.c_int]),
    ("DestroyGather", "RT64_DestroyGather", None, [_P]),
    ("GetGatherDirectHandle", "RT64_GetGatherDirectHandle", C.c_size_t, [_P, _P, C.c_size_t]),
    ("SetGatherDirect", "RT64_SetGatherDirect", C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    ("GatherRowOwner", "RT64_GatherRowOwner", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    ("GatherOwnedRows", "RT64_GatherOwnedRows", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("GatherSlotRows", "RT64_GatherSlotRows", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("GatherRowOwnerOf", "RT64_GatherRowOwnerOf", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    ("GatherOwnedRowsOf", "RT64_GatherOwnedRowsOf", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    ("GatherSlotRowsOf", "RT64_GatherSlotRowsOf", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    ("GetGatherBands", "RT64_GetGatherBands", C.c_int, [_P, C.POINTER(C.c_int), C.c_int]),
    ("BalanceGatherBands", "RT64_BalanceGatherBands", None, [C.POINTER(C.c_uint), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    ("RebalanceGatherBands", "RT64_RebalanceGatherBands", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    ("SetGatherBands", "RT64_SetGatherBands", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("HaloPlan", "RT64_HaloPlan", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, _P, C.c_int]),
    ("SetDeviceHaloExchange", "RT64_SetDeviceHaloExchange", C.c_int, [_P, _P, _P, C.POINTER(C.c_int), C.c_int, C.c_int]),
]
HALO_ROWS = 62
HALO_BYTES_PER_PIXEL = 24


class HALO_REGION(C.Structure):
    _fields_ = [("peer", C.c_int), ("send", C.c_int), ("y0", C.c_int), ("y1", C.c_int), ("host", C.c_void_p), ("bytes", C.c_size_t)]


HALO_EXCHANGE = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(HALO_REGION), C.c_int)
GATHER_ID_BYTES = 128


class Library:
    """RT64_LoadLibrary(): dlopen + one dlsym per table member.  Members are attributes (lib.CreateDevice ...)."""

    def __init__(self, path=None):
        path = path or os.environ.get("RT64_LIBRARY_PATH") or DEFAULT_LIBRARY
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: the HIP library is not built (run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback.")
        self.path = path
        self.handle = C.CDLL(path, mode=C.RTLD_LOCAL)
        for member, symbol, restype, argtypes in API + EXT_API:
            fn = getattr(self.handle, symbol)      # AttributeError if an export is missing
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, member, fn)

    def last_error(self):
        e = self.GetLastError()
        return e.decode() if e else ""


def exported_symbols():
    """Every symbol include/rt64.h declares (33 reference exports + extensions)."""
    return [s for _, s, _, _ in API + EXT_API]
