"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports every declared symbol."""
import ctypes as C
import os
import re
import subprocess

import pytest

import __graft_entry__ as graft

ROOT = graft.ROOT


@pytest.fixture(scope="module")
def built():
    lib = os.path.join(graft.PKG_DIR, "librt64.so")
    if not os.path.exists(lib):
        graft.build()
    return lib


def test_library_exports_every_declared_symbol(built):
    from sm64rt_legacy_renderer_amd import rt64
    h = C.CDLL(built, mode=C.RTLD_LOCAL)
    header = open(os.path.join(ROOT, "include", "rt64.h")).read()
    declared = sorted(set(re.findall(r"X\(\w+,\s*(RT64_\w+),", header)))
    assert len(declared) == 33 + 35, declared          # the reference's 33 + the additive exports (15 headless / readback / stats / profiling + 18 multi-GPU gather + 2 halo exchange)
    assert sorted(rt64.exported_symbols()) == declared
    for name in declared:
        assert hasattr(h, name), name


def test_reference_export_names(built):
    """The 33 names the reference's loader resolves (public/rt64.h:358-392) are all present, unmangled."""
    out = subprocess.run(["nm", "-D", "--defined-only", built], stdout=subprocess.PIPE, text=True, check=True).stdout
    names = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    expected = """RT64_GetLastError RT64_CreateDevice RT64_DestroyDevice RT64_DrawDevice RT64_CreateView RT64_SetViewPerspective
    RT64_SetViewDescription RT64_SetViewSkyPlane RT64_GetViewRaytracedInstanceAt RT64_GetViewUpscalerSupport RT64_DestroyView
    RT64_CreateScene RT64_SetSceneDescription RT64_SetSceneLights RT64_DestroyScene RT64_CreateMesh RT64_SetMesh RT64_DestroyMesh
    RT64_CreateShader RT64_DestroyShader RT64_CreateInstance RT64_SetInstanceDescription RT64_DestroyInstance RT64_CreateTexture
    RT64_DestroyTexture RT64_CreateInspector RT64_HandleMessageInspector RT64_SetSceneInspector RT64_SetMaterialInspector
    RT64_SetLightsInspector RT64_PrintClearInspector RT64_PrintMessageInspector RT64_DestroyInspector""".split()
    assert len(expected) == 33
    assert not [n for n in expected if n not in names]
    # nothing but the ABI leaks out of the library
    leaked = [n for n in names if not n.startswith("RT64_")]
    assert not leaked, leaked[:10]


def test_header_compiles_as_c_and_cpp_and_layouts_match(tmp_path):
    src = tmp_path / "abi.c"
    src.write_text('#include "rt64.h"\n#include <stdio.h>\nint main(void){ printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(RT64_MATERIAL), sizeof(RT64_LIGHT),'
                   ' sizeof(RT64_SCENE_DESC), sizeof(RT64_VIEW_DESC), sizeof(RT64_INSTANCE_DESC), sizeof(RT64_TEXTURE_DESC), sizeof(RT64_LIBRARY)); return 0; }\n')
    for cc, std, ext in (("gcc", "-std=c11", "c"), ("g++", "-std=c++17", "cpp")):
        exe = tmp_path / ("abi_" + ext)
        s = tmp_path / ("abi2." + ext)
        s.write_text(src.read_text())
        subprocess.run([cc, std, "-I", os.path.join(ROOT, "include"), str(s), "-o", str(exe), "-ldl"], check=True)
        out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
        assert out == ["132", "60", "84", "32", "336", "32", str(8 + 33 * 8)], out


def test_c_host_loads_table_through_dlopen(built, tmp_path):
    """A C host resolves the whole RT64_LIBRARY table with RT64_LoadLibrary(); without a GPU CreateDevice returns NULL
    and RT64_GetLastError() explains why (no CPU fallback)."""
    src = tmp_path / "host.c"
    src.write_text(r'''
#include "rt64.h"
int main(void) {
    RT64_LIBRARY lib = RT64_LoadLibrary();
    if (!lib.handle) return 2;
    void **members = (void **)&lib;
    for (unsigned i = 1; i < sizeof(lib) / sizeof(void *); i++) if (!members[i]) { printf("missing member %u\n", i); return 3; }
    RT64_LIBRARY_EXT ext = RT64_LoadLibraryExt(lib);
    if (!ext.CreateDeviceHeadless || !ext.ReadbackDevice || !ext.SetDeviceTile || !ext.GetDeviceStats) return 4;
    RT64_DEVICE *dev = lib.CreateDevice(0);
    if (!dev) { printf("no device: %s\n", lib.GetLastError()); RT64_UnloadLibrary(lib); return 0; }
    lib.DestroyDevice(dev);
    RT64_UnloadLibrary(lib);
    printf("device ok\n");
    return 0;
}
''')
    exe = tmp_path / "host"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-ldl"], check=True)
    env = dict(os.environ, RT64_LIBRARY_PATH=built, RT64_WIDTH="64", RT64_HEIGHT="64")
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    assert r.returncode == 0, r.stdout
    assert "device ok" in r.stdout or "no device" in r.stdout, r.stdout


def _oracle_tree_depth(bvh):
    """Inner nodes on the longest root-to-leaf path of an oracle LBVH (what lbvh.hip leaves in BlasHeader::depth)."""
    nodes, n = bvh["nodes"], bvh["count"]
    if n <= 1:
        return 1
    depth, todo = 0, [(0, 1)]
    while todo:
        i, d = todo.pop()
        depth = max(depth, d)
        for c in (int(nodes["left"][i]), int(nodes["right"][i])):
            if not (c & 0x80000000):
                todo.append((c, d + 1))
    return depth


def test_host_side_tree_depth_equals_the_depth_of_the_oracle_lbvh(built, sample_data):
    """RT64_SetMesh sizes the traversal stacks from a depth it computes on the host (no wait for the device build).  RT64_MeshTreeDepth is
    that computation as a pure function: it has to be the depth of the tree the builders make -- the oracle's LBVH, which the device
    builder reproduces bit for bit (tests/test_gpu_parity.py) -- on the sample meshes, random soups, duplicates and degenerate inputs."""
    import numpy as np
    from sm64rt_legacy_renderer_amd import rt64
    from oracle import oracle_py
    lib = rt64.Library(built)
    L = oracle_py.lib()
    rng = np.random.default_rng(11)

    def check(verts, idx, stride=None):
        v = np.ascontiguousarray(verts); i = np.ascontiguousarray(idx, dtype=np.uint32)
        stride = stride or v.dtype.itemsize * (v.shape[1] if v.ndim == 2 else 1)
        m = L.oracle_mesh_create(1)
        try:
            L.oracle_mesh_set(m, v.ctypes.data, len(v), stride, i.ctypes.data, len(i))
            want = _oracle_tree_depth(oracle_py.bvh_to_numpy(L.oracle_mesh_bvh(m)))
        finally:
            L.oracle_mesh_destroy(m)
        got = lib.MeshTreeDepth(v.ctypes.data, len(v), stride, i.ctypes.data, len(i))
        assert got == want, (got, want, len(i) // 3)
        return got

    for m in sample_data.meshes:
        if m.flags & rt64.MESH_RAYTRACE_ENABLED:
            check(m.vertices, m.indices, m.vertices.dtype.itemsize)
    for n in (1, 2, 3, 17, 320, 1500, 4096):
        p = rng.normal(size=(3 * n, 3)).astype(np.float32) * np.float32(5.0)
        check(p, np.arange(3 * n), 12)
    p = np.tile(rng.normal(size=(3, 3)).astype(np.float32), (64, 1))            # 64 copies of one triangle: equal codes, the leaf number decides
    assert check(p, np.arange(192), 12) >= 6
    p = np.zeros((300, 3), dtype=np.float32)                                      # every triangle degenerate at the origin (extent 0 on all axes)
    check(p, np.arange(300), 12)
    p = rng.normal(size=(600, 3)).astype(np.float32); p[:, 1] = 0.0              # flat soup: one axis without extent
    check(p, rng.integers(0, 600, size=900), 12)
    big = rng.normal(size=(3 * 4097, 3)).astype(np.float32)                       # above the single-workgroup builder: "deep" without looking
    assert lib.MeshTreeDepth(big.ctypes.data, len(big), 12, np.arange(3 * 4097, dtype=np.uint32).ctypes.data, 3 * 4097) == 255


def test_bench_configurations_are_consistent():
    """sample_scene.BENCH_CONFIGS is what bench.py, tools/band_costs.py and tests/test_gpu_configs.py all read: every configuration bench.py offers exists, the ones
    that run BASELINE.json's wording through the library's extensions say which, every note belongs to a configuration, and the option keys they set are keys
    RT64_SetDeviceOption knows (looked up in the host source: the call itself needs a device)."""
    import os
    import re
    from sm64rt_legacy_renderer_amd import sample_scene
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = open(os.path.join(root, "bench.py")).read()
    choices = re.search(r'"--config", default="C2", choices=\[([^\]]*)\]', bench).group(1)
    offered = set(re.findall(r'"([^"]+)"', choices))
    assert offered == set(sample_scene.BENCH_CONFIGS) and "C2" in offered
    host = open(os.path.join(root, "sm64rt-legacy-renderer_amd", "csrc", "rt64_host.cpp")).read()
    for name, cfg in sample_scene.BENCH_CONFIGS.items():
        assert {"width", "height", "gi_samples", "denoiser"} <= set(cfg)
        for key in set(cfg) - {"width", "height", "gi_samples", "denoiser"}:
            assert 'k == "%s"' % key in host, key                 # an option RT64_SetDeviceOption knows
        assert name.endswith("-literal") == bool(set(cfg) & {"primary_spp", "gi_bounces"})
    assert set(sample_scene.BENCH_DEVIATIONS) <= set(sample_scene.BENCH_CONFIGS)
    assert all(n in sample_scene.BENCH_DEVIATIONS for n in ("C4", "C5", "C4-literal", "C5-literal"))


def _macro_constants(text):
    return {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^[ \t]*#[ \t]*define[ \t]+(RT64_[A-Z0-9_]+)[ \t]+(0x[0-9A-Fa-f]+|\d+)u?[ \t]*(?:/\*.*)?$", text, re.M)}


def test_constants_are_macros_like_the_references_and_equal_the_enumerators():
    """public/rt64.h:11-86 defines its constants with #define: a host may test them with #ifdef / #if.  include/rt64.h carries each one as an enumerator and as a
    macro of the same value; with the reference at hand every numeric macro of its header has to exist here with the reference's value."""
    header = open(os.path.join(ROOT, "include", "rt64.h")).read()
    macros = _macro_constants(header)
    head = header.split("/* ---- opaque handles", 1)[0]
    enums = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"\b(RT64_[A-Z0-9_]+)\s*=\s*(0x[0-9A-Fa-f]+|\d+)", head)}
    enums.update({"RT64_ATTRIBUTE_" + m.group(1): int(m.group(2), 0) for m in re.finditer(r"X\((\w+),\s*(0x[0-9A-Fa-f]+),\s*\w+\)", head)})
    enums.pop("RT64_ATTRIBUTE_ALL_")
    assert len(enums) >= 55
    for name, value in enums.items():
        assert macros.get(name) == value, name
    ref = "/root/reference/src/rt64lib/public/rt64.h"
    if os.path.exists(ref):
        theirs = _macro_constants(open(ref).read())
        assert len(theirs) >= 55
        for name, value in theirs.items():
            assert macros.get(name) == value, (name, value, macros.get(name))
    # ... and the preprocessor sees them: #ifdef and #if arithmetic on a constant of every group
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "m.c")
        open(src, "w").write('#include "rt64.h"\n#if !defined(RT64_ATTRIBUTE_DIFFUSE_COLOR_MIX) || RT64_ATTRIBUTE_DIFFUSE_COLOR_MIX != 0x4000 || RT64_MESH_RAYTRACE_UPDATABLE != 2 || '
                             'RT64_UPSCALER_FSR != 3 || RT64_TEXTURE_FORMAT_DDS != 2 || RT64_MATERIAL_CC_SHADER_TEXEL1 != 7 || RT64_INSTANCE_DISABLE_BACKFACE_CULLING != 2\n#error constants\n#endif\n'
                             'int main(void) { return RT64_ATTRIBUTE_NONE; }\n')
        subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", src], check=True)


def _struct_members(text):
    """{struct name: [(type, member), ...]} of every `typedef struct { ... } RT64_X;` in a C header: comments dropped, `a, b` declarators split, arrays kept on the name."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{([^{}]*)\}\s*(RT64_\w+)\s*;", text):
        members = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(.+?)\s*(\**\w+(?:\[\d+\])*(?:\s*,\s*\**\w+(?:\[\d+\])*)*)$", decl)
            assert m, decl
            ctype = " ".join(m.group(1).split())
            for d in m.group(2).split(","):
                d = d.strip()
                stars = len(d) - len(d.lstrip("*"))
                members.append((ctype + " *" * stars, d.lstrip("*")))
        out[name] = members
    return out


def test_pod_structs_have_the_references_members_in_the_references_order():
    """The descriptors a host fills and passes by value (public/rt64.h:98-205): member for member the reference's types, names and order -- compared with the reference's header
    itself where it is at hand (the build container), and compiled to the sizes / offsets SURVEY 8(b) lists everywhere (test_header_compiles_as_c_and_cpp_and_layouts_match)."""
    ours = _struct_members(open(os.path.join(ROOT, "include", "rt64.h")).read())
    pods = ["RT64_VECTOR2", "RT64_VECTOR3", "RT64_VECTOR4", "RT64_MATRIX4", "RT64_RECT", "RT64_MATERIAL", "RT64_LIGHT", "RT64_SCENE_DESC", "RT64_VIEW_DESC", "RT64_INSTANCE_DESC", "RT64_TEXTURE_DESC"]
    for name in pods:
        assert name in ours and ours[name], name
    assert [m for _, m in ours["RT64_MATERIAL"]][:4] == ["diffuseTexIndex", "normalTexIndex", "specularTexIndex", "ignoreNormalFactor"] and ours["RT64_MATERIAL"][-1] == ("int", "enabledAttributes")
    assert ours["RT64_VIEW_DESC"][-1][1] == "denoiserEnabled" and ours["RT64_INSTANCE_DESC"][0] == ("RT64_MESH *", "mesh")
    ref = "/root/reference/src/rt64lib/public/rt64.h"
    if os.path.exists(ref):
        theirs = _struct_members(open(ref).read())
        for name in pods:
            assert ours[name] == theirs[name], (name, ours[name], theirs[name])
        # ... and the function table: the same member names in the same order behind the module handle (public/rt64.h:305-342)
        table = re.search(r"typedef\s+struct\s*\{([^{}]*)\}\s*RT64_LIBRARY\s*;", re.sub(r"//[^\n]*", " ", open(ref).read()))
        their_members = re.findall(r"(\w+)\s*;", re.sub(r"#\w+[^\n]*", " ", table.group(1)))          # (the #ifndef RT64_MINIMAL around the full table dropped)
        header = open(os.path.join(ROOT, "include", "rt64.h")).read()
        core = re.search(r"#define RT64_API_LIST_CORE\(X\)(.*?)\n\n", header, re.S).group(1)
        full = re.search(r"#define RT64_API_LIST_FULL\(X\)(.*?)\n\n", header, re.S).group(1)
        our_members = ["handle"] + re.findall(r"X\((\w+),\s*RT64_\w+,", core) + re.findall(r"X\((\w+),\s*RT64_\w+,", full)
        assert our_members == their_members, (our_members, their_members)
