#!/usr/bin/env python3
"""Per-frame cost of a scene whose meshes are all re-sent every frame (how the SM64 host drives RT64): tools/dynamic_meshes.py [meshes] [tris]
Reports ms per frame for the SetMesh calls and for RT64_DrawDevice (which builds the recorded BLASes in one batch)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene

M = int(sys.argv[1]) if len(sys.argv) > 1 else 300
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W, H, K = 1920, 1080, 60
rng = np.random.default_rng(1)
data = sample_scene.make_sample_scene()
data.meshes = list(data.meshes); data.instances = list(data.instances)
for k in range(M):
    v = np.zeros(3 * T, dtype=sample_scene.VERTEX_DTYPE)
    c = np.array([-9.0 + 18.0 * (k % 20) / 19.0, 0.4, -6.0 + 10.0 * (k // 20) / max(1, M // 20)]) + rng.normal(0, 0.2, size=(T, 1, 3))
    p = c + rng.normal(0, 0.08, size=(T, 3, 3))
    v["position"][:, :3] = p.reshape(-1, 3).astype(np.float32); v["position"][:, 3] = 1.0
    v["normal"] = (0.0, 1.0, 0.0); v["input1"] = 1.0
    data.meshes.append(sample_scene.MeshData("dyn%d" % k, rt64.MESH_RAYTRACE_ENABLED | rt64.MESH_RAYTRACE_UPDATABLE, v, np.arange(3 * T, dtype=np.uint32)))
    inst = copy.copy(data.instances[1]); inst.mesh = len(data.meshes) - 1; inst.material = sample_scene.copy_material(data.instances[1].material); inst.name = "dyn%d" % k
    data.instances.append(inst)
lib = rt64.Library()
scene = sample_scene.Rt64Scene(lib, data, W, H, hip_device=0)
first = len(data.meshes) - M
for _ in range(5):
    scene.draw()
for mode, flagmask in (("refit (UPDATABLE, same shape)", None), ("rebuild (vertex count changes)", 3)):
    t_set = t_draw = 0.0
    for f in range(K):
        t0 = time.perf_counter()
        for k in range(first, len(data.meshes)):
            m = data.meshes[k]
            n = len(m.vertices) - (flagmask * (f % 2) if flagmask else 0)
            scene.set_mesh(scene.meshes[k], m.vertices[:n], m.indices[:n])
        t1 = time.perf_counter()
        scene.draw()
        t2 = time.perf_counter()
        t_set += t1 - t0; t_draw += t2 - t1
    st = scene.stats()
    print("%d meshes x %d triangles, %s: SetMesh calls %.3f ms/frame, DrawDevice %.3f ms/frame (GPU %.3f ms: builds %.3f, trace %.3f, shade %.3f, direct %.3f)" % (
        M, T, mode, t_set * 1e3 / K, t_draw * 1e3 / K, st.msTotal, st.msBuild, st.msPrimaryTrace, st.msPrimaryShade, st.msDirect))
scene.option("count_traversal", 1); scene.draw(); st = scene.stats()
print("static frame: nodes/primary ray %.2f, tris/primary ray %.2f, nodes/shadow ray %.2f; lean %d" % (st.nodesPrimary / max(st.primaryRays, 1), st.trianglesPrimary / max(st.primaryRays, 1), st.nodesDirect / max(st.shadowRays, 1), st.leanFrame))
scene.option("count_traversal", 0)
for _ in range(3): scene.draw()
st = scene.stats()
print("static frame (tables cached): GPU %.3f ms: build %.3f trace %.3f shade %.3f direct %.3f" % (st.msTotal, st.msBuild, st.msPrimaryTrace, st.msPrimaryShade, st.msDirect))
scene.close()
