"""Random and pathological meshes through the BLAS builders, against the oracle's builder (oracle/oracle_bvh.c, Geometry spec G1-G6) bit for bit.

The builders' own tests use the sample's meshes (an icosphere, a grid).  Here the sphere of the sample scene is replaced by a seeded triangle
soup of one of several kinds -- scattered, clustered on a few points (equal Morton keys), repeated triangles, zero-area triangles, all in one
plane or on one line (an axis without extent), far from the origin, a single triangle -- at sizes on both sides of the builders' switches (one
workgroup in LDS up to 4096 leaves, the multi-kernel radix path above).  Compared: Morton keys, sorted order, every node, the triangle records;
then the frame the scene renders: hit records and visit counters; for every second seed the mesh is UPDATABLE and is sent again with moved vertices (refit).  tools/exp/r04_fuzz_meshes.py runs the generator over hundreds of seeds."""
import copy
import random

import numpy as np
import pytest

from test_gpu_parity import _accel

pytestmark = pytest.mark.gpu

KINDS = ["scattered", "clustered", "repeated", "degenerate", "planar", "collinear", "far", "sliver", "mixed"]
SIZES = [1, 2, 3, 7, 64, 257, 1000, 4095, 4096, 4097, 6000]


def random_mesh(seed):
    """-> (positions [3n, 3] float32, kind, triangle count)"""
    rng = np.random.default_rng(seed)
    r = random.Random(seed)
    kind, n = r.choice(KINDS), r.choice(SIZES)
    c = rng.uniform(-2.0, 2.0, size=(n, 1, 3)).astype(np.float32) + np.array([0.0, 1.5, 0.0], dtype=np.float32)
    e = rng.uniform(-0.3, 0.3, size=(n, 3, 3)).astype(np.float32)
    p = c + e
    if kind == "clustered":                                   # a handful of centres: many equal keys, long runs the radix tree splits by index
        centres = rng.uniform(-2.0, 2.0, size=(max(1, n // 50 + 1), 3)).astype(np.float32) + np.array([0.0, 1.5, 0.0], dtype=np.float32)
        p = centres[rng.integers(0, len(centres), size=n)][:, None, :] + (e * np.float32(1e-4))
    elif kind == "repeated":                                  # every triangle several times over
        m = max(1, n // 4)
        p = p[rng.integers(0, m, size=n)]
    elif kind == "degenerate":                                # zero-area triangles among ordinary ones (two equal corners, or three)
        k = rng.random(n)
        p[k < 0.3, 1] = p[k < 0.3, 0]
        p[k < 0.1, 2] = p[k < 0.1, 0]
    elif kind == "planar":
        p[..., r.choice([0, 1, 2])] = np.float32(r.uniform(0.5, 2.0))
    elif kind == "collinear":                                 # every vertex on one line: two axes without extent
        t = rng.uniform(-2.0, 2.0, size=(n, 3, 1)).astype(np.float32)
        p = np.array([0.0, 1.0, 0.0], dtype=np.float32) + t * np.array([1.0, 0.0, 0.0], dtype=np.float32)
    elif kind == "far":                                       # large coordinates, small triangles: few mantissa bits left for the offsets
        p = p * np.float32(1e-3) + np.array([4096.0, 2048.0, -8192.0], dtype=np.float32)
    elif kind == "sliver":                                    # long thin triangles across the whole mesh
        p[:, 1] = p[:, 0] + rng.uniform(-4.0, 4.0, size=(n, 3)).astype(np.float32)
        p[:, 2] = p[:, 0] + (p[:, 1] - p[:, 0]) * np.float32(0.5) + rng.uniform(-1e-3, 1e-3, size=(n, 3)).astype(np.float32)
    elif kind == "mixed":
        k = rng.random(n)
        p[k < 0.2] = p[0]
        p[(k >= 0.2) & (k < 0.4), :, 1] = np.float32(1.0)
    return np.ascontiguousarray(p.reshape(-1, 3), dtype=np.float32), kind, n


def scene_with_mesh(sample_data, positions):
    from sm64rt_legacy_renderer_amd import sample_scene
    d = copy.copy(sample_data)
    d.meshes = [copy.copy(m) for m in sample_data.meshes]
    v = np.zeros(len(positions), dtype=sample_scene.VERTEX_DTYPE)
    v["position"][:, :3] = positions; v["position"][:, 3] = 1.0
    v["normal"] = (0.0, 1.0, 0.0); v["uv"] = positions[:, :2] * np.float32(0.25); v["input1"] = (1.0, 1.0, 1.0, 1.0)
    d.meshes[0] = sample_scene.MeshData("soup", sample_data.meshes[0].flags, v, np.arange(len(v), dtype=np.uint32))
    if np.abs(positions).max() > 100.0:                       # the "far" mesh: bring it in front of the camera with the instance transform (a BLAS is built in object space)
        d.instances = [copy.copy(i) for i in sample_data.instances]
        t = np.array(d.instances[1].transform, dtype=np.float32).copy(); t[3, :3] = t[3, :3] - positions.mean(axis=0) + np.array([0.0, 1.0, 0.0], dtype=np.float32)
        d.instances[1].transform = t; d.instances[1].previous_transform = t
    return d


def compare_mesh(rt64_lib, sample_data, seed):
    """-> list of findings"""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    positions, kind, n = random_mesh(seed)
    refit = seed % 2 == 0
    data = scene_with_mesh(sample_data, positions)
    if refit:
        data.meshes[0] = sample_scene.MeshData("soup", data.meshes[0].flags | rt64.MESH_RAYTRACE_UPDATABLE, data.meshes[0].vertices, data.meshes[0].indices)
    s = sample_scene.Rt64Scene(rt64_lib, data, 160, 90, hip_device=0)
    o = oracle_py.OracleScene(data)
    bad = []
    try:
        s.option("count_traversal", 1)
        s.draw()
        ref = o.render(160, 90)
        rb = o.mesh_bvh(0); h = s.meshes[0]
        if not np.array_equal(_accel(rt64_lib, rt64_lib.ReadbackMeshAccel, h, rt64.ACCEL_MORTON, np.uint32), rb["morton"]): bad.append("morton keys")
        if not np.array_equal(_accel(rt64_lib, rt64_lib.ReadbackMeshAccel, h, rt64.ACCEL_SORTED_INDEX, np.uint32), rb["sortedIndex"]): bad.append("sorted order")
        nodes = _accel(rt64_lib, rt64_lib.ReadbackMeshAccel, h, rt64.ACCEL_NODES, oracle_py.NODE_DTYPE)
        for f in ("left", "right", "lmin", "lmax", "rmin", "rmax"):
            if not np.array_equal(nodes[f].view(np.uint32), rb["nodes"][f].view(np.uint32)): bad.append("nodes." + f)
        tris = _accel(rt64_lib, rt64_lib.ReadbackMeshAccel, h, rt64.ACCEL_TRIANGLES, oracle_py.TRI_DTYPE)[:n]
        rt = o.mesh_tris(0)
        for f in ("v0", "v1", "v2", "prim"):
            if not np.array_equal(tris[f].view(np.uint32), rt[f][:n].view(np.uint32)): bad.append("triangles." + f)
        hdr = _accel(rt64_lib, rt64_lib.ReadbackMeshAccel, h, rt64.ACCEL_HEADER, np.float32)
        if not (np.array_equal(hdr[0:3], rb["bmin"]) and np.array_equal(hdr[4:7], rb["bmax"])): bad.append("mesh box")
        if not np.array_equal(s.readback(rt64.IMAGE_PRIMARY_HIT), ref["primaryHit"]): bad.append("hit records of the frame")
        st = s.stats(); c = ref["counters"]
        if (st.nodesVisited, st.trianglesTested, st.primaryRays, st.shadowRays) != (c["nodesVisited"], c["trianglesTested"], c["primaryRays"], c["shadowRays"]):
            bad.append("counters %s against %s" % ((st.nodesVisited, st.trianglesTested, st.primaryRays, st.shadowRays), (c["nodesVisited"], c["trianglesTested"], c["primaryRays"], c["shadowRays"])))
        if rt64_lib.last_error() and "dropped" in rt64_lib.last_error(): bad.append(rt64_lib.last_error())
        if refit:                      # RT64_SetMesh with the same counts on an UPDATABLE mesh: the tree keeps its topology, the boxes are refitted (rt64_mesh.cpp:129,149-157)
            rng = np.random.default_rng(seed + 7)
            v = data.meshes[0].vertices.copy()
            v["position"][:, :3] += rng.uniform(-0.4, 0.4, size=(len(v), 3)).astype(np.float32) * np.float32(1e-3 if kind == "far" else 1.0)
            s.set_mesh(s.meshes[0], v, data.meshes[0].indices); o.set_mesh(o.meshes[0], v, data.meshes[0].indices)
            s.draw(); ref = o.render(160, 90)
            rb = o.mesh_bvh(0)
            nodes = _accel(rt64_lib, rt64_lib.ReadbackMeshAccel, h, rt64.ACCEL_NODES, oracle_py.NODE_DTYPE)
            for f in ("left", "right", "lmin", "lmax", "rmin", "rmax"):
                if not np.array_equal(nodes[f].view(np.uint32), rb["nodes"][f].view(np.uint32)): bad.append("refit: nodes." + f)
            if not np.array_equal(s.readback(rt64.IMAGE_PRIMARY_HIT), ref["primaryHit"]): bad.append("refit: hit records of the frame")
            st = s.stats(); c = ref["counters"]
            if (st.nodesVisited, st.trianglesTested) != (c["nodesVisited"], c["trianglesTested"]): bad.append("refit: counters")
    finally:
        s.close(); o.close()
    return ["%s, %d triangles%s: %s" % (kind, n, ", refitted" if refit else "", b) for b in bad]


@pytest.mark.parametrize("seed", list(range(1, 21)))
def test_random_meshes_build_the_oracles_tree(rt64_lib, sample_data, seed):
    bad = compare_mesh(rt64_lib, sample_data, seed)
    assert not bad, (seed, bad)
