#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference ships no golden vectors for this path and cannot run here, so these fixtures pin the ORACLE's
own outputs (regression guard for the checker) on the reference's sample scene:
  c1_256_hits.npz   BASELINE config C1: 256x256 primary visibility -- instance (i8), primitive (u16), t/u/v (f32 bits)
  c2_240x135.npz    BASELINE config C2 at 1/8 size: final RGBA8, composed RGBA32F, direct light, diffuse, instance ids
  kats.json         known-answer values of the small helpers (RNG, Halton, Morton, combiner decode, half floats, BC7 hash)
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from sm64rt_legacy_renderer_amd import sample_scene  # noqa: E402
from oracle import oracle_py  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    import ctypes as C
    data = sample_scene.make_sample_scene()
    L = oracle_py.lib()
    o = oracle_py.OracleScene(data)
    r = o.render(256, 256)
    hit = r["primaryHit"]
    miss = hit[..., 3] == 0xFFFFFFFF
    inst = np.where(miss, -1, (hit[..., 3] >> 24).astype(np.int32)).astype(np.int8)
    prim = np.where(miss, 0xFFFF, hit[..., 3] & 0xFFFF).astype(np.uint16)
    np.savez_compressed(os.path.join(HERE, "c1_256_hits.npz"), instance=inst, prim=prim, t=hit[..., 0], u=hit[..., 1], v=hit[..., 2])
    o.close()

    o = oracle_py.OracleScene(data)
    r = o.render(240, 135)
    np.savez_compressed(os.path.join(HERE, "c2_240x135.npz"), final=r["final"], output=r["output"][..., :3].astype(np.float32),
                        direct=r["directLight"][..., :3].astype(np.float16), diffuse=(r["diffuse"] * 255.0 + 0.5).astype(np.uint8),
                        instanceId=r["instanceId"].astype(np.int8),
                        counters=np.array([r["counters"][k] for k in ("primaryRays", "shadowRays", "nodesVisitedPrimary", "trianglesTestedPrimary",
                                                                       "nodesVisitedShadow", "trianglesTestedShadow")], dtype=np.int64))
    # texture decode hash (BC7 sphere albedo, all 10 mips)
    tex = o.textures[0]
    h = hashlib.sha256()
    w, hh, m = C.c_int(), C.c_int(), C.c_int()
    L.oracle_texture_info(tex, C.byref(w), C.byref(hh), C.byref(m))
    for mip in range(m.value):
        mw, mh = C.c_int(), C.c_int()
        p = L.oracle_texture_mip(tex, mip, C.byref(mw), C.byref(mh))
        h.update(np.ctypeslib.as_array(p, shape=(mw.value * mh.value * 4,)).tobytes())
    bvh = o.mesh_bvh(0)
    o.close()

    seeds = [(0, 0), (1, 0), (12345, 7), (1920 * 1080 - 1, 63)]
    rng = []
    for a, b in seeds:
        s = L.oracle_init_rand(a, b, 16)
        st = C.c_uint32(s)
        vals = [float(L.oracle_next_rand(C.byref(st))) for _ in range(3)]
        rng.append({"val0": a, "val1": b, "seed": int(s), "next": vals})
    comb = (C.c_int * 28)()
    L.oracle_decode_combiner(0x01200a00, comb)
    kats = {
        "init_rand": rng,
        "halton": {"base2": [float(L.oracle_halton(i, 2)) for i in range(1, 9)], "base3": [float(L.oracle_halton(i, 3)) for i in range(1, 9)]},
        "morton30": [[1023, 0, 0, int(L.oracle_morton30(1023, 0, 0))], [0, 1023, 0, int(L.oracle_morton30(0, 1023, 0))],
                     [5, 9, 1000, int(L.oracle_morton30(5, 9, 1000))]],
        "combiner_0x01200a00": list(comb),
        "bc7_grass_dif_sha256": h.hexdigest(), "bc7_mips": m.value,
        "sphere_blas": {"count": int(bvh["count"]), "morton_sha256": hashlib.sha256(bvh["morton"].tobytes()).hexdigest(),
                        "sorted_sha256": hashlib.sha256(bvh["sortedIndex"].tobytes()).hexdigest(),
                        "nodes_sha256": hashlib.sha256(bvh["nodes"].tobytes()).hexdigest()},
    }
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(kats, f, indent=1)
    print("golden fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
