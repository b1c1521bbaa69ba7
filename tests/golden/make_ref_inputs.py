"""Regenerates tests/golden/ref_inputs.json + ref_sphere_posnrm.f32 from the reference's own asset loaders (build container only: needs /root/reference).
    python tests/golden/make_ref_inputs.py
Builds oracle/_ref/ref_inputs_dump (oracle/ref_inputs.mk: stb_image.h + tiny_obj_loader.h compiled unmodified from /root/reference/src/sample/contrib),
runs it on /root/reference/src/sample/res and copies its index + the sphere's positions / normals here.  The fixtures are DATA -- sizes, hashes and the
float arrays tiny_obj_loader returns for sphere.obj -- not reference source."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "ref_inputs.mk", "-B"])
    shutil.copyfile(os.path.join(ROOT, "oracle", "_ref", "index.json"), os.path.join(HERE, "ref_inputs.json"))
    shutil.copyfile(os.path.join(ROOT, "oracle", "_ref", "sphere.posnrm"), os.path.join(HERE, "ref_sphere_posnrm.f32"))
    print(open(os.path.join(HERE, "ref_inputs.json")).read())
