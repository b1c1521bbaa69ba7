#!/usr/bin/env python3
"""Copy the sample scene's DATA inputs (meshes, textures, blue-noise table) out of the
read-only reference checkout into assets/.

These files are the inputs of the benchmark scene that BASELINE.json names ("src/sample scene"):
  /root/reference/src/sample/res/*            (scene of src/sample/main.cpp:201-412)
  /root/reference/src/rt64lib/res/bluenoise/  (sample sequence of shaders/BlueNoise.hlsli:7-13)
They are data, not source.  /root/reference does not exist on the GPU box, so tests, smoke()
and bench.py read the copies under assets/ only.

The blue-noise BMP is converted to a raw 512x512 RGBA8 table (the layout the renderer uploads).
Channel mapping: the reference loads an array named *_BGRA8 as RGBA8, so
shader .r = BMP blue channel.  We keep that mapping: table byte 0 = BMP blue, 1 = green, 2 = red.
"""
import os, shutil, sys
import numpy as np
from PIL import Image

REF = "/root/reference/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAMPLE = ["sphere.obj", "grass_dif.dds", "grass_nrm.png", "grass_spc.png", "clouds.png",
          "tiles_dif.png", "tiles_nrm.png", "tiles_spc.png"]

def main():
    dst = os.path.join(ROOT, "assets", "sample")
    os.makedirs(dst, exist_ok=True)
    for name in SAMPLE:
        shutil.copyfile(os.path.join(REF, "sample", "res", name), os.path.join(dst, name))
        os.chmod(os.path.join(dst, name), 0o644)
    bmp = Image.open(os.path.join(REF, "rt64lib", "res", "bluenoise", "LDR_64_64_64_RGB1.bmp")).convert("RGB")
    rgb = np.asarray(bmp, dtype=np.uint8)
    assert rgb.shape == (512, 512, 3)
    table = np.empty((512, 512, 4), dtype=np.uint8)
    table[..., 0] = rgb[..., 2]   # BGRA byte order read as RGBA
    table[..., 1] = rgb[..., 1]
    table[..., 2] = rgb[..., 0]
    table[..., 3] = 255
    table.tofile(os.path.join(ROOT, "assets", "bluenoise_512x512_rgba8.bin"))
    print("assets imported")

if __name__ == "__main__":
    sys.exit(main())
