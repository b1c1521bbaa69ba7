#!/bin/bash
# A wave's pixel block inside a 16 x 16 tile: 8 x 8 (shipped) against 16 x 4 (whole 64-byte pieces of image rows per store).   build | run
cd "$(dirname "$0")/../.."
CS=sm64rt-legacy-renderer_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/exp/build
  for f in passes passes_simple; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DRT_WAVE_BLOCK_W=16 -c $CS/$f.hip -o tools/exp/build/${f}_wb16.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_wb16.so $CS/build/lbvh.o tools/exp/build/passes_wb16.o tools/exp/build/passes_simple_wb16.o $CS/build/bc7.o $CS/build/svgf.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  exit 0
fi
for v in 8 16; do
  L=$PWD/tools/exp/build/librt64_wb16.so; [ $v = 8 ] && L=$PWD/sm64rt-legacy-renderer_amd/librt64.so
  for c in C2 C3; do
    RT64_ASSETS_DIR=$PWD/assets RT64_LIBRARY_PATH=$L python bench.py --config $c --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r04_wb${v}_$c.json 2> gpurun_out/r04_wb${v}_$c.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/r04_wb${v}_$c.json").read())
print("wave block $v x", 64 // $v, "$c", d["ms_per_step"], (d.get("enqueued_frames") or {}).get("ms_per_step"), d["roofline"]["ms_per_launch"], d.get("parity", {}).get("pass"))
PY
  done
done
