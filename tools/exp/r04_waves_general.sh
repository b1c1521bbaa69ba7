#!/bin/bash
# HIT_WAVES / REFLECT_WAVES 4 against 3 on the GENERAL kernels (device option simple_kernels = 0: what frames with non-power-of-two textures or translucent shadows run).   build | run
cd "$(dirname "$0")/../.."
CS=sm64rt-legacy-renderer_amd/csrc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-slp-vectorize"
if [ "$1" = build ]; then
  mkdir -p tools/exp/build
  for f in passes passes_simple; do
    /opt/rocm/bin/hipcc $BASE -DHIT_WAVES=3 -DREFLECT_WAVES=3 -c $CS/$f.hip -o tools/exp/build/${f}_w3.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_w3.so $CS/build/lbvh.o tools/exp/build/passes_w3.o tools/exp/build/passes_simple_w3.o $CS/build/bc7.o $CS/build/svgf.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  exit 0
fi
for v in w4 w3 w4 w3; do
  L=$PWD/tools/exp/build/librt64_w3.so; [ $v = w4 ] && L=$PWD/sm64rt-legacy-renderer_amd/librt64.so
  for c in C3 C5 C5-literal; do
    RT64_ASSETS_DIR=$PWD/assets RT64_LIBRARY_PATH=$L python bench.py --config $c --no-cpu-baseline --no-parity --steps 100 --warmup 10 --option simple_kernels=0 > gpurun_out/r04_wg_${v}_$c.json 2> gpurun_out/r04_wg_${v}_$c.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/r04_wg_${v}_$c.json").read())
print("general kernels, hit / reflection waves $v $c", d["ms_per_step"])
PY
  done
done
