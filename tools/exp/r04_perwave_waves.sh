#!/bin/bash
# Per-wave form of the one-kernel frame at 2 / 4 waves per SIMD (PERWAVE_WAVES; shipped: 3): build here, run on the GPU box.
#   tools/exp/r04_perwave_waves.sh build | run
cd "$(dirname "$0")/../.."
CS=sm64rt-legacy-renderer_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/exp/build
  for w in 2 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DPERWAVE_WAVES=$w -c $CS/passes_simple.hip -o tools/exp/build/passes_simple_pw$w.o &
  done; wait
  for w in 2 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_pw$w.so $CS/build/lbvh.o $CS/build/passes.o tools/exp/build/passes_simple_pw$w.o $CS/build/bc7.o $CS/build/svgf.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  done
  exit 0
fi
for w in 3 2 4; do
  L=$PWD/tools/exp/build/librt64_pw$w.so; [ $w = 3 ] && L=$PWD/sm64rt-legacy-renderer_amd/librt64.so
  RT64_ASSETS_DIR=$PWD/assets RT64_LIBRARY_PATH=$L python bench.py --subdiv 7 --floor-grid 256 --no-cpu-baseline --no-parity --steps 100 --warmup 10 > gpurun_out/r04_stress_pw$w.json 2> gpurun_out/r04_stress_pw$w.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_stress_pw$w.json").read())
print("PERWAVE_WAVES $w", d["ms_per_step"], d["enqueued_frames"]["ms_per_step"], d["roofline"]["ms_per_launch"])
PY
done
