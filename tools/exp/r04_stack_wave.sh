#!/bin/bash
# LDS stack entries per lane of the per-wave frame kernel: 24 (shipped) against 40 / 48 -- needs profiles/r04_experiments/stack_lds_wave.patch applied to csrc/passes.hip (the macro RT_STACK_LDS_WAVE is not in the shipped source: no effect was measured).   build | run
cd "$(dirname "$0")/../.."
CS=sm64rt-legacy-renderer_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/exp/build
  for w in 24 40 48; do
    for f in passes passes_simple; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DRT_STACK_LDS_WAVE=$w -c $CS/$f.hip -o tools/exp/build/${f}_sw$w.o &
    done
  done; wait
  for w in 24 40 48; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_sw$w.so $CS/build/lbvh.o tools/exp/build/passes_sw$w.o tools/exp/build/passes_simple_sw$w.o $CS/build/bc7.o $CS/build/svgf.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  done
  exit 0
fi
for w in 24 40 48; do
  L=$PWD/tools/exp/build/librt64_sw$w.so
  RT64_ASSETS_DIR=$PWD/assets RT64_LIBRARY_PATH=$L python bench.py --subdiv 7 --floor-grid 256 --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r04_stress_sw$w.json 2> gpurun_out/r04_stress_sw$w.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_stress_sw$w.json").read())
print("RT_STACK_LDS_WAVE $w", d["ms_per_step"], d["enqueued_frames"]["ms_per_step"], d["roofline"]["ms_per_launch"], d["parity"]["pass"], d["parity"]["hit_mismatches"])
PY
done
