#!/bin/bash
# The ray kernels compiled without the SLP vectorizer (no v_pk_* pairs, fewer register moves, 151 instead of 168 VGPRs, no scratch in the C2 frame kernel) against the shipped build.   build | run
cd "$(dirname "$0")/../.."
CS=sm64rt-legacy-renderer_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/exp/build
  for f in passes passes_simple; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-slp-vectorize -c $CS/$f.hip -o tools/exp/build/${f}_noslp.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_noslp.so $CS/build/lbvh.o tools/exp/build/passes_noslp.o tools/exp/build/passes_simple_noslp.o $CS/build/bc7.o $CS/build/svgf.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  exit 0
fi
for v in ship noslp ship noslp; do
  L=$PWD/tools/exp/build/librt64_noslp.so; [ $v = ship ] && L=$PWD/sm64rt-legacy-renderer_amd/librt64.so
  for c in C2 C3 C5; do
    RT64_ASSETS_DIR=$PWD/assets RT64_LIBRARY_PATH=$L python bench.py --config $c --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r04_slp_${v}_$c.json 2> gpurun_out/r04_slp_${v}_$c.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/r04_slp_${v}_$c.json").read())
print("$v $c", d["ms_per_step"], (d.get("enqueued_frames") or {}).get("ms_per_step"), d["roofline"]["ms_per_launch"], d.get("parity", {}).get("pass"))
PY
  done
done
