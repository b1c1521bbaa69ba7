#!/bin/bash
# Code-generation flags for the ray kernels (passes.hip, passes_simple.hip) on top of -fno-slp-vectorize, and svgf.hip without SLP.   build | run
cd "$(dirname "$0")/../.."
CS=sm64rt-legacy-renderer_amd/csrc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
declare -A V
V[ilp]="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
V[memclause]="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause"
V[occ100]="-fno-slp-vectorize -mllvm -amdgpu-schedule-metric-bias=100"
V[slpthr8]="-mllvm -slp-threshold=8"
V[o2]="-fno-slp-vectorize -O2"
V[svgf]="-fno-slp-vectorize"
if [ "$1" = build ]; then
  mkdir -p tools/exp/build
  for v in ilp memclause occ100 slpthr8 o2; do
    for f in passes passes_simple; do
      /opt/rocm/bin/hipcc $BASE ${V[$v]} -c $CS/$f.hip -o tools/exp/build/${f}_$v.o &
    done
    [ $v = memclause ] && wait
    [ $v = slpthr8 ] && wait
  done
  /opt/rocm/bin/hipcc $BASE -fno-slp-vectorize -c $CS/svgf.hip -o tools/exp/build/svgf_noslp.o &
  wait
  for v in ilp memclause occ100 slpthr8 o2; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_$v.so $CS/build/lbvh.o tools/exp/build/passes_$v.o tools/exp/build/passes_simple_$v.o $CS/build/bc7.o $CS/build/svgf.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_svgf.so $CS/build/lbvh.o tools/exp/build/passes_noslp.o tools/exp/build/passes_simple_noslp.o $CS/build/bc7.o tools/exp/build/svgf_noslp.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  ls -la tools/exp/build/*.so
  exit 0
fi
for v in noslp ilp memclause occ100 slpthr8 o2 svgf noslp; do
  L=$PWD/tools/exp/build/librt64_$v.so
  for c in C2 C3 C5; do
    RT64_ASSETS_DIR=$PWD/assets RT64_LIBRARY_PATH=$L python bench.py --config $c --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r04_fl_${v}_$c.json 2> gpurun_out/r04_fl_${v}_$c.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/r04_fl_${v}_$c.json").read())
print("$v $c", d["ms_per_step"], (d.get("enqueued_frames") or {}).get("ms_per_step"), d["roofline"]["ms_per_launch"], d.get("parity", {}).get("pass"))
PY
  done
done
