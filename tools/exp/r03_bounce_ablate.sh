#!/bin/bash
# Prices the parts of bounce_trace_plain_kernel by stubbing them out (RT_ABLATE builds of passes_simple.hip; wrong pictures, never shipped):
#   build here:  tools/exp/r03_bounce_ablate.sh build        run on the GPU box:  tools/exp/r03_bounce_ablate.sh run [config]
# 1: sky term of the misses = constant   2: no traversal (every ray a miss) + constant sky   4: no traversal, real sky term.  RT64_ASSETS_DIR is set because the
# diagnostic libraries do not sit next to assets/ (a library that cannot find its blue-noise table refuses to create a device).
cd "$(dirname "$0")/../.."
CS=sm64rt-legacy-renderer_amd/csrc
if [ "$1" = build ]; then
  for a in 1 2 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DRT_ABLATE=$a -c $CS/passes_simple.hip -o tools/exp/build/passes_simple_ab$a.o &
  done; wait
  for a in 1 2 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_ab$a.so $CS/build/lbvh.o $CS/build/passes.o tools/exp/build/passes_simple_ab$a.o $CS/build/bc7.o $CS/build/svgf.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  done
  exit 0
fi
C=${2:-C5}
OUT=gpurun_out/r03_bounce_ablate_$C.jsonl; : > $OUT
for a in 0 1 2 4; do
  L=$PWD/tools/exp/build/librt64_ab$a.so; [ $a = 0 ] && L=$PWD/sm64rt-legacy-renderer_amd/librt64.so
  RT64_ASSETS_DIR=$PWD/assets RT64_LIBRARY_PATH=$L python bench.py --config $C --no-cpu-baseline --no-parity --steps 30 --warmup 5 >> $OUT 2>> gpurun_out/r03_bounce_ablate.err
done
python - <<PY
import json
for a, l in enumerate(open("$OUT")):
    j = json.loads(l)
    print("ablate", a, j["ms_per_step"], {k.split("(")[0]: round(v["ms"], 4) for k, v in j["roofline"]["kernels"].items()})
PY
