#!/bin/bash
# Waves-per-SIMD targets of the ray kernels revisited now that they are compiled without SLP pairs (lower register pressure everywhere): LEAN_WAVES 4, DIRECT_WAVES 4, SPLIT_WAVES 6.   build | run
cd "$(dirname "$0")/../.."
CS=sm64rt-legacy-renderer_amd/csrc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-slp-vectorize"
declare -A V
V[lw4]="-DLEAN_WAVES=4"
V[dw4]="-DDIRECT_WAVES=4"
V[sw6]="-DSPLIT_WAVES=6"
V[sw6dw4]="-DSPLIT_WAVES=6 -DDIRECT_WAVES=4"
V[tw5]="-DTRACE_WAVES=5"
V[tw6]="-DTRACE_WAVES=6"
V[hw4]="-DHIT_WAVES=4"
V[rw4]="-DREFLECT_WAVES=4"
V[hw4rw4]="-DHIT_WAVES=4 -DREFLECT_WAVES=4"
LIST="${LIST:-lw4 dw4 sw6 sw6dw4}"
CONFIGS="${CONFIGS:-C2 C3 C5}"
if [ "$1" = build ]; then
  mkdir -p tools/exp/build
  n=0
  for v in $LIST; do
    for f in passes passes_simple; do
      /opt/rocm/bin/hipcc $BASE ${V[$v]} -c $CS/$f.hip -o tools/exp/build/${f}_$v.o &
    done
    n=$((n+1)); [ $((n % 2)) = 0 ] && wait
  done
  wait
  for v in $LIST; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_$v.so $CS/build/lbvh.o tools/exp/build/passes_$v.o tools/exp/build/passes_simple_$v.o $CS/build/bc7.o $CS/build/svgf.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  done
  ls -la tools/exp/build/*.so
  exit 0
fi
for v in ship $LIST ship; do
  L=$PWD/tools/exp/build/librt64_$v.so; [ $v = ship ] && L=$PWD/sm64rt-legacy-renderer_amd/librt64.so
  for c in $CONFIGS; do
    RT64_ASSETS_DIR=$PWD/assets RT64_LIBRARY_PATH=$L python bench.py --config $c --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r04_wv_${v}_$c.json 2> gpurun_out/r04_wv_${v}_$c.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/r04_wv_${v}_$c.json").read())
print("$v $c", d["ms_per_step"], (d.get("enqueued_frames") or {}).get("ms_per_step"), d["roofline"]["ms_per_launch"], d.get("parity", {}).get("pass"))
PY
  done
done
