#!/bin/bash
# svgf_atrous_kernel compiled for 6 (shipped: 68 VGPRs, 7 waves resident), 7 and 8 (64 VGPRs, 20 bytes per lane spilled) waves per SIMD.   build | run
cd "$(dirname "$0")/../.."
CS=sm64rt-legacy-renderer_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/exp/build
  for w in 7 8; do
    sed "s/__launch_bounds__(256, 6) void svgf_atrous_kernel/__launch_bounds__(256, $w) void svgf_atrous_kernel/" $CS/svgf.hip > tools/exp/build/svgf_w$w.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$CS -c tools/exp/build/svgf_w$w.hip -o tools/exp/build/svgf_w$w.o &
  done; wait
  for w in 7 8; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/build/librt64_aw$w.so $CS/build/lbvh.o $CS/build/passes.o $CS/build/passes_simple.o $CS/build/bc7.o tools/exp/build/svgf_w$w.o $CS/build/raster.o $CS/build/upscale.o $CS/build/gather.o $CS/build/rt64_host.o -ldl
  done
  exit 0
fi
for w in 6 7 8 6; do
  L=$PWD/tools/exp/build/librt64_aw$w.so; [ $w = 6 ] && L=$PWD/sm64rt-legacy-renderer_amd/librt64.so
  for c in C3 C5; do
    RT64_ASSETS_DIR=$PWD/assets RT64_LIBRARY_PATH=$L python bench.py --config $c --no-cpu-baseline --steps 100 --warmup 10 --pass-events-every 1 > gpurun_out/r04_aw${w}_$c.json 2> gpurun_out/r04_aw${w}_$c.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/r04_aw${w}_$c.json").read())
pe = d.get("passes_ms") or d.get("pass_ms") or {}
print("atrous waves $w $c", d["ms_per_step"], d.get("parity", {}).get("pass"), {k: v for k, v in pe.items() if "svgf" in k.lower() or "denois" in k.lower()} if isinstance(pe, dict) else "")
PY
  done
done
