#!/bin/bash
# A/B of the bounce walk's launch shape (device options bounce_groups / bounce_block): tools/exp/r03_bounce_grid.sh [config]
C=${1:-C5}
OUT=gpurun_out/r03_bounce_grid_$C.jsonl
: > $OUT
run() { python bench.py --config $C --no-cpu-baseline --no-parity --steps 40 --warmup 5 "$@" >> $OUT 2>> gpurun_out/r03_bounce_grid_$C.err; }
run
for g in 768 1024 1280 1536 4096 8192; do run --option bounce_groups=$g; done
for g in 0 512 768 1024 4096; do run --option bounce_block=512 --option bounce_groups=$g; done
for g in 0 1280 4096; do run --option bounce_block=257 --option bounce_groups=$g; done
python - <<PY
import json
for l in open("$OUT"):
    try: j = json.loads(l)
    except Exception: continue
    print(j["config"].get("options"), j["ms_per_step"], {k.split("(")[0]: round(v["ms"], 4) for k, v in j["roofline"]["kernels"].items()})
PY
