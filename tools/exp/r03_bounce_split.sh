#!/bin/bash
# A/B of the two-phase bounce walk (device option bounce_split): tools/exp/r03_bounce_split.sh
OUT=gpurun_out/r03_bounce_split.jsonl
: > $OUT
run() { python bench.py --no-cpu-baseline --no-parity --steps 40 --warmup 5 "$@" >> $OUT 2>> gpurun_out/r03_bounce_split.err; }
for C in C5 C3 C4; do
  run --config $C
  run --config $C --option bounce_split=1
  run --config $C --option bounce_split=1 --option bounce_groups=4096
  run --config $C --option bounce_split=1 --option bounce_groups=8192
  run --config $C --option bounce_split=1 --option bounce_groups=4096 --option bounce_block=257
done
python - <<PY
import json
for l in open("$OUT"):
    try: j = json.loads(l)
    except Exception: continue
    print(j["config"]["workload"][:2], j["config"].get("options"), j["ms_per_step"], {k.split("(")[0]: round(v["ms"], 4) for k, v in j["roofline"]["kernels"].items()})
PY
