#!/bin/bash
# Rehearsal of bench.py's N > 1 paths with FOUR ranks on device 0 of a one-GPU box, through the driver's own launcher (torch.distributed.run): C2 strips through the
# fallback gatherer (RCCL refuses several ranks on one GPU: every rank agrees and falls back together), C3 / C5 bands with the denoiser halo, C5 with the halo
# exchanged over gloo is not possible (the exchange is the library's RCCL path): recompute only.  The gathered frame must have the N = 1 checksum.
P=29611
run() { name=$1; shift
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 4 --same-device --steps 10 --warmup 2 --prewarm 5 --no-cpu-baseline "$@" > gpurun_out/r04_n4_$name.json 2> gpurun_out/r04_n4_$name.err
  echo "$name rc=$?"; P=$((P+1)); }
run C2_nccl
run C2_gloo --backend gloo
run C3_gloo --backend gloo --config C3
run C5_nccl --config C5
python bench.py --steps 10 --warmup 2 --prewarm 5 --no-cpu-baseline --no-parity > gpurun_out/r04_n4_ref_C2.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --prewarm 5 --no-cpu-baseline --no-parity --config C3 > gpurun_out/r04_n4_ref_C3.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --prewarm 5 --no-cpu-baseline --no-parity --config C5 > gpurun_out/r04_n4_ref_C5.json 2>/dev/null
python - <<'PY'
import json
for f in ("C2_nccl", "C2_gloo", "C3_gloo", "C5_nccl", "ref_C2", "ref_C3", "ref_C5"):
    try:
        d = json.loads(open("gpurun_out/r04_n4_%s.json" % f).read())
        print(json.dumps({"run": f, "n_gpus": d["n_gpus"], "ms_per_step": d["ms_per_step"], "frame_checksum": d.get("frame_checksum"), "partition": d["config"].get("partition"), "gather": (d.get("pipeline") or {}).get("gather"), "parity": (d.get("parity") or {}).get("pass")}))
    except Exception as e:
        print(f, "no line", e)
PY
