python -m pytest tests -q -x -m gpu > gpurun_out/r04_full4.txt 2>&1
tail -15 gpurun_out/r04_full4.txt
for e in 1 0; do
  python bench.py --config C5 --steps 60 --warmup 10 --no-cpu-baseline --no-parity --option reflection_early=$e > gpurun_out/r04_c5_early$e.json 2> gpurun_out/r04_c5_early$e.err
  python bench.py --config C5-literal --steps 20 --warmup 5 --no-cpu-baseline --no-parity --option reflection_early=$e > gpurun_out/r04_c5lit_early$e.json 2> gpurun_out/r04_c5lit_early$e.err
done
python - <<'PY'
import json
for f in ("c5_early1", "c5_early0", "c5lit_early1", "c5lit_early0"):
    d = json.loads(open("gpurun_out/r04_%s.json" % f).read())
    print(f, d["ms_per_step"], {k.split("(")[0]: round(v["ms"], 4) for k, v in d["roofline"]["kernels"].items()})
PY
