python -m pytest tests/test_gpu_features.py tests/test_gpu_configs.py tests/test_gpu_halo.py tests/test_gpu_edge_cases.py -q -x -m gpu > gpurun_out/r04_t10.txt 2>&1
tail -5 gpurun_out/r04_t10.txt
python bench.py --config C5 --steps 60 --warmup 10 --no-cpu-baseline --no-parity > gpurun_out/r04_c5_c.json 2> gpurun_out/r04_c5_c.err
python bench.py --config C5 --steps 60 --warmup 10 --no-cpu-baseline --no-parity --option overlap_reflection=0 > gpurun_out/r04_c5_c_serial.json 2> gpurun_out/r04_c5_c_serial.err
python bench.py --config C5-literal --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/r04_c5lit_c.json 2> gpurun_out/r04_c5lit_c.err
python - <<'PY'
import json
for f in ("c5_c", "c5_c_serial", "c5lit_c"):
    d = json.loads(open("gpurun_out/r04_%s.json" % f).read())
    print(f, d["ms_per_step"], {k.split("(")[0]: round(v["ms"], 4) for k, v in d["roofline"]["kernels"].items()})
PY
