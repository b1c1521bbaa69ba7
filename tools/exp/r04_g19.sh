# rehearsals of the N > 1 code paths of bench.py on one GPU (two ranks on device 0) + a few stress A/Bs
python bench.py --gpus 2 --backend gloo --same-device --steps 20 --warmup 3 --prewarm 10 > gpurun_out/r04_reh_gloo.json 2> gpurun_out/r04_reh_gloo.err; echo "gloo rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --same-device --steps 20 --warmup 3 --prewarm 10 > gpurun_out/r04_reh_nccl.json 2> gpurun_out/r04_reh_nccl.err; echo "nccl-same-device rc=$?"
tail -3 gpurun_out/r04_reh_nccl.err
python bench.py --gpus 2 --backend gloo --same-device --config C3 --steps 10 --warmup 3 --prewarm 10 > gpurun_out/r04_reh_gloo_C3.json 2> gpurun_out/r04_reh_gloo_C3.err; echo "gloo C3 rc=$?"
python - <<'PY'
import json
for f in ("reh_gloo", "reh_nccl", "reh_gloo_C3"):
    try:
        d = json.loads(open("gpurun_out/r04_%s.json" % f).read()); print(f, d["n_gpus"], d["ms_per_step"], d["frame_checksum"], d["pipeline"]["gather"][:60], d["pipeline"].get("control_plane"))
    except Exception as e:
        print(f, "no line", e)
PY
python bench.py --steps 50 --no-cpu-baseline --no-parity > gpurun_out/r04_c2_ref.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r04_c2_ref.json').read()); print('N=1 checksum', d['frame_checksum'])"
for o in tile_order=0 tile_order=1; do
  python bench.py --subdiv 7 --floor-grid 256 --steps 100 --warmup 10 --no-cpu-baseline --no-parity --option $o > gpurun_out/r04_stress_$o.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r04_stress_$o.json').read()); print('stress $o', d['ms_per_step'], d['enqueued_frames']['ms_per_step'])"
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
