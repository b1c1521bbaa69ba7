"""Host logic of bench.py that needs no GPU: the `roofline` object is built from the counter profile that belongs to exactly these kernel
sources (refused otherwise), its `frac` is the utilisation of the binding resource (<= 1 by construction), and the SURVEY 8(d) algorithmic
bytes stay in the line under their own name; `--gpus N` without a launcher spawns N ranks (and fails when one fails)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _profile(tmp_path, bench, monkeypatch, source_hash, kernels):
    prof = tmp_path / "profiles"; prof.mkdir()
    (prof / "kernel_counters.json").write_text(json.dumps({"source_hash": source_hash, "workloads": {"C2": {"kernels": kernels}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_hash", lambda: "abc123")
    bench.COUNTER_WORKLOAD[0] = "C2"


LEAN = {"hbm_bytes": 216e6, "fetch_bytes_x2": 65e6, "write_bytes": 151e6, "SQ_INSTS_VALU": 68.9e6, "SQ_ACTIVE_INST_VALU": 70.1e6, "SQ_WAVES": 32640,
        "SQ_WAVE_CYCLES": 7.7e8, "SQ_WAIT_ANY": 5.1e8, "SQ_WAIT_INST_ANY": 6.7e7, "SQ_ACTIVE_INST_ANY": 1.6e8, "SQ_BUSY_CYCLES": 3.6e7, "avg_ns": 166000.0,
        "VGPR_Count": 84, "Scratch_Size": 120}


def test_roofline_uses_the_matching_profile_and_stays_below_one(bench, tmp_path, monkeypatch):
    _profile(tmp_path, bench, monkeypatch, "abc123", {"lean_frame": LEAN})
    r = bench.roofline_object("lean_frame(trace+shade+direct+compose)", 0.166, 1_456_816_660, {"lean_frame(trace+shade+direct+compose)": (0.166, 1_456_816_660)})
    assert r["bound"] in ("valu", "hbm") and 0.0 < r["frac"] <= 1.0 and r["traffic"] == 216000000
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    c = r["counters"]
    assert 0.15 < c["hbm"]["frac"] < 0.18 and 0.3 < c["valu"]["issue_frac"] < 0.4          # 216 MB / 166 us = 1.3 TB/s of 8; 68.9 M x 128 flop-slots / 166 us of 157.3 TF
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s"
    # the SURVEY 8(d) figure is still there, named for what it is (8.8 TB/s "achieved" on algorithmic bytes would be 1.1 x the HBM peak)
    assert r["algorithmic"]["bytes_per_launch"] == 1_456_816_660 and r["algorithmic"]["GBps"] > 8000.0 and "frac" not in r["algorithmic"]
    # neither pipe is half busy: the line has to say what holds the kernel instead (wait share of the wave cycles, waves resident per SIMD)
    if "SQ_WAIT_ANY" in LEAN and "SQ_WAVE_CYCLES" in LEAN:
        assert r["limiter"].startswith("latency") and "waves per SIMD" in r["limiter"]


def test_roofline_refuses_a_profile_of_other_sources(bench, tmp_path, monkeypatch):
    _profile(tmp_path, bench, monkeypatch, "some-other-build", {"lean_frame": LEAN})
    r = bench.roofline_object("lean_frame(trace+shade+direct+compose)", 0.166, 1_456_816_660, {"lean_frame(trace+shade+direct+compose)": (0.166, 1_456_816_660)})
    assert r["frac"] is None and r["traffic"] is None and r["achieved"] is None and "refused" in r["note"]
    assert r["algorithmic"]["bytes_per_launch"] == 1_456_816_660


def test_kernel_groups_sum_their_launches(bench, tmp_path, monkeypatch):
    k = {n: dict(LEAN, hbm_bytes=10e6, SQ_INSTS_VALU=1e6) for n in ("svgf_guide", "svgf_variance", "svgf_atrous")}
    _profile(tmp_path, bench, monkeypatch, "abc123", k)
    r = bench.roofline_object("svgf_denoise(6 launches)", 0.2, 5e8, {"svgf_denoise(6 launches)": (0.2, 5e8)})
    assert r["traffic"] == int(7 * 10e6) and r["counters"]["valu"]["wave_insts_per_launch"] == 7_000_000          # guide + variance + 5 a-trous


def test_committed_profile_belongs_to_the_committed_kernels(bench):
    """profiles/kernel_counters.json must have been collected from the kernel sources in the tree (bench.py refuses it otherwise and the
    driver's line would carry no roofline fraction)."""
    path = os.path.join(ROOT, "profiles", "kernel_counters.json")
    if not os.path.exists(path):
        pytest.skip("no counter profile committed yet")
    doc = json.load(open(path))
    assert doc["source_hash"] == bench.source_hash(), "re-run tools/profile_round.sh on the GPU box and commit profiles/kernel_counters.json"
    assert "lean_frame" in doc["workloads"]["C2"]["kernels"]
    # ... every workload the documents quote was profiled on these sources, and the documents name this profile (not an earlier one)
    assert {"C2", "C3", "C4", "C5", "C4-literal", "C5-literal", "stress_7_256"} <= set(doc["workloads"])
    for name in ("DESIGN.md", os.path.join("profiles", "README.md")):
        assert doc["source_hash"] in open(os.path.join(ROOT, name)).read(), name + " quotes another source hash"
    # the ray kernels keep the flag their measured figures were taken with (csrc/Makefile: 4-5 % on the frame kernel)
    mk = open(os.path.join(ROOT, "sm64rt-legacy-renderer_amd", "csrc", "Makefile")).read()
    assert "build/passes.o build/passes_simple.o: CXXFLAGS += -fno-slp-vectorize" in mk


def test_gpus_flag_spawns_ranks_and_fails_loudly():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself; here (no GPU) both exit non-zero and so does the parent --
    a request for N GPUs never turns into a one-GPU line with rc 0.  With a launcher's WORLD_SIZE that disagrees: refused."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and '"n_gpus": 1' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
