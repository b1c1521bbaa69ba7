#!/usr/bin/env python3
"""What each rank of an N-way band partition costs, measured on ONE GPU: tools/band_costs.py [--config C5] [--ranks 2,4,8] [--frames 40]

For a GI + SVGF configuration the frame is cut into cost-balanced contiguous bands exactly as `bench.py --gpus N` cuts it (RT64_BalanceGatherBands over the
lit pixels of every row).  One device renders band r of N alone, r = 0 .. N-1, synchronously, in the two halo modes:
  recompute : the band re-renders the denoiser's halo (66 rows per side) -- complete, this is what a rank of the recompute partition does;
  exchange  : the band renders its rows + 4, makes its filter input, and runs the a-trous iterations over the rows they still reach; device option
              halo_dry_run = 1 skips the transfer itself (no peer on a one-GPU box), so this is the rank's GPU work WITHOUT the exchange (5.7 MB to and from
              each neighbour at 4K: add 37-75 us at 153-77 GB/s per xGMI link).
The slowest band bounds the frame: whole-frame time / max over ranks = the scaling the partition allows before the gather (which overlaps the next frame).
--rebalance R adds R rounds of measured-cost feedback: the bands are re-cut from the per-rank times just measured (RT64_RebalanceGatherBands: what
`bench.py --gpus N` does between its ranks with one all-gather per round) and measured again.  One JSON line per (N, mode, round)."""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from sm64rt_legacy_renderer_amd import rt64, sample_scene

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C5"); ap.add_argument("--ranks", default="2,4,8"); ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--rebalance", type=int, default=0, help="rounds of measured-cost feedback (RT64_RebalanceGatherBands) after the modelled cut; every round is measured")
ap.add_argument("--modes", default="recompute,exchange")
args = ap.parse_args()
cfg = sample_scene.BENCH_CONFIGS[args.config]
W, H = cfg["width"], cfg["height"]
lib = rt64.Library()


def make():
    data = sample_scene.make_sample_scene()
    sample_scene.apply_bench_config(data, args.config)
    s = sample_scene.Rt64Scene(lib, data, W, H, hip_device=0)
    s.set_view_description(gi_samples=cfg["gi_samples"], denoiser=cfg["denoiser"])
    for k in ("primary_spp", "gi_bounces"):
        s.option(k, cfg.get(k, 1))
    s.option("profile_passes", 0)
    return s


def timed(s, frames):
    for _ in range(12):
        s.draw()
    t0 = time.perf_counter()
    for _ in range(frames):
        s.draw()
    return (time.perf_counter() - t0) * 1e3 / frames


s = make()
whole = timed(s, args.frames)
lit = (s.readback(rt64.IMAGE_FIRST_INSTANCE_ID) >= 0).sum(axis=1).astype(np.uint32)
s.close()
print(json.dumps({"config": args.config, "size": [W, H], "whole_frame_ms": round(whole, 4)}), flush=True)
noop = rt64.HALO_EXCHANGE(lambda user, regions, count: None)
for N in [int(x) for x in args.ranks.split(",")]:
    starts = (C.c_int * (N + 1))()
    lib.BalanceGatherBands(lit.ctypes.data_as(C.POINTER(C.c_uint)), W, H, N, starts)
    st = [int(v) for v in starts]
    for mode in args.modes.split(","):
        cur = (C.c_int * (N + 1))(*st)
        for rnd in range(args.rebalance + 1):
            now = [int(v) for v in cur]
            ms = []
            for r in range(N):
                s = make()
                s.set_tile(now[r], now[r + 1])
                if mode == "exchange":
                    assert lib.SetDeviceHaloExchange(s.device, C.cast(noop, C.c_void_p), None, cur, r, N) == 1
                    assert s.option("halo_dry_run", 1)
                ms.append(timed(s, args.frames))
                s.close()
            print(json.dumps({"config": args.config, "ranks": N, "halo": mode + (" (GPU work only: no transfer)" if mode == "exchange" else ""),
                              "bands": "modelled cut (RT64_BalanceGatherBands)" if rnd == 0 else "after %d round%s of measured-cost feedback (RT64_RebalanceGatherBands)" % (rnd, "" if rnd == 1 else "s"),
                              "band_starts": now, "ms_per_rank": [round(v, 4) for v in ms], "slowest_ms": round(max(ms), 4), "whole_over_slowest": round(whole / max(ms), 2)}), flush=True)
            if rnd < args.rebalance:
                nxt = (C.c_int * (N + 1))()
                msf = (C.c_float * N)(*ms)
                assert lib.RebalanceGatherBands(H, N, cur, msf, nxt) == 1
                cur = nxt
