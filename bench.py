#!/usr/bin/env python3
"""bench.py -- headline benchmark of the RT64 render path on MI355X.

Metric (BASELINE.json): Mrays/s + ms/frame on the reference's sample scene, 1920x1080, 1 spp.
A "step" is one RT64_DrawDevice of configuration C2 (TLAS build, primary rays, any-hit shading, direct light with
shadow rays, constant ambient, compose, post -> RGBA8 back buffer) through the C ABI of librt64.so, with every input
already resident in HBM.  value = (primary + shadow rays of the frame, counted on the device) / frame time.

N GPUs: one process per GPU (torch.distributed, backend nccl == RCCL).  The SAME 1080p frame is image-tile partitioned:
rank r renders the 16-row strips r, r+N, ... and every step ends with one gather of the packed RGBA8 strips to rank 0
over xGMI (strong scaling: total work fixed).

Extra objects in the JSON line:
  roofline      dominant kernel of the step: algorithmic bytes per launch / its mean launch duration (HIP events recorded
                on the library's stream inside the timed region) against the 8 TB/s HBM peak
  cpu_baseline  the scalar C oracle (same BVH, same shading math, OpenMP over rows) on the box's host cores, rank 0, N=1
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HBM_TRIAD_GBS = 4689.0         # measured on the box with tools/micro/triad.hip (stream triad, 3 x 1 GiB; device-to-device copy: 5146 GB/s)

# Algorithmic bytes (DESIGN.md "Kernels and rooflines"): a BVH node record is 64 B, a triangle record 48 B.
NODE_B, TRI_B = 64, 48
PRIMARY_TRACE_PIXEL_B = 16 + 4                 # hit record (t,u,v,prim) + instance id written per primary ray
PRIMARY_SHADE_PIXEL_B = 20 + 94                # hit record read + 15 G-buffer images written (90 B/px + first-instance copy)
PRIMARY_SHADE_PIXEL_LEAN_B = 20 + 8            # lean frame, every pixel: hit record read + diffuse and instance id written
PRIMARY_SHADE_HIT_LEAN_B = 16 + 8 + 8          # lean frame, hit pixels only: position, normal, specular written
PRIMARY_SHADE_HIT_B = 3 * 52 + 4 * 16 * 4      # 3 vertices of the sample layout + 4 bilinear fetches x 4 texels x 4 B
PRIMARY_SHADE_MISS_B = 16                      # one bilinear sky fetch
DIRECT_PIXEL_B = 4 + 8                         # instance id read + RGBA16F light written
DIRECT_HIT_B = 16 + 8 + 8                      # position + normal + specular read for lit pixels
COMPOSE_PIXEL_B = 44 + 16 + 4                  # compose reads 44 B, writes 16 B; post writes 4 B (fused: output not re-read)
COMPOSE_PIXEL_LEAN_B = 4 + 16 + 4              # lean frame, inside direct_kernel: diffuse read, output + back buffer written


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5", "C4-literal", "C5-literal"],
                    help="BASELINE.json configs: C2 primary+shadow 1080p (the metric's config, default) | C3 +1 GI sample +SVGF 1080p | "
                         "C4 1440p, 2 GI samples, SVGF, per-frame SetMesh refit of the UPDATABLE sphere | C5 4K, 4 GI samples, SVGF, reflective floor")
    ap.add_argument("--gi-samples", type=int, default=0, help="C3: 1 (with --denoiser)")
    ap.add_argument("--denoiser", action="store_true")
    ap.add_argument("--subdiv", type=int, default=0, help="stress variant: sphere subdivision levels")
    ap.add_argument("--floor-grid", type=int, default=1, help="stress variant: floor tessellation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size comparison with the oracle after the timed loop (profiling runs)")
    ap.add_argument("--parity-frames", type=int, default=0, help="frames of the GI configurations rendered on a fresh scene + oracle pair for the parity object (0: 2 for C3 / C4 / C5)")
    ap.add_argument("--always-rebuild", action="store_true", help="upload the frame tables and rebuild the TLAS every frame (the reference's behaviour)")
    ap.add_argument("--pass-events-every", type=int, default=0, help="the library records its per-pass HIP events (the live kernel timings of `roofline`) on every n-th timed frame only: an event is a barrier packet "
                    "(~5 us of stream time each, six per GI frame); 0 = auto: 1 for one-kernel frames (C2: two events, 0.4 us), 8 for frames of several passes")
    ap.add_argument("--prewarm", type=int, default=100, help="untimed frames before the W warm-up steps (GPU clock ramp; the line reports them)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="RT64_SetDeviceOption(key, value) before the run (A/B measurements; the line records them under config.options)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="data path of the N > 1 gather: nccl = RCCL over xGMI (the library's own communicator, RT64_CreateGather); gloo: CPU-staged rehearsal of the N>1 path")
    ap.add_argument("--control", default="gloo", choices=["gloo", "nccl"], help="torch.distributed backend of the CONTROL plane (rendezvous of the gather's unique id, barriers, the MAX over ranks, the band-cost all-gather): "
                    "gloo (default) keeps ONE RCCL communicator per process -- the library's; nccl gives torch a second one on the same device (what rounds 1-3 ran)")
    ap.add_argument("--band-rebalance", type=int, default=0, help="N > 1, GI + denoiser bands: rounds of measured-cost feedback after the modelled cut (each rank times its band, one all-gather, "
                    "RT64_RebalanceGatherBands + RT64_SetGatherBands on every rank); 0 (default) keeps the modelled cut.  Opt-in until RT64_SetGatherBands has run between two GPUs (an A/B line, not the headline)")
    ap.add_argument("--halo", default="recompute", choices=["exchange", "recompute"], help="N > 1, GI + denoiser bands: re-render the denoiser's halo rows on every band (default: no mid-frame collective, the path every "
                    "partition test covers) or exchange them between neighbouring bands (RCCL ncclSend / ncclRecv in the middle of the frame: opt-in until it has run between two GPUs)")
    ap.add_argument("--gather", default="auto", choices=["auto", "rccl", "direct"], help="N > 1, in-library gather: rccl = every rank's rows travel to rank 0 through grouped ncclSend / ncclRecv + a reassembly kernel; "
                    "direct = the frame kernels store their rows straight into rank 0's frame slots through an IPC mapping (peer stores over xGMI) and only a 4-byte token per rank goes through RCCL "
                    "(RT64_SetGatherDirect); auto (default) = direct for pixel-local configurations after a self-check -- one frame gathered both ways must give the same bytes on rank 0 -- and rccl otherwise")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on a 1-GPU box: every rank renders on device 0")
    ap.add_argument("--force-gather", action="store_true", help="rehearsal: run the N > 1 code path (enqueued frames + pipelined RCCL gather) with a world of 1")
    ap.add_argument("--pretend-ranks", type=int, default=0, help="diagnosis on a 1-GPU box: render only rank 0's share of a P-way partition, frames enqueued, no gather; `value` is then NOT a throughput of the whole frame")
    ap.add_argument("--cpu-baseline-height", type=int, default=0, help="rows of the frame the CPU baseline renders (0 = all)")
    ap.add_argument("--timed-loop-only", action="store_true", help="profiling runs (tools/profile_round.sh): only the timed loop of `value` -- no always_rebuild leg, no enqueued leg (whose frames overlap on the "
                    "library's render streams: their launches would enter a profiler's per-kernel average with stretched durations), no parity object, no CPU baseline")
    ap.add_argument("--watchdog", type=float, default=600.0, help="N > 1: seconds after which a rank that is still running dumps every thread's Python stack to stderr and exits with status 1 "
                    "(a collective whose peer never arrives cannot be left from inside the process; the launcher then stops the other ranks); 0 = off")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, BEFORE this process imports torch or loads librt64.so
    (no GPU call has happened yet, and none happens in this parent), one child per GPU with the launcher's environment contract,
    and exit with their status.  A child that fails fails the run: a request for N GPUs never ends as a 1-GPU line with rc 0."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # A rank that dies (no such device, a refused option) must not leave the others waiting in a rendezvous or a collective for the backend's own time-out:
    # the first failure ends the run -- the other ranks are stopped, the status is the failing rank's.
    rc, live = 0, list(procs)
    while live:
        for p in list(live):
            st = p.poll()
            if st is None:
                continue
            live.remove(p)
            if st != 0:
                rc = max(rc, abs(st))
                for q in live:
                    q.terminate()
                for q in live:
                    try:
                        q.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        q.kill(); q.wait()
                live = []
                break
        if live:
            time.sleep(0.05)
    raise SystemExit(rc)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    # stdout carries ONE JSON line.  The library reports like the reference does ("Render buffer: WxH", rt64_view.cpp:150) with printf on
    # stdout: from here on file descriptor 1 is stderr, and the line goes out through a duplicate of the real stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import __graft_entry__ as graft
    graft.load_package()
    from sm64rt_legacy_renderer_amd import rt64, sample_scene, tiles

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)       # a hung rendezvous / collective ends as a failed run with its stacks in the log, not as the driver's timeout
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's world size and --gpus must agree" % (args.gpus, world))
    N = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback.")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    G = N > 1 or args.force_gather           # the gather path (always on for N > 1)
    dist = None
    if G:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # The data path (the gather, the halo exchange) is the library's: grouped ncclSend / ncclRecv on its own communicator and stream.  torch.distributed is the
        # control plane only, and on gloo by default: a process then holds ONE RCCL communicator, and nothing of torch's ever runs on the GPU beside the frames.
        torch_backend = "nccl" if (args.backend == "nccl" and args.control == "nccl") else "gloo"
        if N == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        kw = dict(rank=0, world_size=1) if N == 1 else {}
        if torch_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), **kw)
        else:
            # one node: every rank is on this host, so gloo's pairs can use the loopback interface (a container's hostname may not resolve)
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group(backend="gloo", **kw)
    else:
        torch_backend = None
    comm_device = "cuda" if torch_backend == "nccl" else "cpu"

    if args.config != "C2":
        c = sample_scene.BENCH_CONFIGS[args.config]
        args.width, args.height, args.gi_samples, args.denoiser = c["width"], c["height"], c["gi_samples"], c["denoiser"]
    # BASELINE.json's C4 / C5 as worded run through the library's path-tracing extensions (device options; sample_scene.BENCH_CONFIGS)
    ext = {k: sample_scene.BENCH_CONFIGS.get(args.config, {}).get(k, 1) for k in ("primary_spp", "gi_bounces")}
    W, H = args.width, args.height
    COUNTER_WORKLOAD[0] = ("stress_%d_%d" % (args.subdiv, args.floor_grid)) if (args.subdiv or args.floor_grid > 1) else args.config
    lib = rt64.Library()
    data = sample_scene.make_sample_scene(subdiv=args.subdiv, floor_grid=args.floor_grid)
    dbg = os.environ.get("RT64_BENCH_DEBUG", "")          # diagnosis only (never set by the driver): "nosky", "tinytex"
    if "nosky" in dbg:
        data.sky = None
    if "tinytex" in dbg:
        for t in data.textures:
            if t.format == rt64.TEXTURE_FORMAT_RGBA8:
                t.data = np.ascontiguousarray(t.data[:4, :4]); t.width = t.height = 4
    anim = sample_scene.apply_bench_config(data, args.config)     # C4: per-frame SetMesh (host copy + refit) of the UPDATABLE sphere; C5: reflective floor
    scene = sample_scene.Rt64Scene(lib, data, W, H, hip_device=local_rank)
    # Partition: interleaved 16-row strips balance sky against geometry; a frame with GI + denoiser filters across rows, so it is
    # cut into contiguous bands and the library renders each band with the filter's halo (no mid-frame exchange).
    # Rays of one whole frame (the unit of `value`): counted once on every rank before the frame is partitioned, so that rows a
    # rank renders redundantly (denoiser halo) never inflate the throughput.
    if args.gi_samples or args.denoiser:
        scene.set_view_description(gi_samples=args.gi_samples, denoiser=args.denoiser)
    for k, v in ext.items():
        if v != 1 and not scene.option(k, v):
            raise SystemExit("bench.py: librt64.so refused option %s = %s" % (k, v))
    scene.option("count_traversal", 1)
    scene.draw()
    st_full = scene.stats()
    first_frame_build_ms = float(st_full.msBuild)           # every BLAS build recorded by the RT64_SetMesh calls of the scene set-up + the TLAS build: they run at the first RT64_DrawDevice
    rays_total = int(st_full.primaryRays + st_full.shadowRays + st_full.indirectRays + st_full.reflectionRays + st_full.refractionRays)
    scene.option("count_traversal", 0)
    use_bands = N > 1 and args.gi_samples > 0 and args.denoiser
    PR = args.pretend_ranks if (N == 1 and not G) else 0
    if PR > 1 and args.gi_samples > 0 and args.denoiser:
        scene.set_tile(*tiles.band_range(H, 0, PR))
    elif PR > 1:
        scene.set_interleave(0, PR)
    elif use_bands:
        scene.set_tile(*tiles.band_range(H, rank, N))
    else:
        scene.set_interleave(rank, N)
    # The gather itself lives behind the C ABI (RT64_CreateGather / RT64_SubmitGather: grouped ncclSend / ncclRecv on a stream of the
    # library's own, two slots, reassembly kernel on rank 0).  torch.distributed is the control plane only: rendezvous of the unique id,
    # barriers, the MAX over ranks.  The gloo backend (CPU rehearsal) keeps the torch-side gatherer of tiles.py.
    native = G and args.backend == "nccl" and os.environ.get("RT64_BENCH_NATIVE_GATHER", "1") != "0"
    gather = None
    if native:
        uid = torch.zeros(rt64.GATHER_ID_BYTES, dtype=torch.uint8)
        if rank == 0 and not lib.GetGatherUniqueId(uid.data_ptr(), uid.numel()):
            raise SystemExit("RT64_GetGatherUniqueId: " + lib.last_error())
        if N > 1:
            uid_dev = uid.to(comm_device)
            dist.broadcast(uid_dev, 0)
            uid = uid_dev.cpu()
        # frames with GI + denoiser: contiguous bands of about equal modelled cost (cut from the whole frame every rank has just rendered)
        gather = lib.CreateGather(scene.device, uid.data_ptr(), uid.numel(), rank, N, 2 if use_bands else 0)       # also sets this device's share of the frame
        # every rank has to have it, or none uses it: a rank without RCCL behind the library falls back to the torch.distributed gatherer
        # of tiles.py (same layout, same pipelining) -- together with all the others, and the line says which one ran
        ok = torch.tensor([1 if gather else 0], dtype=torch.int32, device=comm_device)
        if N > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not int(ok.item()):
            print("bench.py: rank %d: RT64_CreateGather unavailable on some rank (%s): torch.distributed gather instead" % (rank, lib.last_error() if not gather else "ok here"), file=sys.stderr)
            if gather:
                lib.DestroyGather(gather)
            gather, native = None, False
            if use_bands:
                scene.set_tile(*tiles.band_range(H, rank, N))
            else:
                scene.set_interleave(rank, N)

    # N > 1: frames are ENQUEUED (sync_present = 0) on the renderer's stream, the strips are copied into a gather slot on the same
    # stream, and the RCCL gather of frame k runs beside the rendering of frame k+1 (two slots).  Everything is complete at the
    # closing barrier + synchronize, which is inside the timed region.
    pipelined = G and args.backend == "nccl" and os.environ.get("RT64_BENCH_PIPELINE", "1") != "0"
    if pipelined and not native and torch_backend != "nccl":
        pipelined = False                    # the torch.distributed gatherer of tiles.py pipelines on torch's RCCL communicator only; on gloo it is the CPU-staged rehearsal
    ext_stream = None
    if pipelined and not native:
        try:
            ext_stream = torch.cuda.ExternalStream(lib.GetDeviceStream(scene.device))
        except Exception as e:       # keep measuring: synchronous frames + blocking gather (same result, no overlap)
            print("bench.py: renderer stream not usable from torch (%r): synchronous gather" % (e,), file=sys.stderr)
            pipelined = False
    if ext_stream is not None:
        scene.option("overlap_frames", 0)    # torch orders its copies behind the frames on ONE stream of the library's: keep every frame on it
    gatherer = tiles.FrameGatherer(H, W, rank, N, comm_device, stream=ext_stream, bands=use_bands) if (G and not native) else None
    staging = torch.zeros(max(tiles.strips_per_rank(H, N) * 16, tiles.band_rows(H, N)) * W * 4, dtype=torch.uint8, device="cuda") if (G and not pipelined) else None
    local = torch.zeros(max(tiles.max_owned_rows(H, N), 1) * W * 4, dtype=torch.uint8, device="cuda")
    halo_mode = None
    if native and use_bands:
        band_starts = (C.c_int * (N + 1))()
        if lib.GetGatherBands(gather, band_starts, N + 1) != N:
            raise SystemExit("RT64_GetGatherBands failed")
        my_rows = band_starts[rank + 1] - band_starts[rank]
        # GI + SVGF bands: the filter input of the 62 halo rows travels between neighbouring bands over the gather's communicator (grouped
        # ncclSend / ncclRecv in the middle of the frame) instead of being re-rendered by every band; --halo recompute keeps the re-rendering
        halo_mode = args.halo
        if args.halo == "exchange" and N > 1:
            scene.option("halo_exchange", 1)
        # The modelled cut does not know which rows are expensive (the band over the sphere costs 1.5 x a band over the floor: tools/band_costs.py).  Feedback: every
        # rank times the GPU work of its own band -- frames drawn synchronously, nothing submitted, and with the exchange in dry-run mode so that no rank waits for a
        # neighbour inside the frame --, the figures are shared, every rank computes the same new boundaries and hands them to its gather.
        rebalance_log = []
        for rnd in range(args.band_rebalance if N > 1 else 0):
            if halo_mode == "exchange":
                scene.option("halo_dry_run", 1)
            for _ in range(10):
                scene.draw()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(30):
                scene.draw()
            mine = torch.tensor([(time.perf_counter() - tb) * 1e3 / 30], dtype=torch.float32, device=comm_device)
            scene.option("halo_dry_run", 0)
            every = [torch.zeros_like(mine) for _ in range(N)]
            dist.all_gather(every, mine)
            ms = (C.c_float * N)(*[float(t.item()) for t in every])
            new_starts = (C.c_int * (N + 1))()
            if not lib.RebalanceGatherBands(H, N, band_starts, ms, new_starts) or not lib.SetGatherBands(gather, new_starts):
                raise SystemExit("band rebalancing failed: " + lib.last_error())
            rebalance_log.append({"ms_per_rank": [round(float(v), 4) for v in ms], "starts": [int(v) for v in band_starts]})
            band_starts = new_starts
        my_rows = band_starts[rank + 1] - band_starts[rank]
    elif native:
        my_rows = lib.GatherOwnedRows(H, N, 0, rank)
    my_bytes = (gatherer.owned_bytes() if gatherer else my_rows * W * 4) if G else H * W * 4
    if PR > 1:
        my_bytes = ((tiles.band_range(H, 0, PR)[1] - tiles.band_range(H, 0, PR)[0]) if (args.gi_samples > 0 and args.denoiser) else tiles.owned_rows(H, 0, PR)) * W * 4

    # Direct gather (RT64_SetGatherDirect): every decision below is taken by all ranks together (MIN over ranks of a flag), so that no rank is left in the other mode.
    gather_mode = "rccl" if native else None
    gather_probe = None
    if native and (args.gather == "direct" or (args.gather == "auto" and not use_bands)):
        def agree(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=comm_device)
            if N > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        def gathered_sum():
            scene.draw()
            sl = lib.SubmitGather(gather)
            buf = np.zeros(H * W * 4, dtype=np.uint8)
            got = lib.ReadbackGather(gather, sl, buf.ctypes.data, buf.nbytes, 0)
            return (sl >= 0 and (rank != 0 or got == buf.nbytes)), (int(buf.astype(np.int64).sum()) if rank == 0 else 0)
        handle = torch.zeros(64, dtype=torch.uint8)
        have = True
        if rank == 0:
            have = lib.GetGatherDirectHandle(gather, handle.data_ptr(), handle.numel()) == 64
        if N > 1:
            hd = handle.to(comm_device); dist.broadcast(hd, 0); handle = hd.cpu()
        def pipelined_ms(frames=60):
            """ms per frame of the gather loop as the timed region runs it (frames enqueued, one wait at the end), MAX over ranks"""
            scene.option("sync_present", 0)
            for _ in range(10):
                scene.draw(); lib.SubmitGather(gather)
            torch.cuda.synchronize()
            if N > 1:
                dist.barrier()
            t0 = time.perf_counter()
            sl = -1
            for _ in range(frames):
                scene.draw(); sl = lib.SubmitGather(gather)
            buf = np.zeros(H * W * 4, dtype=np.uint8)
            lib.ReadbackGather(gather, sl, buf.ctypes.data, buf.nbytes, 0)
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) * 1e3 / frames], dtype=torch.float64, device=comm_device)
            if N > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            scene.option("sync_present", 1)
            return float(t.item())
        ok_a, sum_a = gathered_sum()                     # the same static frame through the RCCL exchange first: the reference of the self-check
        gather_probe = None
        if agree(have and ok_a):
            ms_rows = pipelined_ms()
            switched = lib.SetGatherDirect(gather, handle.data_ptr(), handle.numel(), 1) == 1
            if agree(switched):
                ok_b, sum_b = gathered_sum()
                good = agree(ok_b and (rank != 0 or sum_a == sum_b))
                ms_direct = pipelined_ms() if good else None
                gather_probe = {"rows_ms_per_frame": round(ms_rows, 5), "direct_ms_per_frame": (round(ms_direct, 5) if ms_direct is not None else None), "same_bytes": bool(good)}
                if good and args.gather == "auto" and ms_direct > ms_rows:          # correct, but slower on this machine (every rank holds the same two MAX-over-ranks figures: same decision)
                    print("bench.py: rank %d: direct gather is correct here but slower (%.4f against %.4f ms per frame): RCCL exchange of the rows" % (rank, ms_direct, ms_rows), file=sys.stderr)
                    lib.SetGatherDirect(gather, None, 0, 0)
                elif good:
                    gather_mode = "direct"
                else:
                    print("bench.py: rank %d: direct gather self-check failed (%s, %d vs %d): RCCL exchange of the rows instead" % (rank, lib.last_error(), sum_a, sum_b), file=sys.stderr)
                    lib.SetGatherDirect(gather, None, 0, 0)
            else:
                print("bench.py: rank %d: RT64_SetGatherDirect unavailable on some rank (%s): RCCL exchange of the rows instead" % (rank, lib.last_error() if not switched else "ok here"), file=sys.stderr)
                if switched:
                    lib.SetGatherDirect(gather, None, 0, 0)
        if args.gather == "direct" and gather_mode != "direct":
            raise SystemExit("bench.py: --gather direct was asked for and is not available: " + lib.last_error())

    def fetch(dst):
        n = lib.CopyDeviceImage(scene.device, rt64.IMAGE_FINAL_RGBA8, dst.data_ptr(), dst.numel())
        if n != my_bytes:
            raise RuntimeError("RT64_CopyDeviceImage returned %d, expected %d: %s" % (n, my_bytes, lib.last_error()))

    exit_code = [0]
    frame_no = [0]
    step_no = [0]
    packed = [None]              # None: not probed yet; True: frames write the gather slot themselves (RT64_SetDeviceGatherTarget); False: RT64_CopyDeviceImage after each frame

    def step():
        if anim is not None:
            frame_no[0] += 1
            scene.set_mesh(scene.meshes[0], anim[frame_no[0] % len(anim)], data.meshes[0].indices)
        if not G:
            scene.draw()             # returns after the frame is complete in the device's back buffer (HBM): nothing to gather
            return None
        if native:                   # the frame's last kernel writes the rank's packed rows into the slot's send buffer; the exchange runs beside the next frame
            scene.draw()
            slot = lib.SubmitGather(gather)
            if slot < 0:
                raise RuntimeError("RT64_SubmitGather: " + lib.last_error())
            return slot
        slot = step_no[0] % 2
        step_no[0] += 1
        gatherer.wait(slot)          # the gather that last read this slot (two frames ago) is ordered before the refill
        if pipelined and packed[0] is not False:
            lib.SetDeviceGatherTarget(scene.device, gatherer.local(slot).data_ptr(), gatherer.local(slot).numel())     # the frame's last kernel fills the send buffer itself
        scene.draw()
        if pipelined and packed[0] is None:      # first frame: does this kind of frame honour the gather target?  (reading the stats waits for the frame)
            packed[0] = bool(scene.stats().packedFinal)
            if not packed[0]:
                lib.SetDeviceGatherTarget(scene.device, None, 0)
        if pipelined and packed[0]:
            pass
        elif pipelined:
            fetch(gatherer.local(slot))
        else:                        # gloo rehearsal: CPU-staged
            fetch(staging)
            gatherer.local(slot).copy_(staging[:gatherer.local(slot).numel()])
        gatherer.submit(slot)
        return slot

    def barrier():
        torch.cuda.synchronize()
        if G:
            dist.barrier()
        torch.cuda.synchronize()

    # --- instrumented frame (untimed): ray / node / triangle counts of this rank's strips ---
    if args.always_rebuild:
        scene.option("always_rebuild", 1)
    for kv in args.option:
        k, _, v = kv.partition("=")
        scene.option(k, float(v))
    if os.environ.get("RT64_LDS_CACHE"):
        scene.option("lds_cache", int(os.environ["RT64_LDS_CACHE"]))
    if os.environ.get("RT64_FUSED_LEAN"):
        scene.option("fused_lean", int(os.environ["RT64_FUSED_LEAN"]))
    if os.environ.get("RT64_BOUNCE_REFILL"):
        scene.option("bounce_refill", int(os.environ["RT64_BOUNCE_REFILL"]))
    scene.option("count_traversal", 1)
    step()
    st = scene.stats()
    counts = dict(primary=st.primaryRays, shadow=st.shadowRays, indirect=st.indirectRays, reflection=st.reflectionRays + st.refractionRays, nodesPrimary=st.nodesPrimary,
                  trisPrimary=st.trianglesPrimary, nodesDirect=st.nodesDirect, trisDirect=st.trianglesDirect,
                  nodesIndirect=st.nodesIndirect, trisIndirect=st.trianglesIndirect)
    lean = bool(st.leanFrame)
    fused = st.fusedFrame == 1
    fused_full = st.fusedFrame == 2
    hit_pixels = int((scene.readback(rt64.IMAGE_FIRST_INSTANCE_ID) >= 0).sum())
    scene.option("count_traversal", 0)

    if pipelined or PR > 1:
        scene.option("sync_present", 0)      # RT64_DrawDevice enqueues; ordering is on the renderer's stream from here on
    # Before the W warm-up steps: the same number of untimed frames on every rank, so that a short run (the driver's --steps 20 --warmup 5 is
    # 4 ms of GPU work) is not timed on clocks that are still ramping -- measured: 0.174 ms per frame cold against 0.164 after ~100 frames.
    for _ in range(args.prewarm):
        step()
    for _ in range(args.warmup):
        step()
    acc = dict(trace=0.0, shade=0.0, direct=0.0, indirect=0.0, compose=0.0, build=0.0, total=0.0, denoise=0.0, reflect=0.0)

    def add_stats():
        s = scene.stats()
        acc["trace"] += s.msPrimaryTrace; acc["shade"] += s.msPrimaryShade; acc["direct"] += s.msDirect
        acc["indirect"] += s.msIndirect; acc["compose"] += s.msComposePost; acc["build"] += s.msBuild; acc["total"] += s.msTotal
        acc["denoise"] += s.msDenoise; acc["reflect"] += s.msReflectRefract
    barrier()
    events_every = args.pass_events_every if args.pass_events_every > 0 else (1 if (fused and not args.gi_samples) else 8)
    scene.option("profile_every", events_every)
    scene.option("reset_accum", 1)   # the library sums the HIP-event timings of every sampled frame from here on; read once after the loop
    t0 = time.perf_counter()
    last_slot = None
    for _ in range(args.steps):
        last_slot = step()
    enqueue_ms = (time.perf_counter() - t0) * 1e3 / args.steps      # host time per step before the closing barrier (= frame time when frames are synchronous)
    if G and not native:
        for sl in (0, 1):
            gatherer.wait(sl)            # the last two gathers (and rank 0's assembly of them) complete inside the timed region
    barrier()                            # (torch.cuda.synchronize waits for every stream of the device, the library's two included)
    elapsed = time.perf_counter() - t0
    stat_frames = args.steps
    gathered_checksum = None
    if native:                       # every rank waits for its part of the last exchange; rank 0 reads the assembled frame
        host_frame = np.zeros(H * W * 4, dtype=np.uint8)
        got = lib.ReadbackGather(gather, last_slot, host_frame.ctypes.data, host_frame.nbytes, 0)
        if rank == 0:
            if got != host_frame.nbytes:
                raise RuntimeError("RT64_ReadbackGather: " + lib.last_error())
            gathered_checksum = int(host_frame.astype(np.int64).sum())
    elif G and rank == 0:            # the frame assembled on rank 0 by the gather of the last timed step
        gathered_checksum = int(gatherer.frame(last_slot).to(torch.int64).sum().item())
    if not G and PR <= 1:            # N = 1: every timed frame was measured live with HIP events inside the library
        s = scene.stats()
        stat_frames = (args.steps + events_every - 1) // events_every
        assert s.accumFrames == stat_frames, (s.accumFrames, stat_frames)
        acc.update(trace=s.accumMsPrimaryTrace, shade=s.accumMsPrimaryShade, direct=s.accumMsDirect, indirect=s.accumMsIndirect, compose=s.accumMsComposePost,
                   build=s.accumMsBuild, total=s.accumMsTotal, denoise=s.accumMsDenoise, reflect=s.accumMsReflectRefract)
    # The reference uploads its tables and rebuilds the TLAS every frame (rt64_view.cpp:451 updateOnly = false, :1150-1152); the timed
    # loop above ran with the frame-table cache (identical descriptors are not re-uploaded).  Same K steps again with the cache off,
    # outside the timed region of `value`, so that the line carries both figures.
    rebuild = None
    if not G and PR <= 1 and not args.always_rebuild and not args.timed_loop_only:
        scene.option("always_rebuild", 1)
        for _ in range(min(args.warmup, 5)):
            step()
        barrier()
        scene.option("reset_accum", 1)
        tr = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        r_ms = (time.perf_counter() - tr) * 1e3 / args.steps
        sr = scene.stats()
        rebuild = {"ms_per_step": round(r_ms, 5), "value": round(rays_total / (r_ms * 1e-3) / 1e6, 2), "build_ms": round(sr.accumMsBuild / max(sr.accumFrames, 1), 5),
                   "what": "always_rebuild=1: frame tables uploaded + TLAS rebuilt + raster lists re-staged every frame, as the reference does (rt64_view.cpp:451,1150-1152)"}
        scene.option("always_rebuild", 0)
    # ... and with RT64_DrawDevice only enqueueing (sync_present = 0, what the N > 1 path runs): K frames back to back on the stream, one wait
    # at the end.  `value` above keeps the reference's frame-by-frame wait (rt64_device.cpp:1006-1025); this is the same work without the
    # launch-to-completion round trip between frames.
    enqueued = None
    if not G and PR <= 1 and anim is None and not args.timed_loop_only:
        scene.option("sync_present", 0)
        for _ in range(min(args.warmup, 5)):
            step()
        barrier()
        te = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        e_ms = (time.perf_counter() - te) * 1e3 / args.steps
        enqueued = {"ms_per_step": round(e_ms, 5), "value": round(rays_total / (e_ms * 1e-3) / 1e6, 2),
                    "what": "sync_present=0: RT64_DrawDevice enqueues, frames run back to back on the library's stream, one wait after the K-th (cached tables)"}
        scene.option("sync_present", 1)
        step(); barrier()
    # ... and what a game does: an instance moves in EVERY frame, so the tables, the TLAS and the head of the scene-cache image really change (always_rebuild above re-sends
    # identical bytes).  Changed tables go into another of the view's table slots, so such frames stay one-kernel lean frames and -- enqueued -- stay side by side.
    moving = None
    if not G and PR <= 1 and anim is None and not args.timed_loop_only and not (args.gi_samples or args.denoiser):
        import copy as _copy
        k_sphere = next((i for i, inst in enumerate(data.instances) if inst.name == "sphere"), -1)
        if k_sphere >= 0:
            descs = []
            for dx in (0.0, 0.02):
                inst = _copy.copy(data.instances[k_sphere])
                t = np.array(inst.transform, dtype=np.float32).copy(); t[3][0] += dx
                inst.transform = t; inst.previous_transform = t
                descs.append(scene._instance_desc(inst))
            handle = scene.instances[k_sphere]
            moving = {"what": "the sphere's transform changes in every frame (two positions in turn): frame tables + TLAS + scene-cache head uploaded every frame into the next table slot"}
            for name, sync in (("sync", 1), ("enqueued", 0)):
                scene.option("sync_present", sync)
                for f in range(min(args.warmup, 5) + 2):
                    lib.SetInstanceDescription(handle, descs[f % 2]); lib.DrawDevice(scene.device, 1, 16.0)
                barrier()
                tm = time.perf_counter()
                for f in range(args.steps):
                    lib.SetInstanceDescription(handle, descs[f % 2]); lib.DrawDevice(scene.device, 1, 16.0)
                barrier()
                m_ms = (time.perf_counter() - tm) * 1e3 / args.steps
                stm = scene.stats()
                moving[name] = {"ms_per_step": round(m_ms, 5), "value": round(rays_total / (m_ms * 1e-3) / 1e6, 2), "lean_frame": int(stm.leanFrame), "one_kernel": int(stm.fusedFrame == 1), "overlapped": int(stm.overlappedFrame)}
            scene.option("sync_present", 1)
            lib.SetInstanceDescription(handle, descs[0]); step(); barrier()
    scene.option("profile_every", 1)
    if G or PR > 1:                  # per-kernel timings of this rank's strips from a few untimed frames (reading them synchronises)
        stat_frames = 10
        for _ in range(stat_frames):
            step()
            add_stats()
        barrier()
    if N > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = rays_total / (elapsed / args.steps) / 1e6

    if rank == 0:
        K = float(stat_frames)
        kms = {k: v / K for k, v in acc.items()}
        my_pixels = my_bytes // 4 if native else (tiles.band_range(H, 0, N)[1] if use_bands else tiles.owned_rows(H, 0, N)) * W
        if PR > 1:
            my_pixels = my_bytes // 4
        kernels = {
            "primary_trace": (kms["trace"], my_pixels * PRIMARY_TRACE_PIXEL_B + NODE_B * counts["nodesPrimary"] + TRI_B * counts["trisPrimary"]),
            "primary_shade": (kms["shade"], my_pixels * (PRIMARY_SHADE_PIXEL_LEAN_B if lean else PRIMARY_SHADE_PIXEL_B) + hit_pixels * (PRIMARY_SHADE_HIT_B + (PRIMARY_SHADE_HIT_LEAN_B if lean else 0)) + (my_pixels - hit_pixels) * PRIMARY_SHADE_MISS_B),
            "direct": (kms["direct"], my_pixels * (DIRECT_PIXEL_B + (0 if lean else 8)) + hit_pixels * DIRECT_HIT_B + NODE_B * counts["nodesDirect"] + TRI_B * counts["trisDirect"] + 4 * counts["shadow"]
                       + (my_pixels * COMPOSE_PIXEL_LEAN_B if lean else 0)),          # lean frame: direct_kernel<false> composes the pixel itself
        }
        if fused_full:           # full frame, all instances opaque: primary visibility + G-buffer + DirectRayGen are one kernel (same bytes as the three it replaces, minus the hit-record hand-over)
            tb, sb, db = kernels.pop("primary_trace")[1], kernels.pop("primary_shade")[1], kernels.pop("direct")[1]
            kernels["full_frame(trace+gbuffer+direct)"] = (kms["trace"], tb + sb + db - my_pixels * 2 * PRIMARY_TRACE_PIXEL_B + my_pixels * PRIMARY_TRACE_PIXEL_B)
        if fused:
            # one kernel carries the pixel from the primary ray to the back buffer: what it has to move is the hit record it keeps for
            # on-demand G-buffer rebuilds, the direct-light accumulation, the back buffer, the vertex / texel operands of the any-hit
            # program and the BVH records its rays visit (served from the LDS scene cache on this scene -- see `traffic` for the HBM bytes)
            kernels = {"lean_frame(trace+shade+direct+compose)": (kms["trace"],
                       my_pixels * (PRIMARY_TRACE_PIXEL_B + 8 + 4) + hit_pixels * PRIMARY_SHADE_HIT_B + (my_pixels - hit_pixels) * PRIMARY_SHADE_MISS_B
                       + NODE_B * (counts["nodesPrimary"] + counts["nodesDirect"]) + TRI_B * (counts["trisPrimary"] + counts["trisDirect"]) + 4 * counts["shadow"])}
        if not lean:
            kernels["compose_post"] = (kms["compose"], my_pixels * COMPOSE_PIXEL_B)
        else:
            kernels["raster_fg"] = (kms["compose"], 0)                                  # what is left after the last ray pass: the HUD draw
        if args.gi_samples:
            kernels["indirect"] = (kms["indirect"], my_pixels * 12 + hit_pixels * 24 + NODE_B * counts["nodesIndirect"] + TRI_B * counts["trisIndirect"])
        if args.denoiser:
            # SVGF: variance pass (moments 8 + GI 8 + normal 8 + depth 4 read, 8 written) + 5 a-trous iterations (GI 8 + normal 8 + depth 4 read, 8 written)
            kernels["svgf_denoise(6 launches)"] = (kms["denoise"], my_pixels * (36 + 5 * 28))
        if counts["reflection"]:
            kernels["reflection_refraction"] = (kms["reflect"], 0)
        dominant = max((k for k in kernels if kernels[k][1] > 0), key=lambda k: kernels[k][0])
        d_ms, d_bytes = kernels[dominant]
        roofline = roofline_object(dominant, d_ms, d_bytes, kernels)
        roofline.update({"lean_frame": lean, "fused_frame": fused, "frame_gpu_ms": round(kms["total"], 5), "build_ms": round(kms["build"], 5),
                         "nodes_per_primary_ray": round(counts["nodesPrimary"] / max(counts["primary"], 1), 3),
                         "tris_per_primary_ray": round(counts["trisPrimary"] / max(counts["primary"], 1), 3),
                         "nodes_per_shadow_ray": round(counts["nodesDirect"] / max(counts["shadow"], 1), 3),
                         "tris_per_shadow_ray": round(counts["trisDirect"] / max(counts["shadow"], 1), 3)})
        stress = bool(args.subdiv or args.floor_grid > 1)
        if args.config == "C2" and not stress and not (args.gi_samples or args.denoiser) and (W, H) == (1920, 1080):
            metric = "Mrays/s (primary+shadow), sample scene 1080p 1spp"                      # BASELINE.json's metric, on the config it is quoted on
        else:
            metric = "Mrays/s (primary+shadow%s%s), %s %dx%d, %d primary sample%s per pixel%s" % (
                "+GI bounce" if args.gi_samples else "", "+reflection" if counts["reflection"] else "", "stress variant of the sample scene" if stress else "sample scene", W, H,
                ext["primary_spp"], "" if ext["primary_spp"] == 1 else "s",
                (", %d GI sample%s per %s (%d bounce%s each)%s" % (args.gi_samples, "s" if args.gi_samples > 1 else "", "pixel" if ext["primary_spp"] == 1 else "primary sample", ext["gi_bounces"],
                                                                 "" if ext["gi_bounces"] == 1 else "s", " + SVGF" if args.denoiser else "")) if args.gi_samples else "")
        result = {
            "metric": metric, "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: src/sample scene, primary+shadow rays + shading + compose, %dx%d %dspp%s" % (
                args.config, W, H, ext["primary_spp"], "" if not (args.gi_samples or args.subdiv or args.floor_grid > 1) else " [gi=%d denoiser=%d subdiv=%d floor_grid=%d]" % (
                    args.gi_samples, int(args.denoiser), args.subdiv, args.floor_grid)),
                "rays_per_frame": int(rays_total), "width": W, "height": H,
                "partition": ("DIAGNOSIS: rank 0's share of a %d-way partition only, no gather" % PR) if PR > 1 else ("%s x%d + RCCL gather of RGBA8" % (("cost-balanced contiguous bands %s with denoiser halo" % (list(band_starts) if native else "")) if use_bands else "interleaved 16-row strips", N)) if N > 1 else "single GPU"},
            "roofline": roofline,
        }
        if args.config in sample_scene.BENCH_DEVIATIONS:       # how the configuration reads BASELINE.json's wording ("deviation"), or which extensions run it as worded
            result["config"]["extensions" if args.config.endswith("-literal") else "deviation"] = sample_scene.BENCH_DEVIATIONS[args.config]
        if N > 1 and native and use_bands and rebalance_log:
            result["band_rebalance"] = rebalance_log
        result["prewarm_frames"] = args.prewarm
        result["pass_events"] = "HIP events around every pass of every timed frame" if events_every == 1 else (
            "per-pass HIP events on every %dth timed frame (%d of %d frames sampled): each event is a barrier packet of ~5 us on the stream" % (events_every, stat_frames, args.steps))
        if args.option:
            result["config"]["options"] = list(args.option)      # non-default library options: an A/B line, not the headline
        if G:
            result["pipeline"] = {"frames": "enqueued (sync_present=0), 2 gather slots" if pipelined else "synchronous, CPU-staged gather (rehearsal)",
                                  "gather": (("in-library, direct (RT64_SetGatherDirect: the frame kernels store their rows into rank 0's frame slots through an IPC mapping -- peer stores over xGMI --, a 4-byte token per rank through RCCL; self-checked against the RCCL exchange)"
                                              if gather_mode == "direct" else "in-library (RT64_SubmitGather: grouped ncclSend / ncclRecv + reassembly kernel on the library's comm stream)") if native else "torch.distributed gather (tiles.FrameGatherer)"),
                              "denoiser_halo": (("exchanged between neighbouring bands (ncclSend / ncclRecv of 24 B per pixel, 62 rows per side)" if halo_mode == "exchange" else "re-rendered by every band (66 rows per side)") if halo_mode else "none (pixel-local frame)"),
                              "send_buffer": ("none: rows stored once, into the frame slot" if gather_mode == "direct" else ("written by the frame kernel (RT64_SetDeviceGatherTarget)" if (packed[0] or (native and scene.stats().packedFinal)) else "packed after each frame (RT64_CopyDeviceImage layout)")),
                                  "gather_probe": gather_probe,
                                  "control_plane": "torch.distributed on %s (%s)" % (torch_backend, "one RCCL communicator per process: the library's" if (native and torch_backend == "gloo") else
                                                                                     ("a second RCCL communicator beside the library's" if native else "torch's gatherer")),
                                  "host_ms_per_step": round(enqueue_ms, 5)}
        result["accel_build"] = {"first_frame_ms": round(first_frame_build_ms, 4), "triangles": int(st_full.triangleCount), "blas_node_bytes": int(st_full.blasNodeBytes),
                                 "what": "GPU time of all BLAS builds (LBVH: Morton, radix sort, Karras, fit) + the TLAS build, executed at the first frame after the RT64_SetMesh calls"}
        if enqueued is not None:
            if roofline.get("traffic") and fused:
                # the same HBM bytes per frame at the rate frames complete when they are only enqueued (consecutive pixel-local frames overlap on the library's render
                # streams, so a launch no longer ends with an idle chip behind its longest ray): what the memory system sustains per frame, not per launch
                gbps = roofline["traffic"] / (enqueued["ms_per_step"] * 1e-3) / 1e9
                enqueued["hbm"] = {"bytes_per_frame": int(roofline["traffic"]), "GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBS, 4),
                                   "note": "PMC bytes of one launch / enqueued ms per frame / 8 TB/s (launches overlap: per-frame throughput, not a per-launch figure)"}
            result["enqueued_frames"] = enqueued
        if rebuild is not None:
            result["always_rebuild"] = rebuild
        if moving is not None:
            result["moving_instance"] = moving
        result["frame_tables"] = "rebuilt every frame (always_rebuild)" if args.always_rebuild else "cached while the host re-sends identical descriptors (steady state of the sample host, main.cpp:97-134); `always_rebuild` holds the figure with the cache off"
        oracle_frame = None
        if N == 1 and not args.no_cpu_baseline and not args.timed_loop_only:
            result["cpu_baseline"], oracle_frame = cpu_baseline(data, W, H, args.cpu_baseline_height, keep_frame=(args.config == "C2" and not args.gi_samples))
        if N == 1 and not G and PR <= 1 and not args.no_parity and not args.timed_loop_only:
            # Outside the timed region: the measured configuration at its full size against the oracle (BASELINE.json gate: RMSE <= 1e-3 on the
            # composed RGBA32F image; hit records are bit-exact).  A line whose frame is wrong is not a measurement: exit non-zero.
            result["parity"] = parity_object(lib, scene, data, args, W, H, local_rank, oracle_frame)
        if not G:
            fetch(local)
            frame = local[:H * W * 4]
            result["frame_checksum"] = int(frame.to(torch.int64).sum().item())
        else:
            result["frame_checksum"] = gathered_checksum
        os.write(json_fd, (json.dumps(result) + "\n").encode())
        if result.get("parity") and not result["parity"]["pass"]:
            print("bench.py: PARITY FAILED against the oracle at %dx%d: %s" % (W, H, json.dumps(result["parity"])), file=sys.stderr)
            exit_code[0] = 3

    if gatherer:
        gatherer.close()                 # torch's current stream goes back to the default one before the renderer's stream is destroyed
    if gather:
        lib.DestroyGather(gather)
    scene.close()
    if G:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code[0]:
        raise SystemExit(exit_code[0])


VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector peak = 1024 SIMD-32s x 32 lanes x 2 flop x 2.4 GHz, i.e. ONE wave64 instruction per SIMD every 2 cycles
SIMDS, CLOCK_HZ = 1024, 2.4e9


def source_hash():
    """Identity of the kernel sources a counter profile belongs to (the GPU box has no .git): sha256 over csrc/."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sm64rt-legacy-renderer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp", ".inc")) or name == "Makefile":
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def load_counters(workload):
    """profiles/kernel_counters.json (tools/collect_counters.py: rocprofv3 PMC passes of this workload) -- only if it was collected from
    exactly these kernel sources; a stale file is refused (returns None and the reason)."""
    path = os.path.join(ROOT, "profiles", "kernel_counters.json")
    if not os.path.exists(path):
        return None, "no profiles/kernel_counters.json"
    try:
        doc = json.load(open(path))
    except Exception as e:
        return None, "unreadable profiles/kernel_counters.json (%r)" % (e,)
    if doc.get("source_hash") != source_hash():
        return None, "profiles/kernel_counters.json was collected from other kernel sources (%s, these are %s): refused" % (doc.get("source_hash"), source_hash())
    w = doc.get("workloads", {}).get(workload)
    if not w:
        return None, "profiles/kernel_counters.json has no counters for workload %s" % workload
    return w, None


COUNTER_WORKLOAD = [None]
# bench.py's kernel groups -> (profile kernel, launches per frame)
GROUPS = {"indirect": [("bounce_trace", 1), ("bounce_hit", 1), ("bounce_miss", 1), ("bounce_resolve", 1)],        # bounce_miss: only behind the refill walk
          "_optional": {"bounce_miss"},
          "svgf_denoise": [("svgf_guide", 1), ("svgf_variance", 1), ("svgf_atrous", 5)], "reflection_refraction": [("reflection", 2)]}


def group_counters(kernels, key):
    """Counters of one bench.py kernel group per frame: a single profiled kernel, or the sum over the launches the group stands for."""
    if key in kernels:
        return kernels[key]
    if key not in GROUPS or any(k not in kernels and k not in GROUPS["_optional"] for k, _ in GROUPS[key]):
        return None
    out = {}
    for k, n in GROUPS[key]:
        for c, v in kernels.get(k, {}).items():
            if c.startswith("SQ_") or c in ("hbm_bytes", "fetch_bytes_x2", "write_bytes", "avg_ns"):
                out[c] = out.get(c, 0.0) + n * v
    return out


def roofline_object(dominant, d_ms, d_bytes, kernels):
    """What binds the dominant kernel, from counters: `frac` is the utilisation of the binding resource (<= 1 by construction) --
    VALU issue slots (SQ_INSTS_VALU wave-instructions x 2 cycles each at full rate, over 1024 SIMDs x the launch duration measured live
    in this run) or HBM (PMC bytes / the same duration / 8 TB/s), whichever is larger.  The algorithmic bytes of SURVEY 8(d) (every BVH
    record a ray visits, wherever it is served from) stay in the line under their own name and are never called a roofline fraction."""
    key = dominant.split("(")[0]
    secs = d_ms * 1e-3
    alg = {"bytes_per_launch": int(d_bytes), "GBps": round(d_bytes / secs / 1e9, 2) if secs > 0 else 0.0,
           "note": "SURVEY 8(d) accounting: 64 B per node visit + 48 B per triangle test + per-pixel records, counted whether HBM, L2 or the LDS scene cache serves them -- not HBM traffic, not a roofline fraction"}
    w, why = load_counters(COUNTER_WORKLOAD[0])
    c = group_counters((w or {}).get("kernels", {}), key)
    out = {"kernel": dominant, "ms_per_launch": round(d_ms, 5), "algorithmic": alg,
           "kernels": {k: {"ms": round(v[0], 5), "alg_bytes": int(v[1])} for k, v in kernels.items()}}
    if not c or secs <= 0:
        out.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "note": "no counter profile for this build (%s): the binding resource is not claimed" % (why or "kernel %s not in the profile" % key)})
        return out
    hbm_bytes = c["hbm_bytes"]
    hbm_gbps = hbm_bytes / secs / 1e9
    hbm_frac = hbm_gbps / HBM_PEAK_GBS
    valu_tflops = c["SQ_INSTS_VALU"] * 128.0 / secs / 1e12          # one wave64 instruction = 64 lanes x 2 flop of issue capacity (FMA-equivalent)
    valu_frac = valu_tflops / VALU_PEAK_TFLOPS
    detail = {"hbm": {"bytes_per_launch": int(hbm_bytes), "fetch_bytes_x2": int(c["fetch_bytes_x2"]), "write_bytes": int(c["write_bytes"]), "GBps": round(hbm_gbps, 1), "frac": round(hbm_frac, 4),
                      "frac_of_measured_copy_roof_6290GBps": round(hbm_gbps / 6290.0, 4)},
              "valu": {"wave_insts_per_launch": int(c["SQ_INSTS_VALU"]), "issue_frac": round(valu_frac, 4),
                       "active_quad_cycles_per_launch": int(c.get("SQ_ACTIVE_INST_VALU", 0)),
                       "active_frac": round(c.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / (SIMDS * secs * CLOCK_HZ), 4),
                       "note": "issue_frac prices every VALU wave-instruction at 2 cycles (the SIMD-32 full rate, reachable with >= 2 waves per SIMD); active_frac = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x duration x 2.4 GHz) is the counter's own busy figure (a wave holds the VALU 4 cycles per instruction) and can overlap between waves"},
              "wave": {k: int(c[k]) for k in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES") if k in c},
              "registers": {k: c[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size") if k in c},
              "profile": {"source_hash": source_hash(), "rocprofv3_avg_ns": c.get("avg_ns"), "live_avg_ns": round(d_ms * 1e6, 1)}}
    if valu_frac >= hbm_frac:
        out.update({"bound": "valu", "achieved": round(valu_tflops, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(valu_frac, 4),
                    "note": "VALU issue capacity in fp32 FMA-equivalent TFLOP/s (wave-instructions x 128 / s); the kernel traces rays (integer / compare / select work included), it does not do that many flops"})
    else:
        out.update({"bound": "hbm", "achieved": round(hbm_gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4)})
    out["traffic"] = int(hbm_bytes)
    out["counters"] = detail
    if "SQ_WAVE_CYCLES" in c and c.get("SQ_WAVE_CYCLES"):
        wc = float(c["SQ_WAVE_CYCLES"])
        share = {k: round(c[k] / wc, 3) for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in c}
        out["counters"]["wave"]["share_of_wave_cycles"] = share
        # `bound` names the busier of the two pipes; when neither is busy, what holds the kernel is how long a wave waits, and this says so
        resident = wc * 4.0 / (SIMDS * secs * CLOCK_HZ)          # SQ_WAVE_CYCLES counts quad-cycles: average waves resident per SIMD
        if out["frac"] is not None and out["frac"] < 0.5 and "SQ_WAIT_ANY" in share:
            out["limiter"] = ("latency: a wave spends %.0f %% of its cycles in s_waitcnt (memory) and %.0f %% issuing VALU; %.1f waves per SIMD are resident on average, too few to fill the "
                              "gaps, so neither the VALU (%.0f %%) nor HBM (%.0f %%) is the limit" % (100 * share["SQ_WAIT_ANY"], 100 * share.get("SQ_ACTIVE_INST_VALU", 0), resident, 100 * valu_frac, 100 * hbm_frac))
    return out


def host_threads():
    """Threads the CPU baseline may really use: the affinity mask, cut by the cgroup's CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(path).read().split()
            if path.endswith("cpu.max"):
                if t[0] != "max":
                    n = min(n, max(1, int(float(t[0]) / float(t[1]) + 0.5)))
            else:
                q = int(t[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(data, W, H, rows, keep_frame=False):
    """The scalar C oracle (kind "port": same LBVH, same traversal, same shading math) on the host cores."""
    from oracle import oracle_py
    threads = host_threads()            # every host core this process may use (omp_get_max_threads())
    ora = oracle_py.OracleScene(data)
    try:
        tile = (0, rows) if rows and rows < H else None
        # bounded sample: whole frames of the same workload until about 10 s of CPU work have been timed
        frames, secs, rays, build, t0 = 0, 0.0, 0, 0.0, time.perf_counter()
        kept = None
        while secs < 10.0 and frames < 400:
            r = ora.render(W, H, threads=threads, tile=tile)
            if keep_frame and kept is None and tile is None:
                kept = {k: r[k] for k in ("output", "final", "primaryHit")}      # the frame the GPU's measured frame is compared with (parity_object)
            c = r["counters"]
            rays += c["primaryRays"] + c["shadowRays"] + c["indirectRays"]
            secs += c["secondsRender"]; build += c["secondsBuild"]; frames += 1
        dt = time.perf_counter() - t0
        cpu = "unknown CPU"
        try:
            cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except Exception:
            pass
        return {"value": round(rays / secs / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": "%d frames of the same C2 workload at %dx%d (%d rows), %d rays, %.2f s render + %.3f s BVH build; %d OpenMP threads on %s (%d logical CPUs)" % (
                    frames, W, H, (tile[1] if tile else H), rays, secs, build, threads, cpu, os.cpu_count() or 0),
                "ms_per_frame": round(dt * 1e3 / frames, 1)}, kept
    finally:
        ora.close()


def compare_frames(got_out, got_final, got_hit, ref, W, H, what):
    import numpy as np
    d_out = got_out[..., :3].astype(np.float64) - ref["output"][..., :3].astype(np.float64)
    d_fin = got_final[..., :3].astype(np.float64) / 255.0 - ref["final"][..., :3].astype(np.float64) / 255.0
    p = {"size": [W, H], "what": what,
         "rmse_output": float(np.sqrt(np.mean(d_out ** 2))), "rmse_final": float(np.sqrt(np.mean(d_fin ** 2))),
         "max_rgba8_diff": int(np.abs(got_final.astype(np.int32) - ref["final"].astype(np.int32)).max()),
         "hit_mismatches": int((got_hit != ref["primaryHit"]).any(axis=-1).sum()),
         "gate": {"rmse_output": 1e-3, "rmse_final": 1e-3, "hit_mismatches": 0}}
    p["pass"] = bool(p["rmse_output"] <= 1e-3 and p["rmse_final"] <= 1e-3 and p["hit_mismatches"] == 0)
    p["rmse_output"] = float("%.4g" % p["rmse_output"]); p["rmse_final"] = float("%.4g" % p["rmse_final"])
    return p


def parity_object(lib, scene, data, args, W, H, hip_device, oracle_frame):
    """The measured configuration at its BASELINE size against the oracle (test infrastructure used as the checker, outside the timed region).
    C2 is a static frame: the images of the scene that was just timed are compared with the frame the cpu_baseline leg rendered.  The GI
    configurations depend on the frame number (blue-noise slice, history): a fresh scene + oracle pair renders the first F frames of the
    same call sequence (C4: with its per-frame SetMesh) and the F-th frames are compared."""
    from sm64rt_legacy_renderer_amd import rt64, sample_scene
    from oracle import oracle_py
    threads = host_threads()
    if not (args.gi_samples or args.denoiser):
        if oracle_frame is None:
            ora = oracle_py.OracleScene(data)
            try:
                r = ora.render(W, H, threads=threads)
                oracle_frame = {k: r[k] for k in ("output", "final", "primaryHit")}
            finally:
                ora.close()
        got = [scene.readback(i) for i in (rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_PRIMARY_HIT)]
        return compare_frames(got[0], got[1], got[2], oracle_frame, W, H, "the measured frame (static scene) vs the oracle's frame of the cpu_baseline leg")
    F = args.parity_frames or 2
    import copy
    d2 = copy.copy(data)
    d2.meshes = [copy.copy(m) for m in data.meshes]
    anim = sample_scene.c4_animation(d2) if args.config in ("C4", "C4-literal") else None       # (the material edits of C5 are already in `data`)
    ext = {k: sample_scene.BENCH_CONFIGS.get(args.config, {}).get(k, 1) for k in ("primary_spp", "gi_bounces")}
    s2 = sample_scene.Rt64Scene(lib, d2, W, H, hip_device=hip_device)
    ora = oracle_py.OracleScene(d2)
    try:
        s2.set_view_description(gi_samples=args.gi_samples, denoiser=args.denoiser)
        for k, v in ext.items():
            s2.option(k, v)
        for kv in args.option:
            k, _, v = kv.partition("=")
            s2.option(k, float(v))
        for f in range(F):
            if anim is not None:
                v = anim[(f + 1) % len(anim)]
                s2.set_mesh(s2.meshes[0], v, d2.meshes[0].indices); ora.set_mesh(ora.meshes[0], v, d2.meshes[0].indices)
            s2.draw()
            ref = ora.render(W, H, threads=threads, giSamples=args.gi_samples, denoiserEnabled=int(args.denoiser), denoiserMode=1, primarySpp=ext["primary_spp"], giBounces=ext["gi_bounces"])
        got = [s2.readback(i) for i in (rt64.IMAGE_OUTPUT_RGBA32F, rt64.IMAGE_FINAL_RGBA8, rt64.IMAGE_PRIMARY_HIT)]
        return compare_frames(got[0], got[1], got[2], ref, W, H, "frame %d of a fresh scene + oracle pair running this configuration's call sequence" % F)
    finally:
        s2.close(); ora.close()


if __name__ == "__main__":
    main()
