#!/usr/bin/env python3
"""bench.py — frame-render hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N                  (no launcher around it: starts the line above as a child process itself)
    python bench.py --gpus N --mode group     (ONE process, N GPUs behind one handle: RtxConfig.numDevices)

A step = one frame = one sample per pixel over the whole image (the reference's
"N spp" is N accumulated frames). Default workload = BASELINE.json
configs[3] shape: synthetic 10 M-triangle mesh + textured materials, 3840x2160
(rendered at 3840x2176 after the reference's rounding to multiples of 32), scene
replicated on every GPU, screen tiles dealt round-robin to the ranks. Total work is
the same for every N (strong scaling).

What `value` is: rays traced by all ranks / wall time of { K x rtx_render(pixels = NULL) in the library's DEFAULT
traversal mode (RTX_TRAVERSE_WIDE: the reference's result by construction) and default configuration (queued frames are
rendered config.frames_per_batch at a time, config.frames_in_flight batches overlap) ; gather of the tiles to rank 0 (RCCL) ;
one device-to-host copy of the finished frame } — the delivered-batch pattern of BASELINE.json configs[3] (64 spp,
one image). Scene and running mean are resident in HBM when the clock starts. Printed beside it, as top-level keys
of the same line:
  sync_boundary   the reference's own calling pattern — Render(pixels) synchronously every frame, each frame copied
                  to the host (src/WindowManager.cpp:154-158) — measured in a child process that leaves
                  GPU_MAX_HW_QUEUES at HIP's default (PCIe-inclusive; never `value`)
  culled_opt_in   the same frames with distance culling (RTX_TRAVERSE_CULLED, opt-in, not exact by construction)
  roofline        dominant kernel: algorithmic bytes (SURVEY.md §8d constants x device-counted visits) over its HIP-event
                  duration, HBM traffic from the committed rocprofv3 PMC pass if it belongs to this very kernel source
  cpu_baseline    the oracle (CPU restatement of ThreadedShader, static row bands) on a bounded sample, strict and -Ofast

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

WORKLOADS = {
    # name: (triangles, width, height, textured, grid)
    "cfg4_10Mtri_4k": (10_000_000, 3840, 2160, True, None),
    "cfg3_1Mtri_1080p": (1_000_000, 1920, 1080, False, None),
    "cfg5_100Mtri_4k": (100_000_000, 3840, 2160, True, (10, 10)),
    "small_100ktri_992": (100_000, 1000, 1000, True, None),
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def csrc_digest():
    """Identity of the kernel source the committed PMC numbers were taken with."""
    h = hashlib.sha256()
    d = os.path.join(HERE, "raytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def sync_only(args):
    """Child process: the reference's calling pattern. Render(pixels) per frame, synchronous, frame copied to the host
    every time, whatever GPU_MAX_HW_QUEUES the environment has (HIP's default 4 when the parent removed it)."""
    from raytracer_amd import _capi
    from raytracer_amd.renderer import HIPShader
    from raytracer_amd.scene import synthetic_scene
    ntri, w, h, textured, grid = WORKLOADS[args.workload]
    scene = synthetic_scene(ntri, w, h, seed=12345, textured=textured, nthreads=16, grid=grid)
    trav = {"wide": _capi.TRAVERSE_WIDE, "culled": _capi.TRAVERSE_CULLED, "exact": _capi.TRAVERSE_EXACT}[args.traversal]
    res = {}
    for key, look in (("no_look_ahead", -1), ("default", 0), ("page_locked", 0)):
        shader = HIPShader(scene, device=0, traversal=trav, look_ahead=look)
        px = np.zeros((shader.height, shader.width, 4), dtype=np.float32)
        if key == "page_locked":          # a caller that page-locks its pixel array gets the frame by one direct copy (rtx_api.hip deliver_bytes)
            import torch
            pinned = torch.zeros((shader.height, shader.width, 4), dtype=torch.float32).pin_memory()
            px = pinned.numpy()
        warm = 12                               # the loop is recognised after two frames; the sets started ahead grow over the next few
        for i in range(warm):
            shader.frame_counter = i
            shader.Render(px)
        # (no counters() / reset here: they would make the renderer forget the loop it has seen; rays are taken from a frame count)
        n = int(os.environ.get("RTX_SYNC_FRAMES", max(16, min(args.steps, 64))))
        t0 = time.perf_counter()
        for i in range(n):
            shader.frame_counter = warm + i         # src/WindowManager.cpp:156-157
            shader.Render(px)
        dt = time.perf_counter() - t0
        c = shader.counters()
        info = shader.info()
        res[key] = {"ms_per_step": dt / n * 1e3, "mrays_per_s": float(c.rays) * n / (warm + n) / dt / 1e6, "frames": n, "look_ahead": info.lookAhead,
                    "hw_queues_seen": info.hwQueues}
        shader.close()

    def wrong_guess_ms():
        """The loop is left while sets of frames are ahead (the caller jumps to another frame counter): time of that one call."""
        sh = HIPShader(scene, device=0, traversal=trav, look_ahead=0)
        buf = np.zeros((sh.height, sh.width, 4), dtype=np.float32)
        out, base = [], 0
        for rep in range(3):
            for i in range(14):                       # the loop is recognised and the sets ahead have grown to their full size
                sh.frame_counter = base + i
                sh.Render(buf)
            t1 = time.perf_counter()
            sh.frame_counter = base + 1000
            sh.Render(buf)
            out.append((time.perf_counter() - t1) * 1e3)
            base += 2000
        sh.close()
        return min(out)
    jump = {"sets_cut_short": wrong_guess_ms()}
    os.environ["RTX_NO_ABORT"] = "1"
    jump["sets_run_out"] = wrong_guess_ms()
    del os.environ["RTX_NO_ABORT"]
    jump["note"] = ("ms of the one synchronous Render(pixels) call that leaves the loop (another frame counter) while sets of frames are ahead: "
                    "the library overwrites the work counters of those sets so that their launches run out at once (rtx_api.hip abort_ahead); "
                    "sets_run_out = the same call with RTX_NO_ABORT=1")
    out = dict(res["default"])
    out.update({"wrong_guess_ms": jump, "page_locked_destination": {"ms_per_step": res["page_locked"]["ms_per_step"], "mrays_per_s": res["page_locked"]["mrays_per_s"],
                                                                    "note": "the same loop with the caller's pixel array page-locked: one direct device-to-host copy per frame, no staging"},
                "without_look_ahead": {"ms_per_step": res["no_look_ahead"]["ms_per_step"], "mrays_per_s": res["no_look_ahead"]["mrays_per_s"]},
                "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                "pattern": "rtx_render(pixels) per frame, synchronous, 16 B/pixel copied to a pageable host array every frame; "
                           "look_ahead = frames the library starts before they are asked for (RtxConfig.lookAhead, same results)"})
    print(json.dumps(out))


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become N ranks. The launcher is started as a CHILD
    process, before this process has imported torch or touched the GPU in any way (never an exec of a process that has),
    with the command line the driver documents; rank 0's JSON line is relayed, the exit code is the launcher's."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.startswith("{"):
            try:
                out = json.loads(line)
                out.setdefault("config", {})["launched_by"] = "bench.py itself: child `python -m torch.distributed.run --nproc-per-node N` (WORLD_SIZE was unset)"
                line = json.dumps(out) + "\n"
            except ValueError:
                pass
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)      # stdout carries the JSON line only
        sys.stdout.flush()
    raise SystemExit(proc.wait())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", default="ranks", choices=["ranks", "group"],
                    help="how N > 1 GPUs are driven: ranks = one process per GPU, tiles gathered over torch.distributed (RCCL); "
                         "group = ONE process, a device group behind one handle (RtxConfig.numDevices), tiles gathered by peer copies")
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="cfg4_10Mtri_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--traversal", default="wide", choices=["wide", "culled", "exact"],
                    help="wide = the library default (the reference's result by construction)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sync_boundary / culled_opt_in legs (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="RtxConfig.framesInFlight for the timed region: 0 = library default for the process' hardware queues, "
                         "1 = strictly serial frames")
    ap.add_argument("--orbit", action="store_true", help="-O camera: new camera + frameCounter=0 every frame (src/Main.cpp:68-82)")
    ap.add_argument("--timed-only", action="store_true", help="profiling runs: skip the roofline pass too (the JSON line then has no roofline: not a bench line)")
    ap.add_argument("--sync-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.sync_only:
        return sync_only(args)
    if args.gpus > 1 and args.mode == "ranks" and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    # One GPU: HIP's defaults (4 hardware queues per process; three batches of frames in flight need no more — measured equal
    # with 16). Ranks of a multi-GPU run keep up to 14 shorter batches in flight, one stream each: they get a hardware
    # queue per stream (HIP reads this once, when the process first touches the GPU).
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or (args.mode == "group" and args.gpus > 1)) and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 16:
        os.environ["GPU_MAX_HW_QUEUES"] = "16"

    # stdout carries the JSON line and nothing else: libraries that write to file descriptor 1 on their own (RCCL prints a version
    # banner there when the process group initialises, gloo its connection chatter) are sent to stderr for the whole run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    group_n = args.gpus if (args.mode == "group" and args.gpus > 1) else 0
    if group_n and world > 1:
        raise SystemExit("--mode group is one process: do not start it under torch.distributed.run")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP backend has no CPU fallback)")
    # RTX_BENCH_BACKEND=gloo rehearses the multi-rank path where the ranks cannot each have a GPU (several ranks on
    # one device, tiles staged through host memory around the collective); the measured configuration is nccl.
    backend = os.environ.get("RTX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # RTX_BENCH_FORCE_DIST=1 takes the process-group path with a single rank too (a one-GPU check that RCCL
    # initialises next to the renderer's own streams and that the gather runs).
    use_dist = world > 1 or os.environ.get("RTX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    red_device = "cuda" if backend == "nccl" else "cpu"

    from raytracer_amd import _capi
    from raytracer_amd.dist import DeviceGather
    from raytracer_amd.renderer import HIPShader
    from raytracer_amd.scene import orbit as orbit_camera, synthetic_scene

    ntri, w, h, textured, grid = WORKLOADS[args.workload]
    cores = max(1, min(16, os.cpu_count() or 1))
    t0 = time.time()
    scene = synthetic_scene(ntri, w, h, seed=12345, textured=textured, nthreads=cores, grid=grid)
    t_scene = time.time() - t0
    modes = {"wide": _capi.TRAVERSE_WIDE, "culled": _capi.TRAVERSE_CULLED, "exact": _capi.TRAVERSE_EXACT}
    trav = modes[args.traversal]
    t0 = time.time()
    group_devices = None
    if group_n:
        present = max(1, torch.cuda.device_count())
        group_devices = [i % present for i in range(group_n)]          # fewer GPUs than members: rehearsal, members share devices
        shader = HIPShader(scene, devices=group_devices, traversal=trav, frames_in_flight=args.frames_in_flight)
    else:
        shader = HIPShader(scene, device=local_rank, rank=rank, world_size=world, traversal=trav,
                           frames_in_flight=args.frames_in_flight)
    t_create = time.time() - t0
    info = shader.info()
    W, H = shader.width, shader.height
    gather = DeviceGather(shader, rank, world, torch.device("cuda", local_rank), host_staging=backend != "nccl") if use_dist else None
    cam = scene.camera
    frame = np.zeros((H, W, 4), dtype=np.float32) if rank == 0 else None

    def step(sh, i):
        if args.orbit:
            orbit_camera(cam, 0.1 * i, 0.1 * i)
            sh.frame_counter = 0
        else:
            sh.frame_counter = i
        sh.Render(None, cam)

    def deliver():
        """The batch's one exchange and one copy: tiles to rank 0 (RCCL), finished frame to the host."""
        if gather is not None:
            gather.gather()
        if rank == 0:
            shader.download_into(frame)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        shader.counters()            # synchronises the renderer's own streams

    for i in range(args.warmup):
        step(shader, i)
    deliver()
    barrier()
    shader.reset_counters()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(shader, args.warmup + i)
    deliver()
    c = shader.counters()              # stream sync
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # the one device-to-host copy that closes the timed region, once more on its own (nothing is queued, the frame is resident):
    # it follows the last resolve and overlaps nothing, so elapsed - d2h is the time to the finished frame in HBM
    d2h_s = 0.0
    if rank == 0:
        t1 = time.perf_counter()
        shader.download_into(frame)
        d2h_s = time.perf_counter() - t1

    rays_local = float(c.rays)
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
    rsum = torch.tensor([rays_local], dtype=torch.float64, device=red_device)
    per_rank = [{"rank": 0, "rays": int(rays_local), "ms": elapsed * 1e3, "device_ms": c.totalMs}]
    if use_dist:
        mine = torch.tensor([rays_local, elapsed * 1e3, c.totalMs], dtype=torch.float64, device=red_device)
        everyone = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(everyone, mine)
        per_rank = [{"rank": k, "rays": int(v[0].item()), "ms": float(v[1].item()), "device_ms": float(v[2].item())} for k, v in enumerate(everyone)]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays = float(rsum.item())
    kernel_ms = [x / max(1, c.frames) for x in c.kernelMs]
    names8 = ["primary", "shade0", "trace1", "shade1", "trace2", "shade2", "trace3", "shade3"]

    extra = {"scene_build_s": round(t_scene, 2), "rtx_create_s": round(t_create, 2),
             "rtx_create_phases_s": {"validate_walk": info.createSeconds[0], "own_tree": info.createSeconds[1],
                                     "records": info.createSeconds[2], "upload_alloc": info.createSeconds[3]},
             "device_bytes": int(shader.info().deviceBytes), "device_bytes_at_create": int(info.deviceBytes), "wide_nodes": int(info.wideNodes), "wide_tree_depth": info.wideTreeDepth,
             "overlapped_kernel_ms_per_frame": dict(zip(names8, kernel_ms)),
             "device_ms_per_frame": c.totalMs / max(1, c.frames),
             "rays_per_bounce_rank0": [int(x) for x in c.raysPerBounce], "hits_per_bounce_rank0": [int(x) for x in c.hitsPerBounce]}

    # ---- roofline of the dominant kernel (rank 0) ---------------------------------------------------------
    # The timed region keeps several frames in flight, so a launch there shares the machine with launches of
    # other frames and its duration is not its own. The roofline pass renders the same frames again one at a
    # time (framesInFlight = 1, framesPerBatch = 1) with the counting instantiation of the kernels: node steps, leaf tests and the
    # HIP-event duration then come from the very same launches (the counters cost ~3 %).
    roofline = None
    if rank == 0 and args.timed_only:
        shader.close()
    elif rank == 0:
        shader.close()
        counter = HIPShader(scene, device=local_rank, rank=rank, world_size=world, traversal=trav, count_stats=True,
                            frames_in_flight=1, frames_per_batch=1 if world == 1 else 0)
        step(counter, 0)
        counter.reset_counters()
        for i in range(args.steps):
            step(counter, args.warmup + i)
        cc = counter.counters()
        counter.close()
        serial_ms = [x / max(1, cc.frames) for x in cc.kernelMs]
        trace_ms = [serial_ms[0], serial_ms[2], serial_ms[4], serial_ms[6]]
        b = int(np.argmax(trace_ms))
        inner = cc.innerVisits[b] / args.steps
        leaf = cc.leafTests[b] / args.steps
        # bytes the loads of a step request: quantised 4-wide node 48 B (3 x 16) and 64-B leaf record (wide);
        # 4-wide float node 112 B (7 x 16) and 40 B of the leaf record (culled); binary node 64 B + 40 B (exact)
        node_bytes, leaf_bytes, node_loads, leaf_loads = {"wide": (48.0, 64.0, 3.0, 4.0), "culled": (112.0, 40.0, 7.0, 3.0),
                                                          "exact": (64.0, 40.0, 4.0, 3.0)}[args.traversal]
        # headline `achieved` keeps SURVEY.md §8d's constants (64 B per node step, 48 B per leaf test) whatever the
        # node format, so that rounds stay comparable; the bytes the loads really request are reported beside it
        algo_bytes = 64.0 * inner + 48.0 * leaf
        achieved = algo_bytes / (trace_ms[b] * 1e-3) / 1e9 if trace_ms[b] > 0 else 0.0
        kname = ["k_primary", "k_trace1", "k_trace2", "k_trace3"][b]
        traffic, tcp_accesses, traffic_source, pmc_counters = None, None, None, None
        digest = csrc_digest()
        pmc = os.path.join(HERE, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                key = f"{args.workload}:{args.traversal}"
                if pj.get("csrc_sha256_16") == digest and key in pj and not args.orbit:
                    traffic = pj[key].get(kname)
                    tcp_accesses = pj.get(key + ":tcp_accesses", {}).get(kname)
                    pmc_counters = (pj.get(key + ":counters") or {}).get(kname)
                    traffic_source = f"profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel source, csrc {digest})"
                else:
                    traffic_source = (f"not emitted: profiles/pmc_traffic.json was taken with csrc {pj.get('csrc_sha256_16')} "
                                      f"for {sorted(k for k in pj if ':' in k and not k.endswith('tcp_accesses'))}, this run is csrc {digest} {key}")
            except Exception as e:          # a malformed file must not cost the bench line
                traffic_source = f"not emitted: {e}"
        all_rays = sum(cc.raysPerBounce)
        props = torch.cuda.get_device_properties(local_rank)
        clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
        lane_loads = node_loads * inner + leaf_loads * leaf
        cu_clocks = trace_ms[b] * 1e-3 * clock_hz * props.multi_processor_count
        kernels = {"wide": ["k_primary<WIDEQ, count>", "k_trace_qw<count, 1>", "k_trace_qw<count, 2>", "k_trace_qw<count, 3>"],
                   "culled": ["k_primary<CULLED, count>", "k_trace_refill<count, 1>", "k_trace_refill<count, 2>", "k_trace_refill<count, 3>"],
                   "exact": ["k_primary<EXACT, count>", "k_trace<EXACT, count> bounce 1", "k_trace<EXACT, count> bounce 2",
                             "k_trace<EXACT, count> bounce 3"]}[args.traversal]
        # What limits the launch is DERIVED from the committed counters of this very kernel source (same digest gate as
        # `traffic`), never written down by hand: HBM if the counted traffic is most of the peak, else the busiest of the CU's units.
        launch_s = trace_ms[b] * 1e-3
        frac_counted = (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if (traffic and launch_s > 0) else None
        L1_CEILING = 0.87          # lane loads per clock per CU the vector L1 accepts on this pattern (tools/tcp_rate, profiles/r01_tcp_rate.txt)
        limited_by, limiter = None, None
        if pmc_counters:
            l1_rate = (pmc_counters["tcp_accesses"] / cu_clocks) if (pmc_counters.get("tcp_accesses") and cu_clocks > 0) else None
            units = {"hbm": frac_counted or 0.0, "valu_issue": pmc_counters.get("valu_busy_frac") or 0.0,
                     "vector_l1_front_end": (l1_rate / L1_CEILING) if l1_rate else 0.0}
            limited_by = max(units, key=units.get)
            at_ceiling = sorted(k for k, v in units.items() if v >= 0.85)
            limiter = {"utilisation_of_each_ceiling": units, "at_or_near_ceiling": at_ceiling,
                       "valu_busy_frac": pmc_counters.get("valu_busy_frac"), "valu_lane_util_frac": pmc_counters.get("valu_lane_util_frac"),
                       "vector_l1_accesses_per_clock_per_cu": l1_rate, "vector_l1_ceiling_per_clock_per_cu": L1_CEILING,
                       "l2_hit_frac": pmc_counters.get("l2_hit_frac"), "wave_time_waiting_frac": pmc_counters.get("wait_any_frac"),
                       "l2_read_latency_cycles": pmc_counters.get("l2_read_latency_cycles"),
                       "rocprof_avg_us_of_this_kernel": pmc_counters.get("avg_us"), "source": traffic_source}
        roofline = {"bound": "hbm", "bound_meaning": "the roofline this gather-type fp32 kernel is priced against (no dense contraction: no MFMA); "
                                                     "what actually limits the launch is `limited_by`, derived from the counters",
                    "limited_by": limited_by, "limiter": limiter, "kernel": kernels[b],
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_source,
                    "hbm_traffic_gbs": (traffic / (trace_ms[b] * 1e-3) / 1e9) if (traffic and trace_ms[b] > 0) else None,
                    "frac_meaning": "`achieved` / `frac` price device-counted visits at SURVEY 8d's 64 B / 48 B (algorithmic bytes, most of which the L2 / "
                                    "Infinity Cache serve); frac_requested = bytes the loads ask for; frac_counted = PMC-counted HBM-side traffic: the HBM utilisation proper",
                    "frac_requested": ((node_bytes * inner + leaf_bytes * leaf) / launch_s / 1e9 / HBM_PEAK_GBS) if launch_s > 0 else None,
                    "frac_counted": frac_counted,
                    "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": trace_ms[b],
                    "launch_ms_overlapped_in_timed_region": [kernel_ms[0], kernel_ms[2], kernel_ms[4], kernel_ms[6]][b],
                    "node_steps_per_ray": [cc.innerVisits[i] / max(1, cc.raysPerBounce[i]) for i in range(4)],
                    "leaf_tests_per_ray": [cc.leafTests[i] / max(1, cc.raysPerBounce[i]) for i in range(4)],
                    "bytes_per_node_step_survey": 64.0, "bytes_per_leaf_test_survey": 48.0,
                    "requested_bytes_per_launch": node_bytes * inner + leaf_bytes * leaf,
                    "requested_bytes_per_node_step": node_bytes, "requested_bytes_per_leaf_test": leaf_bytes,
                    "requested_gbs": (node_bytes * inner + leaf_bytes * leaf) / (trace_ms[b] * 1e-3) / 1e9 if trace_ms[b] > 0 else 0.0,
                    "work_items_per_lane_pass": [(cc.innerVisits[i] + cc.leafTests[i]) / (64.0 * cc.wavePasses[i]) if cc.wavePasses[i] else None
                                                 for i in range(4)],
                    "algorithmic_bytes_per_ray_all_trace": (64.0 * sum(cc.innerVisits) + 48.0 * sum(cc.leafTests)) / max(1, all_rays),
                    "vector_l1_front_end": {"lane_loads_issued_per_launch": lane_loads, "l1_accesses_per_launch_pmc": tcp_accesses,
                                            "issued_per_clock_per_cu": lane_loads / cu_clocks if cu_clocks > 0 else None,
                                            "per_clock_per_cu_pmc": (tcp_accesses / cu_clocks) if (tcp_accesses and cu_clocks > 0) else None,
                                            "ceiling_per_clock_per_cu": 0.87, "ceiling_source": "tools/tcp_rate (profiles/r01_tcp_rate.txt)",
                                            "clock_hz": clock_hz, "cus": props.multi_processor_count}}
        # the reference's own visit counts (its tree, no distance culling, binary nodes) for
        # the same rays, and the bandwidth they would amount to in this kernel's time ("work-equivalent")
        if args.traversal != "exact" and not args.no_extras:
            refc = HIPShader(scene, device=local_rank, rank=rank, world_size=world, traversal=_capi.TRAVERSE_EXACT, count_stats=True,
                             frames_in_flight=1, frames_per_batch=1 if world == 1 else 0)
            step(refc, args.warmup)
            rc = refc.counters()
            refc.close()
            ref_inner = [rc.innerVisits[i] / max(1, rc.raysPerBounce[i]) for i in range(4)]
            ref_leaf = [rc.leafTests[i] / max(1, rc.raysPerBounce[i]) for i in range(4)]
            rays_b = cc.raysPerBounce[b] / args.steps
            roofline["reference_semantics"] = {
                "node_visits_per_ray": ref_inner, "leaf_tests_per_ray": ref_leaf,
                "work_equivalent_gbs": (64.0 * ref_inner[b] + 48.0 * ref_leaf[b]) * rays_b / (trace_ms[b] * 1e-3) / 1e9 if trace_ms[b] > 0 else 0.0,
                "note": "visit counts of RTX_TRAVERSE_EXACT (the reference's tree, 64-byte binary nodes) on the same rays; "
                        "work_equivalent = those bytes over this launch's time. leaf_tests_per_ray equals the wide walk's: same leaf set"}
        extra["serial_kernel_ms_per_frame"] = dict(zip(names8, serial_ms))
        extra["serial_device_ms_per_frame"] = cc.totalMs / max(1, cc.frames)

    # ---- the same frames with distance culling (opt-in mode; not exact by construction) ---------------------------
    culled = None
    if rank == 0 and world == 1 and not group_n and args.traversal == "wide" and not args.no_extras:
        other = HIPShader(scene, device=local_rank, traversal=_capi.TRAVERSE_CULLED, frames_in_flight=args.frames_in_flight)
        for i in range(args.warmup):
            step(other, i)
        other.counters(); other.reset_counters()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(other, args.warmup + i)
        other.download_into(frame)
        cw = other.counters()
        dt = time.perf_counter() - t1
        culled = {"traversal": "culled", "mrays_per_s": float(cw.rays) / dt / 1e6, "ms_per_step": dt / args.steps * 1e3,
                  "cull_slack": other.info().cullSlack,
                  "note": "RTX_TRAVERSE_CULLED, opt-in: nearest-first walk of 128-byte float nodes with distance culling; same image here, not a proof"}
        other.close()

    # ---- the reference's calling pattern, default hardware queues, in a process of its own ----------------------
    sync = None
    if rank == 0 and world == 1 and not group_n and not args.no_extras and not args.orbit:
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--sync-only", "--workload", args.workload,
                                  "--traversal", args.traversal, "--steps", str(args.steps)], env=env, capture_output=True, text=True, timeout=240)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            sync = json.loads(line[-1]) if line else {"error": (out.stderr or out.stdout)[-400:]}
        except Exception as e:
            sync = {"error": str(e)}

    # ---- CPU baseline: the oracle (CPU restatement of ThreadedShader) on a bounded sample ----
    cpu = None
    if rank == 0 and world == 1 and not group_n and not args.no_cpu_baseline:
        import oracle
        desc = scene.desc()
        fc = args.warmup
        px = np.zeros((H, W, 4), dtype=np.float32)
        mid = H // 2

        def time_oracle(L, seconds):
            t1 = time.perf_counter()
            oracle.render_bands(desc, cam, fc, px, nthreads=cores, rows=(mid - 8, mid + 8), L=L)
            probe = time.perf_counter() - t1
            rows = int(max(cores, min(H, 16 * seconds / max(probe, 1e-3))))
            y0 = max(0, mid - rows // 2)
            t1 = time.perf_counter()
            st = oracle.render_bands(desc, cam, fc, px, nthreads=cores, rows=(y0, y0 + rows), L=L)
            dt = time.perf_counter() - t1
            return st.rays / dt / 1e6, f"rows {y0}..{y0 + rows} of {H} ({rows * W} pixels) of frame {fc} of the same scene, {st.rays} rays, {dt:.1f} s"
        v, sample = time_oracle(None, args.cpu_seconds)
        try:
            import psutil
            physical = psutil.cpu_count(logical=False)
        except Exception:
            physical = None
        cpu = {"value": v, "unit": "Mrays/s", "cores": cores, "kind": "port",
               "cores_note": f"threads used = the CPU share of a one-GPU box (at most 16); the machine has {physical} physical cores ({os.cpu_count()} logical)",
               "sample": sample + "; oracle/liboracle.so = strict -O2 restatement of ThreadedShader, static row bands as src/ThreadedShader.cpp:186-207"}
        Lf = oracle.lib_ofast()
        if Lf is not None:
            vf, samplef = time_oracle(Lf, args.cpu_seconds / 2)
            cpu["ofast"] = {"value": vf, "unit": "Mrays/s", "cores": cores,
                            "sample": samplef + "; the same source with the reference's shipped flags (-Ofast, CMakeLists.txt:11)"}

    if rank == 0:
        if group_n:
            collective = {"kind": "one process, device group: hipMemcpyPeerAsync fan-in to devices[0] (no process group)", "members": group_n,
                          "devices": group_devices, "gpus_present": torch.cuda.device_count(), "rays_all_members": int(rays_local)}
        elif use_dist:
            collective = {"kind": "torch.distributed.gather to rank 0", "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                          "gpus_present": torch.cuda.device_count(), "per_rank": per_rank, "rays_sum_over_ranks": int(rays)}
        else:
            collective = None
        out = {"metric": "Mrays/s", "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": group_n if group_n else world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": args.workload, "triangles": ntri, "resolution": [W, H], "bounces": 4,
                          "spp_per_step": 1, "traversal": args.traversal, "traversal_in_effect": {v: k for k, v in modes.items()}[info.traversal],
                          "exact_by_construction": args.traversal != "culled",
                          "partition": f"tiles64x32 round-robin x{group_n if group_n else world}", "mode": args.mode if (group_n or world > 1) else "single",
                          "camera": "orbit" if args.orbit else "fixed", "collective": collective,
                          "timed_region": "K x rtx_render(pixels=NULL) + tile gather to rank 0 + one device-to-host copy of the frame",
                          "frames_in_flight": info.framesInFlight, "frames_per_batch": info.framesPerBatch,
                          "GPU_MAX_HW_QUEUES": info.hwQueues},
               "value_device_resident": {"value": rays / max(elapsed - d2h_s, 1e-9) / 1e6, "unit": "Mrays/s", "final_d2h_copy_ms": d2h_s * 1e3,
                                         "note": "the same timed region without its closing PCIe copy (K frames + tile gather, finished frame resident in HBM on rank 0); "
                                                 "`value` keeps the copy in, as in earlier rounds"},
               "sync_boundary": sync, "culled_opt_in": culled,
               "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
