#!/usr/bin/env python3
"""bench.py — frame-render hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one frame = one sample per pixel over the whole image (the reference's
"N spp" is N accumulated frames). Default workload = BASELINE.json
configs[3] shape: synthetic 10 M-triangle mesh + textured materials, 3840x2160
(rendered at 3840x2176 after the reference's rounding to multiples of 32), scene
replicated on every GPU, screen tiles dealt round-robin to the ranks, tiles
gathered to rank 0 once at the end of the timed batch (RCCL). Total work is the
same for every N (strong scaling). The frame stays resident in HBM; the optional
device-to-host copy is reported separately (--d2h), never inside `value`.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
# frames are queued several deep, one stream each: give the process a hardware queue per stream
# (HIP reads this once, when the process first touches the GPU; the usual value is 4)
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

WORKLOADS = {
    # name: (triangles, width, height, textured)
    "cfg4_10Mtri_4k": (10_000_000, 3840, 2160, True),
    "cfg3_1Mtri_1080p": (1_000_000, 1920, 1080, False),
    "cfg5_100Mtri_4k": (100_000_000, 3840, 2160, True),
    "small_100ktri_992": (100_000, 1000, 1000, True),
}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="cfg4_10Mtri_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--traversal", default="culled", choices=["culled", "exact", "wide"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="RtxConfig.framesInFlight for the timed region: 0 = library default (14 with the 16 hardware queues set above), "
                         "1 = strictly serial frames")
    ap.add_argument("--d2h", action="store_true", help="also time K frames with the host copy (reported in `extra`)")
    ap.add_argument("--orbit", action="store_true", help="-O camera: new camera + frameCounter=0 every frame (src/Main.cpp:68-82)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

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

    ntri, w, h, textured = WORKLOADS[args.workload]
    cores = max(1, min(16, os.cpu_count() or 1))
    t0 = time.time()
    scene = synthetic_scene(ntri, w, h, seed=12345, textured=textured, nthreads=cores)
    t_scene = time.time() - t0
    trav = {"culled": _capi.TRAVERSE_CULLED, "exact": _capi.TRAVERSE_EXACT, "wide": _capi.TRAVERSE_WIDE}[args.traversal]
    t0 = time.time()
    shader = HIPShader(scene, device=local_rank, rank=rank, world_size=world, traversal=trav,
                       frames_in_flight=args.frames_in_flight)
    t_create = time.time() - t0
    W, H = shader.width, shader.height
    gather = DeviceGather(shader, rank, world, torch.device("cuda", local_rank), host_staging=backend != "nccl") if use_dist else None
    cam = scene.camera

    def step(i):
        if args.orbit:
            orbit_camera(cam, 0.1 * i, 0.1 * i)
            shader.frame_counter = 0
            shader.Render(None, cam)
        else:
            shader.frame_counter = i
            shader.Render(None, cam)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        shader.counters()            # synchronises the renderer's own stream

    for i in range(args.warmup):
        step(i)
    if gather is not None:
        gather.gather()
    barrier()
    shader.reset_counters()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if gather is not None:
        gather.gather()                # one exchange per delivered frame batch
    c = shader.counters()              # stream sync
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    rays_local = float(c.rays)
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
    rsum = torch.tensor([rays_local], dtype=torch.float64, device=red_device)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rsum, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays = float(rsum.item())
    kernel_ms = [x / max(1, c.frames) for x in c.kernelMs]
    rays_per_bounce = [int(x) for x in c.raysPerBounce]
    hits_per_bounce = [int(x) for x in c.hitsPerBounce]

    extra = {"scene_build_s": round(t_scene, 2), "rtx_create_s": round(t_create, 2),
             "overlapped_kernel_ms_per_frame": {"primary": kernel_ms[0], "shade0": kernel_ms[1], "trace1": kernel_ms[2],
                                     "shade1": kernel_ms[3], "trace2": kernel_ms[4], "shade2": kernel_ms[5],
                                     "trace3": kernel_ms[6], "shade3": kernel_ms[7]},
             "device_ms_per_frame": c.totalMs / max(1, c.frames),
             "rays_per_bounce_rank0": rays_per_bounce, "hits_per_bounce_rank0": hits_per_bounce}

    if args.d2h and rank == 0:
        px = np.zeros((H, W, 4), dtype=np.float32)
        shader.Render(px, cam)
        t1 = time.perf_counter()
        for i in range(args.steps):
            shader.frame_counter = args.warmup + args.steps + i
            shader.Render(px, cam)
        extra["ms_per_frame_with_d2h"] = (time.perf_counter() - t1) / args.steps * 1e3

    # ---- roofline of the dominant kernel (rank 0) ---------------------------------------------------------
    # The timed region keeps several frames in flight, so a launch there shares the machine with launches of
    # other frames and its duration is not its own. The roofline pass renders the same frames again one at a
    # time (framesInFlight = 1) with the counting instantiation of the kernels: node steps, leaf tests and the
    # HIP-event duration then come from the very same launches (the counters cost ~3 %).
    roofline = None
    if rank == 0:
        shader.close()
        counter = HIPShader(scene, device=local_rank, rank=rank, world_size=world, traversal=trav, count_stats=True,
                            frames_in_flight=1)
        def count_step(i):
            if args.orbit:
                orbit_camera(cam, 0.1 * i, 0.1 * i)
                counter.frame_counter = 0
            else:
                counter.frame_counter = i
            counter.Render(None, cam)
        count_step(0)
        counter.reset_counters()
        for i in range(args.steps):
            count_step(args.warmup + i)
        cc = counter.counters()
        counter.close()
        serial_ms = [x / max(1, cc.frames) for x in cc.kernelMs]
        trace_ms = [serial_ms[0], serial_ms[2], serial_ms[4], serial_ms[6]]
        b = int(np.argmax(trace_ms))
        inner = cc.innerVisits[b] / args.steps
        leaf = cc.leafTests[b] / args.steps
        # bytes a traversal step reads: culled path = one 4-wide node, 7 x 16 B (6 plane vectors + 4 refs);
        # exact path = one binary node of the reference topology, 64 B; a leaf test reads 48 B
        node_bytes = 112.0 if args.traversal in ("culled", "wide") else 64.0
        requested_bytes = node_bytes * inner + 48.0 * leaf
        # headline `achieved` keeps SURVEY.md §8d's constants (64 B per node step, 48 B per leaf test) whatever the
        # node width, so that rounds stay comparable; the bytes the loads really request are reported beside it
        algo_bytes = 64.0 * inner + 48.0 * leaf
        achieved = algo_bytes / (trace_ms[b] * 1e-3) / 1e9 if trace_ms[b] > 0 else 0.0
        traffic, tcp_accesses = None, None
        pmc = os.path.join(HERE, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                kname = ["k_primary", "k_trace1", "k_trace2", "k_trace3"][b]
                traffic = pj.get(args.workload, {}).get(kname)
                tcp_accesses = pj.get(args.workload + ":tcp_accesses", {}).get(kname)
            except Exception:
                traffic, tcp_accesses = None, None
        all_rays = sum(cc.raysPerBounce)
        # What actually stops this launch is not HBM but the rate at which a CU's vector L1 accepts 16-byte lane
        # loads (tools/tcp_rate, profiles/r01_tcp_rate.txt: 0.87 per clock per CU for this access pattern):
        # a node step of the culled path is 7 such loads per lane, a leaf test 3.
        props = torch.cuda.get_device_properties(local_rank)
        clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
        lane_loads = (7.0 if args.traversal in ("culled", "wide") else 4.0) * inner + 3.0 * leaf
        cu_clocks = trace_ms[b] * 1e-3 * clock_hz * props.multi_processor_count
        l1 = {"lane_loads_issued_per_launch": lane_loads,                  # 7 I' + 3 L' from the device counters
              # what the L1 counted for this launch (TCP_TOTAL_CACHE_ACCESSES, profiles/pmc_traffic.json): lanes of a
              # wave that read the same address — the top of the tree — arrive merged, so this is below `issued`
              "l1_accesses_per_launch_pmc": tcp_accesses,
              "per_clock_per_cu": (tcp_accesses / cu_clocks) if (tcp_accesses and cu_clocks > 0) else None,
              "issued_per_clock_per_cu": lane_loads / cu_clocks if cu_clocks > 0 else None,
              "ceiling_per_clock_per_cu": 0.87, "ceiling_source": "tools/tcp_rate mode 2 (7 loads per scattered line, L2-resident)",
              "clock_hz": clock_hz, "cus": props.multi_processor_count}
        names = (["k_primary<CULLED, count>", "k_trace_refill<count, 1>", "k_trace_refill<count, 2>", "k_trace_refill<count, 3>"]
                 if args.traversal in ("culled", "wide") else ["k_primary<EXACT, count>", "k_trace<EXACT, count> bounce 1",
                                                     "k_trace<EXACT, count> bounce 2", "k_trace<EXACT, count> bounce 3"])
        roofline = {"bound": "hbm", "kernel": names[b],
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic,
                    "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": trace_ms[b],
                    "launch_ms_overlapped_in_timed_region": [kernel_ms[0], kernel_ms[2], kernel_ms[4], kernel_ms[6]][b],
                    "node_steps_per_ray": [cc.innerVisits[i] / max(1, cc.raysPerBounce[i]) for i in range(4)],
                    "leaf_tests_per_ray": [cc.leafTests[i] / max(1, cc.raysPerBounce[i]) for i in range(4)],
                    "bytes_per_node_step_survey": 64.0, "bytes_per_leaf_test": 48.0,
                    "requested_bytes_per_launch": requested_bytes, "requested_bytes_per_node_step": node_bytes,
                    "requested_gbs": requested_bytes / (trace_ms[b] * 1e-3) / 1e9 if trace_ms[b] > 0 else 0.0,
                    "work_items_per_lane_pass": [(cc.innerVisits[i] + cc.leafTests[i]) / (64.0 * cc.wavePasses[i]) if cc.wavePasses[i] else None
                                                 for i in range(4)],
                    "algorithmic_bytes_per_ray_all_trace": (64.0 * sum(cc.innerVisits) + 48.0 * sum(cc.leafTests)) / max(1, all_rays),
                    "vector_l1_front_end": l1}
        # the reference's own visit counts (its tree, no distance culling, binary nodes) for
        # the same rays, and the bandwidth they would amount to in this kernel's time ("work-equivalent")
        if args.traversal != "exact":
            refc = HIPShader(scene, device=local_rank, rank=rank, world_size=world, traversal=_capi.TRAVERSE_EXACT, count_stats=True,
                             frames_in_flight=1)
            refc.frame_counter = args.warmup
            refc.Render(None, cam)
            rc = refc.counters()
            refc.close()
            ref_inner = [rc.innerVisits[i] / max(1, rc.raysPerBounce[i]) for i in range(4)]
            ref_leaf = [rc.leafTests[i] / max(1, rc.raysPerBounce[i]) for i in range(4)]
            rays_b = cc.raysPerBounce[b] / args.steps
            roofline["reference_semantics"] = {
                "node_visits_per_ray": ref_inner, "leaf_tests_per_ray": ref_leaf,
                "work_equivalent_gbs": (64.0 * ref_inner[b] + 48.0 * ref_leaf[b]) * rays_b / (trace_ms[b] * 1e-3) / 1e9 if trace_ms[b] > 0 else 0.0,
                "note": "visit counts of RTX_TRAVERSE_EXACT (the reference's tree, 64-byte binary nodes) on the same rays; work_equivalent = those bytes over this launch's time"}
        extra["serial_kernel_ms_per_frame"] = {"primary": serial_ms[0], "shade0": serial_ms[1], "trace1": serial_ms[2],
                                               "shade1": serial_ms[3], "trace2": serial_ms[4], "shade2": serial_ms[5],
                                               "trace3": serial_ms[6], "shade3": serial_ms[7]}
        extra["serial_device_ms_per_frame"] = cc.totalMs / max(1, cc.frames)

    # ---- the same frames without distance culling (RTX_TRAVERSE_WIDE: the reference's leaf set by construction) ----
    if rank == 0 and world == 1 and args.traversal == "culled":
        wide = HIPShader(scene, device=local_rank, traversal=_capi.TRAVERSE_WIDE, frames_in_flight=args.frames_in_flight)
        for i in range(args.warmup):
            wide.frame_counter = i
            wide.Render(None, cam)
        wide.counters(); wide.reset_counters()
        t1 = time.perf_counter()
        for i in range(args.steps):
            if args.orbit:
                orbit_camera(cam, 0.1 * i, 0.1 * i)
                wide.frame_counter = 0
            else:
                wide.frame_counter = args.warmup + i
            wide.Render(None, cam)
        cw = wide.counters()
        dt = time.perf_counter() - t1
        extra["without_distance_culling"] = {"traversal": "wide", "mrays_per_s": float(cw.rays) / dt / 1e6, "ms_per_step": dt / args.steps * 1e3}
        wide.close()

    # ---- CPU baseline: the oracle (CPU restatement of ThreadedShader) on a bounded sample ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        desc = scene.desc()
        fc = args.warmup
        px = np.zeros((H, W, 4), dtype=np.float32)
        mid = H // 2
        t1 = time.perf_counter()
        st = oracle.render(desc, cam, fc, px, nthreads=cores, rows=(mid - 4, mid + 4), stats=1)
        probe = time.perf_counter() - t1
        rows = int(max(8, min(H, 8 * args.cpu_seconds / max(probe, 1e-3))))
        y0 = max(0, mid - rows // 2)
        dt, cpu_rays, cpu_frames = 0.0, 0, 0
        while cpu_frames == 0 or (rows == H and dt < args.cpu_seconds and cpu_frames < 8):
            t1 = time.perf_counter()
            st = oracle.render(desc, cam, fc + cpu_frames, px, nthreads=cores, rows=(y0, y0 + rows), stats=1)
            dt += time.perf_counter() - t1
            cpu_rays += st.rays
            cpu_frames += 1
        cpu = {"value": cpu_rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
               "sample": f"rows {y0}..{y0 + rows} of {H} ({rows * W} pixels) of frames {fc}..{fc + cpu_frames - 1} of the same scene, "
                         f"{cpu_rays} rays, oracle/liboracle.so (strict -O2 restatement of ThreadedShader), {dt:.1f} s"}

    if rank == 0:
        out = {"metric": "Mrays/s", "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": args.workload, "triangles": ntri, "resolution": [W, H], "bounces": 4,
                          "spp_per_step": 1, "traversal": args.traversal, "partition": f"tiles64x32 round-robin x{world}",
                          "camera": "orbit" if args.orbit else "fixed", "collective": backend if use_dist else None, "frames_in_flight": args.frames_in_flight or 14, "frames_per_batch": int(os.environ.get("RTX_BATCH", min(world, 8))), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
               "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
