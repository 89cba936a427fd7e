#!/usr/bin/env python3
"""Summarise the rocprofv3 output of scripts/prof_round.sh.

Per kernel — and for the frame pipeline's trace/shade launches per bounce (`#b1` = the launch
after the first shade of a frame, found by walking the dispatches in order) — the launch count,
average duration and the per-launch average of every collected counter, plus derived ratios.
GRBM_GUI_ACTIVE arrives summed over the 8 XCDs and is divided by 8 (it is a clock).

usage: python scripts/pmc_summary.py gpurun_out/prof_<tag> [summary.json [traffic.json workload:traversal]]
(traffic.json is stamped with the digest of raytracer_amd/csrc/*.h,*.hip it was measured with; bench.py emits
`roofline.traffic` only while that digest matches the source it runs)
"""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


class BounceNamer:
    """k_primary starts a frame; each following k_trace*/k_shade launch gets its bounce number."""
    def __init__(self):
        self.trace = 0
        self.shade = 0

    def name(self, kernel):
        k = short(kernel)
        if k.startswith("k_primary"):
            self.trace, self.shade = 0, 0
            return k
        if k.startswith("k_trace_refill") or k.startswith("k_trace_q"):
            return f"{k}#b{k[k.index('<') + 1:k.index('>')].split(',')[1].strip()}"     # the bounce is a template argument
        if k.startswith("k_trace<"):
            self.trace += 1
            return f"{k}#b{self.trace}"
        if k.startswith("k_shade"):
            n = f"{k}#b{self.shade}"
            self.shade += 1
            return n
        return k


def main():
    root = sys.argv[1]
    out = {}
    for f in glob.glob(os.path.join(root, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        # enqueue order (frames are queued one after the other even when they overlap on the device)
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id") or r["Start_Timestamp"]))
        namer = BounceNamer()
        dur = defaultdict(list)
        for r in rows:
            us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            dur[namer.name(r["Kernel_Name"])].append(us)
        for k, v in dur.items():
            out[k] = {"launches": len(v), "avg_us": sum(v) / len(v), "total_ms": sum(v) / 1e3}
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        kernel_of = {}
        for r in csv.DictReader(open(f)):
            d = int(r["Dispatch_Id"])
            per_dispatch[(d, r["Counter_Name"])] += float(r["Counter_Value"])
            kernel_of[d] = r["Kernel_Name"]
        namer = BounceNamer()
        name_of = {d: namer.name(kernel_of[d]) for d in sorted(kernel_of)}
        acc = defaultdict(lambda: defaultdict(list))
        for (d, c), v in per_dispatch.items():
            if c.startswith("GRBM_"):
                v /= 8.0
            acc[name_of[d]][c].append(v)
        for k, cs in acc.items():
            o = out.setdefault(k, {})
            for c, v in cs.items():
                o[c] = sum(v) / len(v)
    for k, o in out.items():
        g = o.get
        if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
            # one count = one VALU instruction occupying its SIMD for 4 cycles; 1024 SIMDs
            o["valu_busy_frac"] = g("SQ_ACTIVE_INST_VALU") * 4 / (1024 * g("GRBM_GUI_ACTIVE"))
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            o["valu_lane_util_frac"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
            o["l2_hit_frac"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        if g("FETCH_SIZE") is not None:
            o["fetch_bytes_raw"] = g("FETCH_SIZE") * 1024.0       # FETCH_SIZE is in KiB
        if g("WRITE_SIZE") is not None:
            o["write_bytes_raw"] = g("WRITE_SIZE") * 1024.0
        if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
            o["wait_any_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
            o["l2_read_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(json.dumps(out, indent=1, sort_keys=True) + "\n")
    if len(sys.argv) > 4:
        workload = sys.argv[4]
        def hbm(k):
            o = out.get(k, {})
            if "fetch_bytes_raw" not in o:
                return None
            # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2; WRITE_SIZE exact
            return 2.0 * o["fetch_bytes_raw"] + o.get("write_bytes_raw", 0.0)
        def targs(k):
            return [a.strip() for a in k[k.index("<") + 1:k.index(">")].split(",")] if "<" in k else []
        # the counting instantiations (COUNT = true): bench.py's roofline pass runs them one frame at a time
        counting = lambda k: (k.startswith("k_primary") and targs(k)[1:2] == ["true"]) or \
                             (k.startswith("k_trace") and targs(k)[:1] == ["true"]) or \
                             (k.startswith("k_trace<") and targs(k)[1:2] == ["true"])
        names = {"k_primary": [k for k in out if k.startswith("k_primary") and counting(k)],
                 "k_trace1": [k for k in out if k.startswith("k_trace") and counting(k) and k.endswith("#b1")],
                 "k_trace2": [k for k in out if k.startswith("k_trace") and counting(k) and k.endswith("#b2")],
                 "k_trace3": [k for k in out if k.startswith("k_trace") and counting(k) and k.endswith("#b3")]}
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import csrc_digest
        t = {"csrc_sha256_16": csrc_digest(),
             "_note": "HBM-side bytes per launch = 2 x FETCH_SIZE KiB x 1024 + WRITE_SIZE KiB x 1024, separate rocprofv3 --pmc passes over "
                      "`python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras` (scripts/prof_round.sh), launches of bench.py's roofline pass; the x2 is the guide's gfx950 "
                      "correction (128-B requests tallied at 64 B), calibrated for this access pattern with tools/gather_ceiling (7 x 16 B "
                      "from one scattered 128-B line per lane: one TCC_EA0_RDREQ and 64 B of FETCH_SIZE per line); Infinity-Cache hits "
                      "are counted, not excluded",
             workload: {n: (hbm(ks[0]) if ks else None) for n, ks in names.items()},
             # accesses the vector L1 (TCP) counted for the same launches: lanes of a wave that read one address are merged
             workload + ":tcp_accesses": {n: (out.get(ks[0], {}).get("TCP_TOTAL_CACHE_ACCESSES_sum") if ks else None) for n, ks in names.items()},
             # what the counters say about the same launches (bench.py derives roofline.limited_by from these, never from prose)
             workload + ":counters": {n: ({"kernel": ks[0], "avg_us": out[ks[0]].get("avg_us"), "gpu_cycles": out[ks[0]].get("GRBM_GUI_ACTIVE"),
                                           "valu_busy_frac": out[ks[0]].get("valu_busy_frac"), "valu_lane_util_frac": out[ks[0]].get("valu_lane_util_frac"),
                                           "l2_hit_frac": out[ks[0]].get("l2_hit_frac"), "wait_any_frac": out[ks[0]].get("wait_any_frac"),
                                           "l2_read_latency_cycles": out[ks[0]].get("l2_read_latency_cycles"),
                                           "tcp_accesses": out[ks[0]].get("TCP_TOTAL_CACHE_ACCESSES_sum"), "tcp_tcc_read_req": out[ks[0]].get("TCP_TCC_READ_REQ_sum"),
                                           "waves": out[ks[0]].get("SQ_WAVES"), "valu_insts": out[ks[0]].get("SQ_INSTS_VALU")} if ks else None)
                                      for n, ks in names.items()}}
        open(sys.argv[3], "w").write(json.dumps(t, indent=1) + "\n")
    for k in sorted(out, key=lambda k: -out[k].get("total_ms", 0)):
        o = out[k]
        print(f"{k:34s} n={o.get('launches',0):3d} avg={o.get('avg_us',0):8.1f}us valu_busy={o.get('valu_busy_frac',0):.2f} "
              f"lane_util={o.get('valu_lane_util_frac',0):.2f} l2hit={o.get('l2_hit_frac',0):.2f} fetch={o.get('fetch_bytes_raw',0)/1e9:.2f}GB "
              f"write={o.get('write_bytes_raw',0)/1e9:.2f}GB wait={o.get('wait_any_frac',0):.2f} lat={o.get('l2_read_latency_cycles',0):.0f} "
              f"valu/wave={o.get('SQ_INSTS_VALU',0)/max(o.get('SQ_WAVES',1),1):.0f}")


if __name__ == "__main__":
    main()
