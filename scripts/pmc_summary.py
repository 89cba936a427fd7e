#!/usr/bin/env python3
"""Summarise rocprofv3 output of scripts/prof_pmc.sh: per kernel, the launch count, average
duration and the per-launch average of every collected counter, plus derived ratios.

usage: python scripts/pmc_summary.py gpurun_out/prof_<tag> [out.json]
"""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    name = name.replace("void ", "")
    return name.strip()


def main():
    root = sys.argv[1]
    out = {}
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(root, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        out[k] = {"launches": len(v), "avg_us": sum(v) / len(v), "total_ms": sum(v) / 1e3}
    ctr = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (d, c), v in per_dispatch.items():
            if c.startswith("GRBM_"):
                v /= 8.0                   # reported summed over the 8 XCDs: a clock, not a sum
            ctr[names[d]][c].append(v)
    for k, cs in ctr.items():
        o = out.setdefault(k, {})
        for c, v in cs.items():
            o[c] = sum(v) / len(v)
    for k, o in out.items():
        g = o.get
        if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
            o["valu_busy_frac"] = g("SQ_ACTIVE_INST_VALU") * 4 / (1024 * g("GRBM_GUI_ACTIVE"))
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            o["lane_util_frac"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
            o["l2_hit_frac"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        if g("FETCH_SIZE") is not None:
            o["hbm_fetch_bytes"] = g("FETCH_SIZE") * 1024.0      # FETCH_SIZE is in KiB
        if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
            o["wait_any_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
            o["l2_read_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
    text = json.dumps(out, indent=1, sort_keys=True)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    for k in sorted(out, key=lambda k: -out[k].get("total_ms", 0)):
        o = out[k]
        print(f"{k:40s} n={o.get('launches',0):4d} avg={o.get('avg_us',0):9.1f}us  valu_busy={o.get('valu_busy_frac',0):.2f} "
              f"lane_util={o.get('lane_util_frac',0):.2f} l2hit={o.get('l2_hit_frac',0):.2f} fetch={o.get('hbm_fetch_bytes',0)/1e9:.2f}GB "
              f"wait={o.get('wait_any_frac',0):.2f} lat={o.get('l2_read_latency_cycles',0):.0f} "
              f"valu/wave={o.get('SQ_INSTS_VALU',0)/max(o.get('SQ_WAVES',1),1):.0f} vmem/wave={o.get('SQ_INSTS_VMEM_RD',0)/max(o.get('SQ_WAVES',1),1):.0f}")


if __name__ == "__main__":
    main()
