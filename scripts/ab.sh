#!/bin/bash
# A/B of library builds on the default bench workload: scripts/ab.sh <tag> <lib> [<lib> ...]   ("-" = the default librtx_hip.so)
tag=$1; shift
for lib in "$@"; do
  name=${lib//[^a-zA-Z0-9_]/}
  if [ "$lib" = "-" ]; then unset RTX_HIP_LIB; name=default; else export RTX_HIP_LIB=$lib; fi
  timeout -k 10 240 python bench.py --steps 32 --warmup 4 --no-extras --no-cpu-baseline ${AB_ARGS:-} > gpurun_out/ab_${tag}_${name}.json 2> gpurun_out/ab_${tag}_${name}.err || { echo "$lib FAILED"; tail -3 gpurun_out/ab_${tag}_${name}.err; exit 1; }
  python - "$lib" gpurun_out/ab_${tag}_${name}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
s = d["extra"]["serial_kernel_ms_per_frame"]
r = d["roofline"]
print(f"{sys.argv[1]:28s} {d['value']:7.0f} Mrays/s {d['ms_per_step']:6.3f} ms/frame | serial " + " ".join(f"{k}={v:.2f}" for k, v in s.items()) +
      f" | steps/ray {r['node_steps_per_ray'][1]:.2f} leaf/ray {r['leaf_tests_per_ray'][1]:.2f} wi/lp {r['work_items_per_lane_pass'][1]}")
PY
done
