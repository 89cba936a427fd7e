#!/bin/bash
# A/B of one environment switch on the default bench workload: scripts/ab_env.sh <tag> <VAR> <value> [<value> ...]
tag=$1; var=$2; shift 2
for v in "$@"; do
  out=gpurun_out/ab_${tag}_${var}_${v}
  env $var=$v timeout -k 10 240 python bench.py --steps 32 --warmup 4 --no-extras --no-cpu-baseline ${AB_ARGS:-} > $out.json 2> $out.err || { echo "$var=$v FAILED"; tail -3 $out.err; exit 1; }
  python - "$var=$v" $out.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
s = d["extra"]["serial_kernel_ms_per_frame"]
r = d["roofline"]
print(f"{sys.argv[1]:28s} {d['value']:7.0f} Mrays/s {d['ms_per_step']:6.3f} ms/frame | serial " + " ".join(f"{k}={v:.2f}" for k, v in s.items()) +
      f" | steps/ray {r['node_steps_per_ray'][1]:.2f} leaf/ray {r['leaf_tests_per_ray'][1]:.2f} wi/lp {r['work_items_per_lane_pass'][1]}")
PY
done
