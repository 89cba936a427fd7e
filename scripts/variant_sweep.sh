#!/bin/bash
# Builds kernel variants (CPU side) or benches them (GPU side).
#   scripts/variant_sweep.sh build  "a:-DX=1" "b:-DX=2" ...     -> raytracer_amd/librtx_hip_<name>.so
#   scripts/variant_sweep.sh bench  a b ...                      -> gpurun_out/variants.txt (run on the GPU box)
mode=$1; shift
if [ "$mode" = build ]; then
  for v in "$@"; do
    name=${v%%:*}; defs=${v#*:}
    make -C raytracer_amd/csrc -s hip-variant NAME=$name DEFS="$defs" &
  done
  wait
else
  mkdir -p gpurun_out
  : > gpurun_out/variants.txt
  for name in "$@"; do
    lib=librtx_hip_$name.so; [ "$name" = base ] && lib=librtx_hip.so
    RTX_HIP_LIB=$lib timeout -k 10 90 python bench.py --no-extras --no-cpu-baseline --steps 32 ${RTX_SWEEP_ARGS:-} > gpurun_out/variant_$name.json 2> gpurun_out/variant_$name.err || { echo "$name FAILED (rc $?): stopping, nothing more is started on the GPU" >> gpurun_out/variants.txt; break; }
    python - "$name" <<'PY' >> gpurun_out/variants.txt
import json, sys
n = sys.argv[1]
j = json.load(open(f"gpurun_out/variant_{n}.json"))
s = j["extra"]["serial_kernel_ms_per_frame"]
print(f"{n:12s} {j['value']:8.1f} Mrays/s {j['ms_per_step']:6.3f} ms | serial primary {s['primary']:.3f} t1 {s['trace1']:.3f} t2 {s['trace2']:.3f} t3 {s['trace3']:.3f} | wipl {j['roofline']['work_items_per_lane_pass']}")
PY
  done
  cat gpurun_out/variants.txt
fi
