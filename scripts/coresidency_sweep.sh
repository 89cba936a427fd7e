#!/bin/bash
# Do smaller persistent grids let launches of DIFFERENT batches share the CUs (a VALU-bound trace launch of one batch beside the
# HBM-bound shade / late-bounce launches of another)? Sweeps blocks per CU of the trace and shade grids, batches in flight and
# frames per batch on the queued path (K frames, timed region only). Run on the GPU box from the repo root.
K=${1:-96}
mkdir -p gpurun_out/cores
printf "%-46s %10s %10s\n" "trace_bpc shade_bpc late_bpc in_flight batch" "Mrays/s" "ms/frame"
run() {
  out=gpurun_out/cores/t$1_s$2_l$3_f$4_b$5.json
  env RTX_TRACE_BLOCKS_PER_CU=$1 RTX_SHADE_BLOCKS_PER_CU=$2 RTX_TRACE_BPC_LATE=$3 RTX_FRAMES_IN_FLIGHT=$4 RTX_BATCH=$5 \
    timeout -k 10 120 python bench.py --timed-only --no-extras --no-cpu-baseline --steps $K --warmup $5 > $out 2> ${out%.json}.err || { echo "FAILED $*"; tail -3 ${out%.json}.err; exit 1; }
  python - "$*" $out <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:46s} {d['value']:10.0f} {d['ms_per_step']:10.3f}", flush=True)
PY
}
run 6 2 4 3 32
run 6 2 4 3 16
run 3 1 3 3 16
run 3 1 3 4 16
run 3 1 3 6 16
run 4 1 4 4 16
run 3 2 3 4 16
run 4 2 4 3 32
run 3 1 3 3 32
run 2 1 2 6 16
