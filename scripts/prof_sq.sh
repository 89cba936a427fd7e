#!/bin/bash
# two rocprofv3 passes (kernel trace, SQ counters) over a short bench run: scripts/prof_sq.sh <tag>   (RTX_HIP_LIB selects a library variant)
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras ${RTX_PROF_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVES_LT_64 SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.log 2>&1
echo done
