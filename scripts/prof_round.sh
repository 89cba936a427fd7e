#!/bin/bash
# rocprofv3 passes over the default bench.py command (run on the GPU box from the repo root).
# $1 = output tag (gpurun_out/prof_$1). Kernel trace first, then one --pmc pass per counter group.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1
mkdir -p $OUT
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras ${RTX_PROF_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
echo trace done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
echo fetch done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
echo write done
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $B > $OUT/pmc_l2.log 2>&1
echo l2 done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
echo sq done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVES_LT_64 SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1
echo sq2 done
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp -- $B > $OUT/pmc_tcp.log 2>&1
echo tcp done
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.log 2>&1
echo grbm done
