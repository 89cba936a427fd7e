#!/bin/bash
# rocprofv3 passes for round 1 (run on the GPU box from the repo root)
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r01
mkdir -p $OUT
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
echo trace done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
echo fetch done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
echo write done
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $B > $OUT/pmc_l2.log 2>&1
echo l2 done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
echo sq done
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $B > $OUT/pmc_grbm.log 2>&1
echo grbm done
find $OUT -name "*.csv" | head -50
du -sh $OUT
