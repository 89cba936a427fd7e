#!/bin/bash
# BASELINE.json configs[4] at its real size on the GPU box: 100 M explicit triangles (1 M base mesh x 100), orbit camera.
# Run from the repo root: gpurun --timeout 1200 -- 'bash scripts/gpu_cfg5_full.sh'
set -e
mkdir -p gpurun_out
RTX_TEST_ORBIT_TRIS=100000000 timeout -k 10 1100 python -u -m pytest tests/test_gpu_configs.py -m gpu -x -v -k cfg5 \
    --timeout 1000 --timeout-method thread 2>&1 | tee gpurun_out/cfg5_full.log
