set -e
for cfg in "8 6" "12 10" "16 14" "16 10" "24 16"; do
  set -- $cfg
  echo "=== world 8, queues $1 depth $2"
  GPU_MAX_HW_QUEUES=$1 RTX_FRAMES_IN_FLIGHT=$2 RTX_QUICK_WORLD=8 RTX_VERBOSE=0 timeout -k 10 150 python -u scripts/gpu_quick.py 2>&1 | grep -E "^ms/frame"
done
for cfg in "12 10" "16 14"; do
  set -- $cfg
  echo "=== world 1, queues $1 depth $2"
  GPU_MAX_HW_QUEUES=$1 RTX_FRAMES_IN_FLIGHT=$2 RTX_VERBOSE=0 timeout -k 10 150 python -u scripts/gpu_quick.py 2>&1 | grep -E "^ms/frame"
done
