set -e
export GPU_MAX_HW_QUEUES=8
for v in "" _v1 _v2 _v3 _v4; do
  echo "=== variant librtx_hip$v.so (serial)"
  RTX_HIP_LIB=librtx_hip$v.so RTX_VERBOSE=0 RTX_QUICK_FIF=1 timeout -k 10 150 python -u scripts/gpu_quick.py 2>&1 | grep -E "ms/frame"
done
echo "=== pipelined, main lib"
RTX_VERBOSE=0 timeout -k 10 150 python -u scripts/gpu_quick.py 2>&1 | grep -E "ms/frame"
