#!/bin/bash
# the reference's synchronous loop (bench.py --sync-only) under different hardware-queue / look-ahead settings
for cfg in "4 3 2" "8 6 3" "8 6 5" "16 8 4" "16 8 7" "16 12 11"; do
  set -- $cfg
  out=$(GPU_MAX_HW_QUEUES=$1 RTX_FRAMES_IN_FLIGHT=$2 RTX_LOOKAHEAD=$3 timeout -k 10 200 python bench.py --sync-only --steps 32 2>/dev/null | grep '^{')
  echo "queues $1 contexts $2 lookahead $3: $(echo $out | python -c "import json,sys; d=json.load(sys.stdin); print('%.2f ms/frame (look_ahead %s)' % (d['ms_per_step'], d['look_ahead']))")"
done
