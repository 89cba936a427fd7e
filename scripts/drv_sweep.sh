#!/bin/bash
# the driver's own command line (python3 bench.py --gpus 1 --steps 20 --warmup 5) under different batch settings
for cfg in "8 0" "4 0" "4 4" "2 0" "5 0" "8 4"; do
  set -- $cfg
  fif=""; [ "$2" != "0" ] && fif="RTX_FRAMES_IN_FLIGHT=$2"
  for rep in 1 2; do
  out=$(env RTX_BATCH=$1 $fif timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep '^{')
  echo "batch $1 in flight $2: $(echo $out | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f Mrays/s %.3f ms/frame (fif %s, batch %s)' % (d['value'], d['ms_per_step'], d['config']['frames_in_flight'], d['config']['frames_per_batch']))")"
  done
done
