#!/bin/bash
# library variants on the driver's own command line and on the default one: scripts/ab_drv.sh <tag> <lib> ...   ("-" = default library)
tag=$1; shift
for lib in "$@"; do
  name=${lib//[^a-zA-Z0-9_]/}
  if [ "$lib" = "-" ]; then unset RTX_HIP_LIB; name=default; else export RTX_HIP_LIB=$lib; fi
  for cmd in "--gpus 1 --steps 20 --warmup 5" "--steps 64 --warmup 4"; do
    timeout -k 10 240 python bench.py $cmd --no-extras --no-cpu-baseline > gpurun_out/abd_${tag}_${name}.json 2> gpurun_out/abd_${tag}_${name}.err || { echo "$lib FAILED"; tail -3 gpurun_out/abd_${tag}_${name}.err; exit 1; }
    python - "$lib [$cmd]" gpurun_out/abd_${tag}_${name}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:60s} {d['value']:7.0f} Mrays/s {d['ms_per_step']:6.3f} ms/frame  device-resident {d['value_device_resident']['value']:7.0f}  device bytes {d['extra']['device_bytes']/2**30:.1f} GiB")
PY
  done
done
