#!/bin/bash
# fp16 gather A/B: rows in flight per wavefront x march occupancy cap (LDS reservation, KiB)
set -o pipefail
out=gpurun_out/r2_f16_sweep.log
: > $out
for u in 4 8 16; do
  for lds in default 26 20 13 0; do
    if [ "$lds" = default ]; then unset VOXPROJ_FH_LDS_KB; else export VOXPROJ_FH_LDS_KB=$lds; fi
    echo "== U=$u LDS=$lds" >> $out
    VOXPROJ_F16_U=$u timeout -k 10 240 python bench.py --dtype f16 --pool-tries 1 --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['measured_stream_read_gbs'])" >> $out || exit 1
  done
done
