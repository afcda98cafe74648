#!/bin/bash
set -o pipefail
out=gpurun_out/r2_f16_exp2.log
: > $out
run() { echo "== $EXTRA $*" >> $out; env "$@" python bench.py --pool-tries 1 --no-cpu-baseline --steps 3 --warmup 1 $EXTRA 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['measured_stream_read_gbs'])" >> $out; }
EXTRA="--dtype f16 --no-pipeline" run VOXPROJ_F16_U=4
EXTRA="--dtype f16 --no-pipeline" run VOXPROJ_F16_U=8
EXTRA="--dtype f16" run VOXPROJ_F16_U=4
EXTRA="--dtype f16" run VOXPROJ_F16_U=8
EXTRA="--dtype f16" run VOXPROJ_F16_U=4 VOXPROJ_FH_LDS_KB=26
EXTRA="--dtype f32" run X=1
