#!/bin/bash
# fp16 gather diagnostics: serial vs pipelined, plain vs non-temporal loads, row size
set -o pipefail
out=gpurun_out/r2_f16_exp.log
: > $out
run() { echo "== $*" >> $out; env "$@" python bench.py --pool-tries 1 --no-cpu-baseline --steps 3 --warmup 1 $EXTRA 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['phase_ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['measured_stream_read_gbs'])" >> $out; }
EXTRA="--dtype f16 --no-pipeline" run X=1
EXTRA="--dtype f16 --no-pipeline" run VOXPROJ_F16_U=4
EXTRA="--dtype f16 --no-pipeline" run VOXPROJ_F16_PLAIN=1
EXTRA="--dtype f16" run VOXPROJ_F16_PLAIN=1
EXTRA="--dtype f16 --no-pipeline --pool 16" run VOXPROJ_BENCH_C=1024
EXTRA="--dtype f32 --no-pipeline --pool 16" run VOXPROJ_BENCH_C=256
EXTRA="--dtype f32 --no-pipeline" run X=1
EXTRA="--dtype f16 --no-pipeline --chunk 16" run X=1
EXTRA="--dtype f16 --no-pipeline --chunk 8" run X=1
