#!/bin/bash
# the default bench N times in fresh processes on one box: value, ms/step, gather fraction of peak, stream read
for i in $(seq 1 ${1:-10}); do python bench.py --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['measured_stream_read_gbs'])"; done
