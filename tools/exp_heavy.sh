#!/bin/bash
set -o pipefail
out=gpurun_out/ab_heavy.log
: > $out
for t in 0 1000 600 300 150 80; do
  echo "== HEAVY_T=$t" >> $out
  if [ $t = 0 ]; then unset VOXPROJ_HEAVY_T; else export VOXPROJ_HEAVY_T=$t; fi
  python tools/probe_ab.py tools/ab/libA.so --f16 2>&1 | grep "round 2" >> $out || exit 1
  python tools/probe_ab.py tools/ab/libA.so --f16 --pipeline 2>&1 | grep "round 2" >> $out || exit 1
  python tools/probe_ab.py tools/ab/libA.so 2>&1 | grep "round 2" >> $out || exit 1
done
