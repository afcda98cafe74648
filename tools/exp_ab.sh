#!/bin/bash
set -o pipefail
out=gpurun_out/ab_worklist.log
: > $out
L="tools/ab/libA.so tools/ab/libW1.so tools/ab/libW4.so"
for mode in "--f16" "--f16 --pipeline" "" "--pipeline" "--r1" "--r1 --pipeline"; do
  echo "== $mode" >> $out
  python tools/probe_ab.py $L $mode 2>&1 | grep "round [12]" >> $out || exit 1
done
