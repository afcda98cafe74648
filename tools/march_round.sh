#!/bin/bash
# SQ counters + kernel trace of k_first_hit running alone, for one or more builds of the library:
#   bash tools/march_round.sh <tag> <lib.so> [<lib.so> ...]
set -o pipefail
tag=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/march_$tag
rm -rf $out; mkdir -p $out
cmd="python3 bench.py --no-pipeline --steps 1 --warmup 0 --views 60 --no-cpu-baseline"
i=0
for lib in "$@"; do
  i=$((i+1)); export VOXPROJ_LIB=$PWD/$lib
  echo "[march] $lib"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace$i -o t --output-format csv -- $cmd > $out/trace$i.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $out/pmcA$i -o c --output-format csv -- $cmd > $out/pmcA$i.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD -d $out/pmcB$i -o c --output-format csv -- $cmd > $out/pmcB$i.log 2>&1 || exit 1
  echo "== $lib" >> $out/summary.txt
  python3 tools/summarize_prof.py $out/trace$i $out/pmcA$i $out/pmcB$i 2>&1 | grep "k_first_hit\|^--" >> $out/summary.txt
done
find $out -name "*kernel_trace.csv" -delete
cat $out/summary.txt
