#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace + stats of the bench legs, FETCH_SIZE / WRITE_SIZE passes (separate, as
# MI355X_MICROARCH.md prescribes).  On the GPU box from the repo root, in parts that each fit one 20-minute call:
#   bash tools/profile_round.sh r06 a      # R2 fp32 / fp16 + R1: kernel traces and PMC passes
#   bash tools/profile_round.sh r06 b      # the trajectory legs R2T / A1, fp32 and fp16 (round 6: the realistic leg)
#   bash tools/profile_round.sh r06 c      # kernel traces of R4 and the entry point
# then, in the build container (gpurun merges every part's output back under gpurun_out/):
#   bash tools/profile_round.sh r06 collect   # summary.txt + gpurun_out/r06_pmc_traffic.json
set -o pipefail
tag=${1:-r06}; part=${2:-a}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
run() { local name=$1; shift; echo "[prof] $name $(date +%T)"; timeout -k 10 400 "$@" > $out/$name.log 2>&1 || { echo "[prof] $name FAILED"; tail -5 $out/$name.log; exit 1; }; }
trace() { local name=$1; shift; run bench_${name}_under_rocprof rocprofv3 --kernel-trace --stats -d $out/trace_$name -o t --output-format csv -- python3 bench.py --no-cpu-baseline "$@"; }
pmc() { local name=$1; shift
  run pmc_fetch_$name rocprofv3 --pmc FETCH_SIZE -d $out/fetch_$name -o c --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@"
  run pmc_write_$name rocprofv3 --pmc WRITE_SIZE -d $out/write_$name -o c --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@"; }
case $part in
a)
  trace f32
  trace f16 --dtype f16
  # two full launches per pass at the views per launch bench.py's default plan gives the dtype (60 fp32 / 100 fp16)
  pmc f32 --views 120 --chunk 60
  pmc f16 --views 200 --chunk 100 --dtype f16
  pmc R1 --workload R1                    # config 2: the default plan's two launches of 50 views
  trace R1 --workload R1 ;;
b)
  for w in R2T A1; do
    pmc $w --workload $w
    trace $w --workload $w
    pmc ${w}_f16 --workload $w --dtype f16
    trace ${w}_f16 --workload $w --dtype f16
  done ;;
c)
  trace R4 --workload R4 --no-line-count
  trace entry_parity --entry parity ;;
collect)
  python3 tools/summarize_prof.py $out/trace_f32 $out/trace_f16 $out/fetch_f32 $out/write_f32 $out/fetch_f16 $out/write_f16 $out/fetch_R1 $out/write_R1 \
          $out/trace_R4 $out/trace_R1 $out/trace_R2T $out/fetch_R2T $out/write_R2T $out/trace_A1 $out/fetch_A1 $out/write_A1 \
          $out/trace_R2T_f16 $out/fetch_R2T_f16 $out/write_R2T_f16 $out/trace_A1_f16 $out/fetch_A1_f16 $out/write_A1_f16 $out/trace_entry_parity > $out/summary.txt
  for f in $out/*.log; do echo "== $f"; grep '^{' $f | tail -1 | cut -c1-400; done >> $out/summary.txt
  python3 bench.py --write-pmc-json $out gpurun_out/${tag}_pmc_traffic.json >> $out/summary.txt ;;
esac
# keep the merge small: the raw traces are large
find $out -name "*kernel_trace.csv" -delete
echo "[prof] part $part done $(date +%T)"
