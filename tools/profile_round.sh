#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace + stats of the default bench command, FETCH_SIZE / WRITE_SIZE passes
# (separate, as MI355X_MICROARCH.md prescribes), fp32 and fp16.  Run on the GPU box from the repo root:
#   bash tools/profile_round.sh r03
set -o pipefail
tag=${1:-r06}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $out/trace_f16 -o t --output-format csv -- python3 bench.py --no-cpu-baseline --dtype f16 > $out/bench_f16_under_rocprof.log 2>&1 || exit 1
# two full launches per pass at the views per launch bench.py's default plan gives the dtype (60 fp32 / 100 fp16)
for dt in f32 f16; do
  if [ $dt = f16 ]; then shape="--views 200 --chunk 100"; else shape="--views 120 --chunk 60"; fi
  rocprofv3 --pmc FETCH_SIZE -d $out/fetch_$dt -o c --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $shape --dtype $dt > $out/pmc_fetch_$dt.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE -d $out/write_$dt -o c --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $shape --dtype $dt > $out/pmc_write_$dt.log 2>&1 || exit 1
done
# R1 (config 2): the default plan's two launches of 50 views
rocprofv3 --pmc FETCH_SIZE -d $out/fetch_R1 -o c --output-format csv -- python3 bench.py --workload R1 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_fetch_R1.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $out/write_R1 -o c --output-format csv -- python3 bench.py --workload R1 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_write_R1.log 2>&1 || exit 1
# the trajectory legs (round 5): one pass each for the counters, the default bench for the kernel trace
for w in R2T A1; do
  rocprofv3 --pmc FETCH_SIZE -d $out/fetch_$w -o c --output-format csv -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_fetch_$w.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE -d $out/write_$w -o c --output-format csv -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_write_$w.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats -d $out/trace_$w -o t --output-format csv -- python3 bench.py --workload $w --no-cpu-baseline > $out/bench_${w}_under_rocprof.log 2>&1 || exit 1
done
# the realistic leg (round 6): fp16 maps on the trajectory scenes
for w in R2T A1; do
  rocprofv3 --pmc FETCH_SIZE -d $out/fetch_${w}_f16 -o c --output-format csv -- python3 bench.py --workload $w --dtype f16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_fetch_${w}_f16.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE -d $out/write_${w}_f16 -o c --output-format csv -- python3 bench.py --workload $w --dtype f16 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_write_${w}_f16.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats -d $out/trace_${w}_f16 -o t --output-format csv -- python3 bench.py --workload $w --dtype f16 --no-cpu-baseline > $out/bench_${w}_f16_under_rocprof.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats -d $out/trace_R4 -o t --output-format csv -- python3 bench.py --workload R4 --no-cpu-baseline --no-line-count > $out/bench_R4_under_rocprof.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $out/trace_R1 -o t --output-format csv -- python3 bench.py --workload R1 --no-cpu-baseline > $out/bench_R1_under_rocprof.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $out/trace_entry_parity -o t --output-format csv -- python3 bench.py --entry parity --no-cpu-baseline > $out/bench_entry_parity_under_rocprof.log 2>&1 || exit 1
python3 tools/summarize_prof.py $out/trace $out/trace_f16 $out/fetch_f32 $out/write_f32 $out/fetch_f16 $out/write_f16 $out/fetch_R1 $out/write_R1 $out/trace_R4 $out/trace_R1 $out/trace_R2T $out/fetch_R2T $out/write_R2T $out/trace_A1 $out/fetch_A1 $out/write_A1 $out/trace_R2T_f16 $out/fetch_R2T_f16 $out/write_R2T_f16 $out/trace_A1_f16 $out/fetch_A1_f16 $out/write_A1_f16 > $out/summary.txt
for f in $out/*.log; do echo "== $f"; grep '^{' $f | tail -1 | cut -c1-400; done >> $out/summary.txt
python3 bench.py --write-pmc-json $out gpurun_out/${tag}_pmc_traffic.json >> $out/summary.txt
# keep the merge small: the raw traces are large
find $out -name "*kernel_trace.csv" -delete
