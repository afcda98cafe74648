#!/bin/bash
# every bench leg of the round, one JSON line each, into gpurun_out/<tag>_bench_*.log   (bash tools/run_benches.sh r04)
set -o pipefail
tag=${1:-r06}
o=gpurun_out
python3 bench.py > $o/${tag}_bench_default.log 2>&1 || exit 1
python3 bench.py --dtype f16 --no-cpu-baseline > $o/${tag}_bench_f16.log 2>&1 || exit 1
python3 bench.py --workload R1 --no-cpu-baseline > $o/${tag}_bench_R1.log 2>&1 || exit 1
python3 bench.py --workload R1 --min-calls 1 --chunk 100 --no-cpu-baseline > $o/${tag}_bench_R1_one_call.log 2>&1 || exit 1
python3 bench.py --workload R2T --no-cpu-baseline > $o/${tag}_bench_R2T.log 2>&1 || exit 1
python3 bench.py --workload A1 --no-cpu-baseline > $o/${tag}_bench_A1.log 2>&1 || exit 1
# the realistic leg (round 6): features are fp16 at rest (script/extract_lseg_features.py:97) and real captures are trajectories
python3 bench.py --workload R2T --dtype f16 --no-cpu-baseline > $o/${tag}_bench_R2T_f16.log 2>&1 || exit 1
python3 bench.py --workload A1 --dtype f16 --no-cpu-baseline > $o/${tag}_bench_A1_f16.log 2>&1 || exit 1
python3 bench.py --workload R4 > $o/${tag}_bench_R4.log 2>&1 || exit 1
python3 bench.py --entry parity --no-cpu-baseline > $o/${tag}_bench_entry_parity.log 2>&1 || exit 1
python3 bench.py --entry parity --entry-no-pipeline --no-cpu-baseline > $o/${tag}_bench_entry_parity_unpipelined.log 2>&1 || exit 1
python3 bench.py --entry fast --no-cpu-baseline > $o/${tag}_bench_entry_fast.log 2>&1 || exit 1
python3 bench.py --no-pipeline --no-cpu-baseline > $o/${tag}_bench_serial.log 2>&1 || exit 1
python3 bench.py --rehearse-dist --no-cpu-baseline > $o/${tag}_bench_rehearse_dist.log 2>&1 || exit 1
