#!/bin/bash
# every bench leg of the round, one JSON line each, into gpurun_out/r2_bench_*.log
set -o pipefail
python bench.py > gpurun_out/r2_bench_default.log 2>&1 || exit 1
python bench.py --dtype f16 --no-cpu-baseline > gpurun_out/r2_bench_f16.log 2>&1 || exit 1
python bench.py --workload R1 --no-cpu-baseline > gpurun_out/r2_bench_R1.log 2>&1 || exit 1
python bench.py --workload R4 > gpurun_out/r2_bench_R4.log 2>&1 || exit 1
python bench.py --entry parity --no-cpu-baseline > gpurun_out/r2_bench_entry_parity.log 2>&1 || exit 1
python bench.py --entry fast --no-cpu-baseline > gpurun_out/r2_bench_entry_fast.log 2>&1 || exit 1
python bench.py --no-pipeline --no-cpu-baseline > gpurun_out/r2_bench_serial.log 2>&1 || exit 1
