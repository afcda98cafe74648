#!/bin/bash
# Counter evidence for config 5 (R4, the RGB path: k_project_colors), VERDICT r4 next #4.  On the GPU box from the repo root:
#   bash tools/r4_round.sh r06
set -o pipefail
tag=${1:-r05}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r4_$tag
rm -rf $out; mkdir -p $out
step() { local name=$1; shift; echo "[r4] $name $(date +%T)"; timeout -k 10 300 "$@" > $out/$name.log 2>&1; local rc=$?; echo "[r4] $name rc $rc"
         if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[r4] $name timed out: stopping"; exit $rc; fi; return 0; }
one="python3 bench.py --workload R4 --steps 1 --warmup 0 --no-cpu-baseline --no-line-count"
step trace rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $one
step sq1 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $out/sq1 -o c --output-format csv -- $one
step sq2 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d $out/sq2 -o c --output-format csv -- $one
step sq3 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS -d $out/sq3 -o c --output-format csv -- $one
step fetch rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o c --output-format csv -- $one
step write rocprofv3 --pmc WRITE_SIZE -d $out/write -o c --output-format csv -- $one
step tcc rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $out/tcc -o c --output-format csv -- $one
# the L1s' miss queues (round 6: the line's roofline is stated against what they sustain): misses, their summed latency, stall cycles
step l1a rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum -d $out/l1a -o c --output-format csv -- $one
step l1b rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d $out/l1b -o c --output-format csv -- $one
python3 tools/summarize_prof.py $out/trace $out/sq1 $out/sq2 $out/sq3 $out/fetch $out/write $out/tcc $out/l1a $out/l1b > $out/summary.txt 2>&1
python3 bench.py --write-r4-pmc $out gpurun_out/${tag}_r4_pmc.json
find $out -name "*kernel_trace.csv" -size +5M -delete
grep "k_project_colors\|k_color_cells\|^==\|^--" $out/summary.txt
grep -h '^{' $out/trace.log | cut -c1-600
