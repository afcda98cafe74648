#!/bin/bash
# Evidence for the one-view blocking drop-in call, on the GPU box from the repo root:
#   bash tools/dropin_round.sh r04 [suffix]
# 1. tools/bench_dropin.py (both shapes, both occupancy modes, both fronts, with the library's per-phase HIP events);
# 2. rocprofv3 --kernel-trace --stats of the R2 / same-tensor / compiled-front loop;  3. counter passes of the same loop
# (FETCH_SIZE, WRITE_SIZE, the SQ wait / issue set), one set per pass.  A pass that times out ends the script.
set -o pipefail
tag=${1:-r04}; sfx=${2:-}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/dropin_$tag$sfx
rm -rf $out; mkdir -p $out
step() { local name=$1; shift; echo "[dropin] $name $(date +%T)"; timeout -k 10 300 "$@" > $out/$name.log 2>&1; local rc=$?; echo "[dropin] $name rc $rc"
         if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[dropin] $name timed out: stopping"; exit $rc; fi; return 0; }
step bench python3 tools/bench_dropin.py --phases
cat $out/bench.log | grep -v amdgpu.ids
one="python3 tools/bench_dropin.py --shape R2 --occ same --front compiled --reps 1 --views 16"
step trace rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $one
step trace_R1 rocprofv3 --kernel-trace --stats -d $out/trace_R1 -o t --output-format csv -- python3 tools/bench_dropin.py --shape R1 --occ same --front compiled --reps 1 --views 16
step fetch rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o c --output-format csv -- $one
step write rocprofv3 --pmc WRITE_SIZE -d $out/write -o c --output-format csv -- $one
step sq1 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $out/sq1 -o c --output-format csv -- $one
step sq2 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU -d $out/sq2 -o c --output-format csv -- $one
step sq3 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $out/sq3 -o c --output-format csv -- $one
oneR1="python3 tools/bench_dropin.py --shape R1 --occ same --front compiled --reps 1 --views 16"
step fetch_R1 rocprofv3 --pmc FETCH_SIZE -d $out/fetch_R1 -o c --output-format csv -- $oneR1
step write_R1 rocprofv3 --pmc WRITE_SIZE -d $out/write_R1 -o c --output-format csv -- $oneR1
step sq1_R1 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $out/sq1_R1 -o c --output-format csv -- $oneR1
python3 tools/summarize_prof.py $out/trace $out/trace_R1 $out/fetch $out/write $out/sq1 $out/sq2 $out/sq3 $out/fetch_R1 $out/write_R1 $out/sq1_R1 > $out/summary.txt 2>&1
find $out -name "*kernel_trace.csv" -size +5M -delete
cat $out/summary.txt | grep -v "^k_stream\|^k_build\|^k_block" | head -80
