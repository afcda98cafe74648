#!/bin/bash
# Counter evidence for the gather on trajectory scenes (round 5): the close-up call of R2T (60 consecutive frames of the dwell:
# every pixel in a split voxel), against one 60-view call of the benign R2 room.  On the GPU box from the repo root:
#   bash tools/traj_round.sh r05 [extra bench args]
set -o pipefail
tag=${1:-r05}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/traj_$tag
rm -rf $out; mkdir -p $out
step() { local name=$1; shift; echo "[traj] $name $(date +%T)"; timeout -k 10 300 "$@" > $out/$name.log 2>&1; local rc=$?; echo "[traj] $name rc $rc"
         if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[traj] $name timed out: stopping"; exit $rc; fi; return 0; }
for w in R2T R2; do
  one="python3 bench.py --workload $w --views 60 --min-calls 1 --steps 1 --warmup 0 --no-cpu-baseline $*"
  step trace_$w rocprofv3 --kernel-trace --stats -d $out/trace_$w -o t --output-format csv -- $one
  step fetch_$w rocprofv3 --pmc FETCH_SIZE -d $out/fetch_$w -o c --output-format csv -- $one
  step write_$w rocprofv3 --pmc WRITE_SIZE -d $out/write_$w -o c --output-format csv -- $one
  step sq1_$w rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $out/sq1_$w -o c --output-format csv -- $one
  step sq2_$w rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU -d $out/sq2_$w -o c --output-format csv -- $one
  step tcc_$w rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $out/tcc_$w -o c --output-format csv -- $one
done
python3 tools/summarize_prof.py $out/trace_R2T $out/fetch_R2T $out/write_R2T $out/sq1_R2T $out/sq2_R2T $out/tcc_R2T \
        $out/trace_R2 $out/fetch_R2 $out/write_R2 $out/sq1_R2 $out/sq2_R2 $out/tcc_R2 > $out/summary.txt 2>&1
find $out -name "*kernel_trace.csv" -size +5M -delete
grep -v "^k_stream\|^k_build\|^k_block\|^k_ws" $out/summary.txt | head -120
