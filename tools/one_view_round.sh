#!/bin/bash
# Round 6, the one-view blocking call on trajectory frames.  On the GPU box from the repo root:
#   bash tools/one_view_round.sh <tag> tests|sweep|counters|dropin [args]
set -o pipefail
tag=${1:-r06}; what=${2:-sweep}; shift; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/one_view_$tag
mkdir -p $out
step() { local name=$1; shift; echo "[one_view] $name $(date +%T)"; timeout -k 10 ${STEP_TIMEOUT:-420} "$@" > $out/$name.log 2>&1; local rc=$?; echo "[one_view] $name rc $rc"
         if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[one_view] $name timed out: stopping"; exit $rc; fi; return $rc; }
case $what in
tests)
  step tests python3 -m pytest tests/test_gpu_one_view.py -x -q "$@" || { tail -40 $out/tests.log; exit 1; }
  tail -5 $out/tests.log ;;
sweep)
  for w in ${WORKLOADS:-R2T A1}; do
    step sweep_$w python3 tools/probe_one_view.py --workload $w --view-ids ${VIEWS:-0,30,59,100,150,200} --reps 5 "$@" || { tail -30 $out/sweep_$w.log; exit 1; }
    grep -v amdgpu.ids $out/sweep_$w.log
  done ;;
counters)
  # k_gather_one on trajectory frames (VERDICT r5 next #1a): close-up frames 0,30,59 and walk / opening / clutter frames 100,150,200,
  # round 5's heavy role (--one-view-split 0: a workgroup of four wavefronts per voxel above 320 pixels) and round 6's parts
  for arm in heavy:0 parts:-1; do
    an=${arm%%:*}; av=${arm#*:}
    for v in 0,30,59 100,150,200; do
      n=${an}_${v//,/_}
      one="python3 tools/bench_dropin.py --shape R2T --occ same --front compiled --reps 2 --view-ids $v --one-view-split $av"
      step trace_$n rocprofv3 --kernel-trace --stats -d $out/trace_$n -o t --output-format csv -- $one
      step sq1_$n rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $out/sq1_$n -o c --output-format csv -- $one
      step sq2_$n rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $out/sq2_$n -o c --output-format csv -- $one
      step fetch_$n rocprofv3 --pmc FETCH_SIZE -d $out/fetch_$n -o c --output-format csv -- $one
      step write_$n rocprofv3 --pmc WRITE_SIZE -d $out/write_$n -o c --output-format csv -- $one
    done
  done
  for arm in heavy parts; do for v in 0_30_59 100_150_200; do n=${arm}_$v
    echo "#### $arm role, R2T frames ${v//_/,}"
    python3 tools/summarize_prof.py $out/trace_$n $out/sq1_$n $out/sq2_$n $out/fetch_$n $out/write_$n | grep "^==\|^--\|^k_gather_one\|^k_combine\|^k_first_hit\|^k_worklist"
  done; done > $out/summary_counters.txt 2>&1
  find $out -name "*kernel_trace.csv" -size +5M -delete
  cat $out/summary_counters.txt ;;
trace)
  # rocprofv3 kernel stats of the probe, one process per arm: trace <workload> <view-ids> ARM [ARM ...]
  w=$1; v=$2; shift; shift
  i=0
  for arm in "$@"; do
    i=$((i+1))
    step trace_${w}_$i rocprofv3 --kernel-trace --stats -d $out/trace_${w}_$i -o t --output-format csv -- python3 tools/probe_one_view.py --workload $w --view-ids $v --reps 5 $arm
    echo "== $w frames $v arm $arm"
    python3 tools/summarize_prof.py $out/trace_${w}_$i | grep "^k_zero\|^k_first\|^k_work\|^k_gather\|^k_combine\|void k_zero"
  done
  find $out -name "*kernel_trace.csv" -size +5M -delete ;;
dropin)
  for w in R2T A1; do
    for v in 0,30,59 100,150,200; do
      step dropin_${w}_${v//,/_} python3 tools/bench_dropin.py --shape $w --occ same --front compiled --phases --view-ids $v
      grep -v amdgpu.ids $out/dropin_${w}_${v//,/_}.log
    done
  done
  step dropin_benign python3 tools/bench_dropin.py --occ same --front compiled --phases
  grep -v amdgpu.ids $out/dropin_benign.log ;;
esac
