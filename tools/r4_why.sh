#!/bin/bash
# What bounds k_project_colors: address-translation and L1 counters of the lane = voxel-ID kernel (a build kept under build/ab/)
# against the curve-ordered one, both in one process on one image allocation.  On the GPU box from the repo root:
#   bash tools/r4_why.sh r05 build/ab/libvoxproj_base.so
set -o pipefail
tag=${1:-r05}; old=${2:-build/ab/libvoxproj_base.so}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r4why_$tag
rm -rf $out; mkdir -p $out
step() { local name=$1; shift; echo "[r4why] $name $(date +%T)"; timeout -k 10 300 "$@" > $out/$name.log 2>&1; local rc=$?; echo "[r4why] $name rc $rc"
         if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[r4why] $name timed out: stopping"; exit $rc; fi; return 0; }
one="python3 tools/probe_colors.py $old 3d-semantic-segmentation_amd/libvoxproj.so --rounds 1"
step tlb1 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum -d $out/tlb1 -o c --output-format csv -- $one
step tlb2 rocprofv3 --pmc TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum -d $out/tlb2 -o c --output-format csv -- $one
step l1a rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum -d $out/l1a -o c --output-format csv -- $one
step l1b rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum -d $out/l1b -o c --output-format csv -- $one
# (TA_* + GRBM in one pass: "exceeds the capabilities of the hardware" -- rocprofv3 aborts and then sits until the timeout; not collected)
python3 tools/summarize_prof.py $out/tlb1 $out/tlb2 $out/l1a $out/l1b > $out/summary.txt 2>&1
grep "k_project_colors\|^==" $out/summary.txt
