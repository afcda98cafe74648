#!/bin/bash
# Counter evidence for the two speed levels of k_gather.  Run on the GPU box from the repo root:
#   bash tools/levels_round.sh [allocs]
# 1. plain run of tools/probe_levels.py (the timing truth: gather per allocation at 32 and 8 views per call, the random-row
#    probe over 1 / 4 / 16 / 35 GB windows);  2. the same program under rocprofv3 --pmc, one counter set per pass (address
#    translation, TCP stalls / latency, TCC->fabric requests and stalls).  A pass that fails (e.g. a counter the block cannot
#    schedule) is reported and skipped; a pass that TIMES OUT ends the script (no further GPU step after a kill).
set -o pipefail
allocs=${1:-5}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/levels
rm -rf $out; mkdir -p $out
t_start=$SECONDS
step() {   # step <name> <command...>
  local name=$1; shift
  if [ $((SECONDS - t_start)) -gt ${LEVELS_BUDGET_S:-840} ]; then echo "[levels] $name: skipped, time budget used"; return 0; fi
  echo "[levels] $name: $(date +%T)"
  timeout -k 10 420 "$@" > $out/$name.log 2>&1
  local rc=$?
  echo "[levels] $name: rc $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[levels] $name timed out: stopping"; exit $rc; fi
  return 0
}
step bench32 python3 bench.py --no-cpu-baseline --pool-tries 3
step bench8 python3 bench.py --no-cpu-baseline --chunk 8
step plain python3 tools/probe_levels.py --allocs $allocs --tag plain
pass() {   # pass <tag> <counters...>
  local tag=$1; shift
  step $tag rocprofv3 --pmc "$@" -d $out/$tag -o c --output-format csv -- python3 tools/probe_levels.py --allocs $allocs --tag $tag --windows 4,0 --reps 1
}
pass tlb1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass tlb2 TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_LFIFO_FULL_sum
pass tlb3 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY
pass tcc1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum
pass tcc2 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_BUSY_sum
pass tcc3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 tools/levels_table.py $out > $out/table.txt 2>&1
# keep the merge small
find $out -name "*.csv" -size +20M -delete
tail -n 60 $out/table.txt
