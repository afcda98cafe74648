#!/bin/bash
# Second counter round for the gather's speed levels: the WRITE side (the output-row stores cost 2.5-7 % of a launch for 0.4 %
# of its bytes and carry the pool x output-rows interaction, profiles/r03_levels_output_row_stores_ab.log) and the wave
# accounting.  Same protocol as tools/levels_round.sh; writes gpurun_out/levels2/.
set -o pipefail
allocs=${1:-5}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/levels2
rm -rf $out; mkdir -p $out
t_start=$SECONDS
pass() {   # pass <tag> <counters...>
  local tag=$1; shift
  if [ $((SECONDS - t_start)) -gt ${LEVELS_BUDGET_S:-700} ]; then echo "[levels2] $tag: skipped, time budget used"; return 0; fi
  echo "[levels2] $tag: $(date +%T)"
  timeout -k 10 300 rocprofv3 --pmc "$@" -d $out/$tag -o c --output-format csv -- python3 tools/probe_levels.py --allocs $allocs --tag $tag --chunks 32 --windows "" --reps 2 --out-dir levels2 > $out/$tag.log 2>&1
  local rc=$?
  echo "[levels2] $tag: rc $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[levels2] $tag timed out: stopping"; exit $rc; fi
}
pass wr1 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
pass wr2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass wr3 TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_WRITE_sum
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
python3 tools/levels_table.py $out > $out/table.txt 2>&1
find $out -name "*.csv" -size +20M -delete
cat $out/table.txt
