#!/bin/bash
# Everything profiles/ needs for a round, on the GPU box from the repo root: bash tools/final_round.sh r06
# (GPU tests, rocprofv3 kernel trace + PMC traffic passes, the R4 counter round, every bench leg, the auxiliary benches).
# Afterwards, in the build container: bash tools/collect_profiles.sh r06
set -o pipefail
tag=${1:-r06}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "[final] gpu tests $(date +%T)"
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/${tag}_gputest.log 2>&1; echo "[final] pytest rc $?"; tail -n 3 gpurun_out/${tag}_gputest.log
# counter passes first: the bench legs after them find profiles/${tag}_pmc_traffic.json / ${tag}_r4_pmc.json stamped with THIS build's
# digest and report roofline.traffic (and the R4 line its L1 ceiling)
echo "[final] profile round $(date +%T)"
for part in a b c collect; do timeout -k 10 1100 bash tools/profile_round.sh $tag $part > gpurun_out/${tag}_prof_${part}_stdout.log 2>&1 || echo "[final] profile_round $part failed"; done
cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
timeout -k 10 600 bash tools/r4_round.sh $tag > gpurun_out/${tag}_r4_stdout.log 2>&1 || echo "[final] r4_round failed"
cp gpurun_out/${tag}_r4_pmc.json profiles/${tag}_r4_pmc.json
echo "[final] benches $(date +%T)"
timeout -k 10 1100 bash tools/run_benches.sh $tag || echo "[final] run_benches failed"
echo "[final] auxiliary $(date +%T)"
timeout -k 10 300 python3 tools/bench_prep.py > gpurun_out/${tag}_bench_prep.log 2>&1
timeout -k 10 300 python3 tools/bench_stage5.py > gpurun_out/${tag}_bench_stage5.log 2>&1
timeout -k 10 300 python3 tools/bench_entry_files.py 24 > gpurun_out/${tag}_bench_entry_files.log 2>&1
timeout -k 10 600 bash tools/one_view_round.sh ${tag}_final dropin > gpurun_out/${tag}_dropin_stdout.log 2>&1
for f in gpurun_out/${tag}_bench_*.log; do echo "== $f"; grep '^{' $f | tail -1 | cut -c1-300; done
echo "[final] done $(date +%T)"
