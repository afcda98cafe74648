#!/bin/bash
# Multi-rank evidence a one-GPU box can give (no 8-GPU node: the driver takes the scaling curve), from the repo root:
#   bash tools/rank_round.sh r04
# 1. the share of the scene ONE rank of 2 / 4 / 8 projects (150 / 75 / 38 views of R2), through the multi-rank step of bench.py
#    over a ONE-rank RCCL communicator (--rehearse-dist): VoxelFeatureAggregator.add_views / add_final_views, both collective
#    arms in one line -- what the cut of the last call COSTS (a one-rank collective moves nothing, so not what it gains);
# 2. the whole 300-view scene with 2 and with 4 ranks started by bench.py itself (`--gpus N`, no launcher), gloo, all ranks on
#    this GPU: exactness of the reduced scene (bench.py asserts counts exactly, sums per channel) -- timings mean nothing here.
set -o pipefail
tag=${1:-r06}
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); c=d.get("collective",{}); print(json.dumps({k:d[k] for k in ("n_gpus","value","ms_per_step","hit_pixels_per_step","reduced_hit_pixels") if k in d}), json.dumps({"views_per_call":d["config"]["views_per_call"],"arms":c.get("arms"),"timed_arm":c.get("timed_arm"),"chosen_by":c.get("timed_arm_chosen_by"),"calibration":c.get("calibration"),"per_rank":c.get("per_rank"),"collectives":c.get("collectives_per_pass"),"exposed_ms":c.get("collective_ms_exposed"),"projection_ms":c.get("projection_ms_per_step"),"backend":c.get("backend"),"gather_frac":d["roofline"]["frac"],"avg_launch_ms":d["roofline"]["avg_launch_ms"]}))'
{
echo "# one rank's share through the multi-rank step, one-rank RCCL communicator (python3 bench.py --rehearse-dist --views N --no-cpu-baseline)"
for v in 150 75 38; do
  echo "## --views $v"
  timeout -k 10 400 python3 bench.py --rehearse-dist --views $v --no-cpu-baseline 2>$o/${tag}_rank_err.log | python3 -c "$pick" || { tail -5 $o/${tag}_rank_err.log; exit 1; }
done
} > $o/${tag}_rank_workloads.log 2>&1 || { cat $o/${tag}_rank_workloads.log; exit 1; }
cat $o/${tag}_rank_workloads.log
{
echo "# the whole 300-view R2 scene, ranks started by bench.py itself (python3 bench.py --gpus N --dist-backend gloo --single-device --steps 2 --warmup 0 --no-cpu-baseline); single process beside it"
timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python3 -c "$pick" || exit 1
for n in 2 4; do
  echo "## --gpus $n"
  timeout -k 10 900 python3 bench.py --gpus $n --dist-backend gloo --single-device --steps 2 --warmup 0 --no-cpu-baseline 2>$o/${tag}_rank_err.log | python3 -c "$pick" || { tail -5 $o/${tag}_rank_err.log; exit 1; }
done
} > $o/${tag}_multi_rank_rehearsals.log 2>&1 || { cat $o/${tag}_multi_rank_rehearsals.log; exit 1; }
cat $o/${tag}_multi_rank_rehearsals.log
