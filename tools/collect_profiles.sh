#!/bin/bash
tag=${1:-r06}
# copies the summaries of tools/profile_round.sh $tag + tools/run_benches.sh (merged back under gpurun_out/) into profiles/
P=profiles
cp gpurun_out/${tag}_pmc_traffic.json $P/${tag}_pmc_traffic.json
cp gpurun_out/prof_$tag/trace_f32/t_kernel_stats.csv $P/${tag}_kernel_stats.csv; cp gpurun_out/prof_$tag/trace_f16/t_kernel_stats.csv $P/${tag}_kernel_stats_f16.csv; cp gpurun_out/prof_$tag/trace_R4/t_kernel_stats.csv $P/${tag}_kernel_stats_R4.csv
cp gpurun_out/prof_$tag/trace_R1/t_kernel_stats.csv $P/${tag}_kernel_stats_R1.csv; cp gpurun_out/prof_$tag/trace_R2T/t_kernel_stats.csv $P/${tag}_kernel_stats_R2T.csv; cp gpurun_out/prof_$tag/trace_A1/t_kernel_stats.csv $P/${tag}_kernel_stats_A1.csv; cp gpurun_out/prof_$tag/trace_entry_parity/t_kernel_stats.csv $P/${tag}_kernel_stats_entry_parity.csv
cp gpurun_out/prof_$tag/trace_R2T_f16/t_kernel_stats.csv $P/${tag}_kernel_stats_R2T_f16.csv; cp gpurun_out/prof_$tag/trace_A1_f16/t_kernel_stats.csv $P/${tag}_kernel_stats_A1_f16.csv
for n in default f16 R1 R1_one_call R2T A1 R2T_f16 A1_f16 R4 entry_parity entry_parity_unpipelined entry_fast serial rehearse_dist; do grep '^{' gpurun_out/${tag}_bench_$n.log | tail -1 > $P/${tag}_bench_$n.json; done
grep '^{' gpurun_out/${tag}_bench_prep.log | tail -1 > $P/${tag}_bench_prep.json; { echo "# ONE view per blocking project_features_cuda call (compiled front, same occupancy tensor), tools/bench_dropin.py --phases (tools/one_view_round.sh ${tag}_final dropin), one box:"
  echo "# trajectory frames of the A1 / R2T legs (0,30,59 = close-up dwell; 100,150,200 = walk, look through the opening, clutter) and the benign rooms (16 views each)."
  echo "# Round 5 on the same frames (profiles/r05_dropin_trajectory.log, committed build): A1 0.3973 / 0.2540, R2T 0.3841 / 0.2527, R2 0.3018, R1 0.1380 ms per call."
  for f in dropin_A1_0_30_59 dropin_A1_100_150_200 dropin_R2T_0_30_59 dropin_R2T_100_150_200 dropin_benign; do grep -v amdgpu gpurun_out/one_view_${tag}_final/$f.log; done; } > $P/${tag}_bench_dropin.log
grep '^{' gpurun_out/${tag}_bench_stage5.log | tail -1 > $P/${tag}_bench_stage5.json; grep '^{' gpurun_out/${tag}_bench_entry_files.log | tail -1 > $P/${tag}_bench_entry_files.json
tail -n 3 gpurun_out/${tag}_gputest.log > $P/${tag}_gputest_tail.txt
grep '^{' gpurun_out/prof_$tag/bench_f32_under_rocprof.log | tail -1 > $P/${tag}_bench_under_rocprof.json
{ echo "# rocprofv3 summaries, round ${tag#r0}, final build (MI355X, ROCm 7.2), produced by tools/profile_round.sh $tag + tools/summarize_prof.py"
  echo "# commands (cd \$GRAFT_REPO_ROOT; TMPDIR=/tmp):"
  echo "#   rocprofv3 --kernel-trace --stats -d <out>/trace     -- python3 bench.py --no-cpu-baseline              (same run's JSON line: profiles/${tag}_bench_under_rocprof.json)"
  echo "#   rocprofv3 --kernel-trace --stats -d <out>/trace_f16 -- python3 bench.py --no-cpu-baseline --dtype f16"
  echo "#   rocprofv3 --pmc FETCH_SIZE -d <out>/fetch_<dt> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --views 120 --chunk 60   |   --views 200 --chunk 100 --dtype f16"
  echo "#   rocprofv3 --pmc WRITE_SIZE -d <out>/write_<dt> -- (same)      (separate passes; counter passes serialise kernels: their durations are not the bench's)"
  echo "#   rocprofv3 --kernel-trace --stats -d <out>/trace_R4  -- python3 bench.py --workload R4 --no-cpu-baseline"
  echo "# k_gather averages over ALL launches of a process (placement pass 10, pre-pass 5, warm-up 5, timed 15, after-pass 10 = 45 at 5 calls of 60 views per pass);"
  echo "# FETCH_SIZE is in KB and counts 64 B per 128-B request for 16-B-per-lane streaming reads on gfx950 -> doubled in profiles/${tag}_pmc_traffic.json's reader"
  cat gpurun_out/prof_$tag/summary.txt | cut -c1-420; } > $P/${tag}_rocprofv3_summary.txt
