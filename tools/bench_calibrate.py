"""Which arm of the multi-rank step does bench.py time?  

`bench.py --gpus N` has two ways to pay the scene's collective (view_sharding.project_final_call_and_reduce): `split` -- the rank's
last call cut by voxel ID, the lower rows reduced under the upper rows' gather -- and `whole`.  Which one is faster depends on the
links, the backend and the rank count, and no multi-GPU node was ever available to measure it, so the timed arm is picked by a short
untimed calibration.  Round 5's calibration (two settling steps per arm, median of three) picked the WRONG arm on a 4-rank gloo
rehearsal whose split arm takes 13-17 s on some of its steps.  This one converges instead of counting:

  * SETTLE each arm until two consecutive steps agree within `agree` (20 %), at most `settle_cap` (6) steps; the number of steps it
    took is reported per arm (`settling_steps`), and whether it converged;
  * then `reps` (3) measured steps per arm, alternating (a drift hits both alike), the MINIMUM per arm;
  * a measured sample more than `reject` (3) times its arm's minimum means that arm is not in a steady state -- four gloo ranks
    sharing one GPU showed split steps of 217, 13 570 and 295 ms in a row (profiles/r06_multi_rank_rehearsals.log): such an arm
    is not timed when the other one is steady (a 13-s step inside the timed region is worse than a few per cent between the
    arms); when BOTH arms jump the calibration REFUSES to pick and falls back to `fallback` ("split", the arm that hides traffic
    by construction).  Either way `chosen_by` says so.

`timed_step(arm)` runs one step of arm "split" | "whole" and returns its duration in seconds -- bracketed and MAX-reduced over the
ranks by the caller, so every rank sees the same numbers and takes the same decisions (the loops below have no rank-local input).
"""

ARMS = ("split", "whole")


def choose_arm(timed_step, agree=0.20, settle_cap=6, reps=3, reject=3.0, fallback="split"):
    settling, converged = {}, {}
    for arm in ARMS:
        prev, n, ok = None, 0, False
        while n < settle_cap:
            t = float(timed_step(arm))
            n += 1
            if prev is not None and abs(t - prev) <= agree * max(t, prev):
                ok = True
                break
            prev = t
        settling[arm], converged[arm] = n, ok
    samples = {arm: [] for arm in ARMS}
    for _ in range(reps):
        for arm in ARMS:
            samples[arm].append(float(timed_step(arm)))
    best = {arm: min(samples[arm]) for arm in ARMS}
    unsteady = [arm for arm in ARMS if max(samples[arm]) > reject * best[arm]]
    if len(unsteady) == 1:
        pick = "whole" if unsteady[0] == "split" else "split"
        how = (f"the steady arm: {unsteady[0]} had a calibration step more than {reject:g}x its own minimum (not in a steady state), "
               f"{pick} had none")
    elif unsteady:
        pick = fallback
        how = (f"fallback: both arms had a calibration step more than {reject:g}x their own minimum (not in a steady state); "
               f"'{fallback}' hides traffic by construction")
    else:
        pick = "split" if best["split"] <= best["whole"] else "whole"
        how = (f"calibration: the arm with the smaller minimum of {reps} untimed steps (MAX over ranks each), after settling each arm until "
               f"two consecutive steps agreed within {agree:.0%} (at most {settle_cap})")
    return {"pick": pick, "chosen_by": how, "ms": {arm: round(best[arm] * 1e3, 3) for arm in ARMS},
            "samples_ms": {arm: [round(t * 1e3, 3) for t in samples[arm]] for arm in ARMS},
            "settling_steps": settling, "settled": converged}
