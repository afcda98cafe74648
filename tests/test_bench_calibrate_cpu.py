"""tools/bench_calibrate.choose_arm: which arm of the multi-rank step bench.py times.  Driven with an injected timer -- a script
of step durations per arm -- so the decisions are checked without ranks or GPUs (round 5's chooser picked the
slower arm on a 4-rank rehearsal whose first steps took 17 s)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_calibrate import choose_arm  # noqa: E402


class Script:
    """timed_step(arm) that replays a list of durations per arm, then the arm's steady value."""

    def __init__(self, split, whole, steady):
        self.q = {"split": list(split), "whole": list(whole)}
        self.steady = steady
        self.calls = []

    def __call__(self, arm):
        self.calls.append(arm)
        return self.q[arm].pop(0) if self.q[arm] else self.steady[arm]


def test_first_touch_steps_do_not_decide():
    # split's first two steps are 50 x slow AND agree with each other (staging set up on first touches): the settle loop is
    # satisfied after two steps, the three measured steps are the steady ones, and the minimum per arm decides
    t = Script(split=[15.0, 15.1], whole=[0.34], steady={"split": 0.30, "whole": 0.34})
    r = choose_arm(t)
    assert r["pick"] == "split" and r["ms"] == {"split": 300.0, "whole": 340.0}
    assert r["settling_steps"] == {"split": 2, "whole": 2} and r["settled"] == {"split": True, "whole": True}
    assert r["chosen_by"].startswith("calibration")
    # ... and when a 50 x step lands among the measured ones instead (four gloo ranks on one GPU: 217, 13 570, 295 ms), that arm is
    # "not in a steady state" and the steady one is timed -- a 13-s step in the timed region costs more than the arms differ by
    t = Script(split=[0.30, 0.30, 15.0], whole=[0.34, 0.34], steady={"split": 0.30, "whole": 0.34})
    r = choose_arm(t)
    assert r["pick"] == "whole" and r["chosen_by"].startswith("the steady arm") and r["samples_ms"]["split"][0] == 15000.0


def test_settling_needs_two_consecutive_steps_that_agree():
    # 17 s, 5 s, 0.3, 0.31: only the last two agree within 20 %
    t = Script(split=[17.0, 5.0, 0.30, 0.31], whole=[0.35, 0.36], steady={"split": 0.30, "whole": 0.35})
    r = choose_arm(t)
    assert r["settling_steps"] == {"split": 4, "whole": 2} and r["pick"] == "split"
    # the measured steps alternate between the arms
    measured = t.calls[6:]
    assert measured == ["split", "whole"] * 3


def test_the_faster_arm_is_picked_by_its_minimum():
    t = Script(split=[2.1, 2.1, 2.2, 2.1, 2.1], whole=[0.35, 0.35, 0.36, 0.60, 0.35], steady={"split": 2.1, "whole": 0.35})
    r = choose_arm(t)
    assert r["pick"] == "whole" and r["ms"]["whole"] == 350.0 and r["ms"]["split"] == 2100.0      # 0.60 is a hiccup, not the arm


def test_an_arm_that_never_settles_is_not_timed():
    # whole keeps jumping (a sample > 3 x its minimum among the measured steps) although its minimum is the smaller one: split
    t = Script(split=[0.3, 0.3], whole=[9.0, 1.0, 4.0, 0.5, 2.0, 0.2, 0.2, 5.0, 0.2], steady={"split": 0.3, "whole": 0.2})
    r = choose_arm(t)
    assert r["settled"]["whole"] is False and r["settling_steps"]["whole"] == 6
    assert r["pick"] == "split" and r["chosen_by"].startswith("the steady arm") and "whole had" in r["chosen_by"]


def test_both_arms_unsteady_refuses_the_pick():
    t = Script(split=[0.3, 0.3, 0.3, 2.0, 0.3], whole=[0.2, 0.2, 0.2, 0.2, 5.0], steady={"split": 0.3, "whole": 0.2})
    r = choose_arm(t)
    assert r["pick"] == "split" and r["chosen_by"].startswith("fallback: both arms")


def test_every_rank_takes_the_same_decisions():
    # the chooser has no input but the (rank-reduced) durations: the same script gives the same calls and the same pick
    a = Script(split=[1.0, 0.5, 0.5], whole=[0.4, 0.4], steady={"split": 0.5, "whole": 0.4})
    b = Script(split=[1.0, 0.5, 0.5], whole=[0.4, 0.4], steady={"split": 0.5, "whole": 0.4})
    ra, rb = choose_arm(a), choose_arm(b)
    assert a.calls == b.calls and ra == rb and ra["pick"] == "whole"
