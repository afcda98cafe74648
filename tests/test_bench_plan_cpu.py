"""bench.py's call plan (how a rank's views are cut into vp_project_features calls and how many maps stay resident): pure host
arithmetic, checked at the BASELINE shapes and at the per-rank view counts of 1 / 2 / 4 / 8 GPUs."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_call_plan_at_the_baseline_shapes():
    plan = _bench().plan_calls
    R2 = (548, 968, 512)
    assert plan(300, *R2, 4) == (60, 5, 60)                  # config 3: five calls of 60 views, 65 GB of maps resident (<= 64)
    assert plan(300, *R2, 2) == (100, 3, 100)                # fp16 maps: the same bytes per call
    assert plan(100, 274, 484, 512, 4) == (50, 2, 50)        # config 2: two calls of 50
    assert plan(8, 64, 64, 32, 4) == (4, 2, 8)               # config 1 (plumbing)
    # config 4: views r::G -- 150 / 75 / 38 / 37 views per rank, evened out over the rank's calls
    assert plan(150, *R2, 4) == (50, 3, 50) and plan(75, *R2, 4) == (38, 2, 38)
    assert plan(38, *R2, 4) == (19, 2, 19) and plan(37, *R2, 4) == (19, 2, 19)
    # the multi-rank step plans with min_calls = 1: 8 GPUs project their 37-38 views in ONE call, 4 and 2 GPUs are cut by memory
    assert plan(38, *R2, 4, min_calls=1) == (38, 1, 38) and plan(37, *R2, 4, min_calls=1) == (37, 1, 37)
    assert plan(75, *R2, 4, min_calls=1) == (38, 2, 38) and plan(150, *R2, 4, min_calls=1) == (50, 3, 50)
    # explicit --chunk / --min-calls / --pool
    assert plan(300, *R2, 4, chunk=32, min_calls=1) == (30, 10, 30)
    assert plan(32, *R2, 4) == (16, 2, 32) and plan(16, *R2, 4) == (8, 2, 16)       # the pool holds all of a short run's views
    assert plan(1, *R2, 4) == (1, 1, 1)
    for n in range(1, 400, 7):
        per_call, n_calls, resident = plan(n, *R2, 4)
        assert per_call * n_calls >= n > per_call * (n_calls - 1) and resident % per_call == 0 and resident >= per_call
