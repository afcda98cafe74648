"""`python bench.py --gpus N` starts its own N ranks (it used to run on ONE GPU and print "n_gpus": 1 when no
launcher had set WORLD_SIZE).  No GPU here: --launch-check runs the rendezvous, the view shares and the call plans only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _json_lines(out):
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


def test_bench_starts_its_own_ranks_without_a_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--launch-check"], capture_output=True,
                       text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1                                      # rank 0's line, relayed by the parent; nothing from rank 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["gpus_requested"] == 2 and d["launch_check"] is True and d["value"] is None
    assert d["views_per_rank"] == [150, 150] and d["calls_per_rank"] == [3, 3] and d["views_per_call"] == [50, 50]


def test_eight_ranks_take_the_whole_scene_in_one_call_each():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--launch-check"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_lines(r.stdout)[0]
    assert d["n_gpus"] == 8 and d["views_per_rank"] == [38, 38, 38, 38, 37, 37, 37, 37] and d["calls_per_rank"] == [1] * 8


def test_fewer_gpus_than_ranks_is_an_error_not_a_one_gpu_run():
    # this container has no GPU at all: `--gpus 2` must say so and exit non-zero, and print no JSON line
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert "visible GPU" in r.stderr


def test_a_rank_refuses_a_world_size_that_is_not_gpus():
    env = dict(_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not _json_lines(r.stdout)
