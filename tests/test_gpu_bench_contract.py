"""bench.py's output contract on the plumbing workload (S0: 10k voxels x 8 views x 64x64x32): ONE JSON line with the
metric, the roofline object and the CPU baseline at N=1; the two-rank code path (views r::2 + all-reduce) rehearsed
with gloo on one GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _last_json(out):
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "S0", "--steps", "2", "--warmup", "1",
                        "--cpu-views", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "Mvoxel-views/sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert d["value"] > 0 and abs(d["value"] - 10000 * 8 / (d["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * d["value"]      # ms_per_step is printed to 3 decimals
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["avg_launch_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]
    assert d["box_miss_voxels"] == 0


def test_bench_two_ranks_gloo_rehearsal():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--workload", "S0", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo",
                        "--single-device"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "cpu_baseline" not in d
    assert "all-reduce" in d["config"]["parallelism"]
    assert d["hit_pixels_per_step"] > 0 and d["value"] > 0
