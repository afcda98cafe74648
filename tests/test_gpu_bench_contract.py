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
    assert "reduce to rank 0" in d["config"]["parallelism"]
    assert d["hit_pixels_per_step"] > 0 and d["value"] > 0


def test_bench_two_ranks_allreduce_variant():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29535", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--workload", "S0", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo",
                        "--single-device", "--collective", "allreduce", "--no-overlap-reduce"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]          # bench verifies reduced counts exactly and sums against the ranks' own
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and "all-reduce" in d["config"]["parallelism"]


def _check_common(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["value"] > 0 and d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3


def test_bench_config2_r1_leg():
    # BASELINE config 2 at its own shape (80k voxels, 484x274x512), a 12-view slice of it to keep the test short
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "R1", "--views", "12", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    _check_common(d)
    assert d["config"]["workload"].startswith("R1: 80000 voxels x 12 views x 484x274x512") and d["box_miss_voxels"] == 0


def test_bench_config5_rgb_leg():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "R4", "--views", "16", "--chunk", "8",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    _check_common(d)
    assert d["config"]["workload"].startswith("R4: 500000 voxels x 16 views") and d["voxel_view_hits_per_step"] > 500000


@pytest.mark.parametrize("mode", ["parity", "fast"])
def test_bench_entry_point_leg(mode):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--entry", mode, "--views", "10", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    _check_common(d)
    e = d["entry"]
    assert e["mode"] == mode and e["ms_per_view"] > 0 and e["dropin_ms_per_view"] > 0 and e["rows_out"] > 1000
    assert "R1 through the entry point" in d["config"]["workload"]


def test_bench_contiguous_allocation_option():
    # --alloc contiguous: resident buffers from hipExtMallocWithFlags(hipDeviceMallocContiguous), reported in the line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "S0", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--alloc", "contiguous"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert set(d["pool_placement"]["allocation"].values()) <= {"contiguous", "default"}
    assert d["pool_placement"]["allocation"]["feature_pool"] == "contiguous" and d["value"] > 0
