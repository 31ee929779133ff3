"""bench.py's own launch logic, on CPU: a bare `python bench.py --gpus N` (no torchrun, WORLD_SIZE unset) must start
N ranks itself, the ranks must find each other, and the line must state the rank count a real all-reduce saw.
(--spawn-selftest swaps the GPU work for a gloo rendezvous: this box has no GPU; the launch path is the same.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


@pytest.mark.parametrize("n", [2, 3])
def test_bare_gpus_n_spawns_n_ranks(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--spawn-selftest"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == n and res["rccl_ranks"] == n and res["spawned_by_bench"] is True


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-traffic"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_config_presets():
    sys.path.insert(0, ROOT)
    import bench

    a = bench.parse_args(["--config", "3"])
    assert (a.k, a.hash, a.reads_per_gpu) == (31, True, 125_000_000)
    a = bench.parse_args(["--config", "4"])
    assert (a.k, a.histogram, a.reads_per_gpu) == (31, 20, 125_000_000)
    a = bench.parse_args(["--config", "2", "-k", "63"])
    assert (a.k, a.reads_per_gpu) == (63, 100_000_000)
    a = bench.parse_args([])
    assert (a.k, a.hash, a.reads_per_gpu, a.gpus) == (31, False, 100_000_000, 1)


def test_traffic_digest():
    sys.path.insert(0, ROOT)
    import bench

    raw = {"FETCH_SIZE": {"calib": {"c": [7324218.75] * 3}, "scan": {"main": [7400000.0] * 3, "second": [20.0] * 3}},
           "WRITE_SIZE": {"scan": {"main": [40.0] * 3}}}
    td = bench.traffic_from_counters(raw, 15.0e9, 3)
    assert abs(td["read_correction"] - 2.0) < 1e-6
    assert abs(td["bytes_per_step"] - ((7400000.0 + 20.0) * 1024 * 2.0 + 40.0 * 1024)) < 1.0
