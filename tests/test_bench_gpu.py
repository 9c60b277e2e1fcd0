"""bench.py end to end on a small graph: the JSON line carries the driver's contract (metric, value, n_gpus, roofline with
algorithmic bytes / HIP-event durations, cpu_baseline from the oracle), plain and under torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_contract_small_graph():
    d = _run([sys.executable, "bench.py", "--graph", "small", "--queries", "64", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["vs_baseline"] is None
    assert "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["algorithmic_counts"].startswith("sequential FIFO oracle")
    assert r["fifo_relaxations_per_query"] <= r["gpu_relaxations_per_query"]   # the schedule's extra work is not credited
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["accuracy"]["holds"] and d["accuracy"]["cpu_vs_gpu_exact_linf"] < 1e-9


def test_bench_topk_and_torchrun_rank():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", "29633", "bench.py", "--gpus", "1", "--graph", "small", "--queries", "32", "--topk", "50", "--with-idx",
              "--steps", "1", "--warmup", "1", "--scaling", "strong"])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"]["k"] == 50 and d["value"] > 0
    assert d["config"]["gather_bytes_per_step"] == 32 * 50 * 12
    # the top-k line carries a roofline of its pushes on FIFO top-k push counts (oracle: orc_topk_push_counts with the GPU run's rounds)
    r = d["roofline"]
    assert r is not None and r["bound"] == "hbm" and 0 < r["frac"] < 1 and "FIFO top-k pushes" in r["algorithmic_counts"]


def test_bench_line_names_the_walk_bound_and_its_sources():
    d = _run([sys.executable, "bench.py", "--graph", "small", "--queries", "64", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-accuracy", "--no-variants"])
    w = d["walk_bound"]  # the dominant kernel of the headline step: gathers per second, not an HBM fraction
    assert w["gathers_per_s"] > 0 and w["steps"] > 0 and "k_walk_dg" in w["kernel"]
