#!/bin/bash
# GPU box: parity tests, then bench.py A/B of the online walk kernels (degree-grouped copy on / off).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_hip_parity_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/ab_tests.log
ARGS="--steps 5 --warmup 2 --no-cpu --no-accuracy --no-variants"
python3 bench.py $ARGS > gpurun_out/ab_dg1.json 2> gpurun_out/ab_dg1.err; tail -c 1500 gpurun_out/ab_dg1.json
FORA_HIP_WALK_DG=0 python3 bench.py $ARGS > gpurun_out/ab_dg0.json 2> gpurun_out/ab_dg0.err
python3 - <<'PY'
import json
for t in ("dg1", "dg0"):
    try:
        d = json.loads(open(f"gpurun_out/ab_{t}.json").read().strip().splitlines()[-1])
        p = d["phases"]; k = d["steps"]
        print(t, "q/s %.0f" % d["value"], "walk_ms %.1f" % (p["walk_ms"] / k), "push %.1f" % ((p["push_expand_ms"] + p["push_accum_ms"] + p["push_tail_ms"]) / k), "walk_accum %.1f" % (p["walk_accum_ms"] / k), "roofline %.3f" % d["roofline"]["frac"])
    except Exception as e:
        print(t, "failed", e)
PY
