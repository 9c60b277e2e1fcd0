#!/bin/bash
# GPU box: top-k tests, then config 5 (Twitter-2010-sized top-k) and the sweep kernels' totals
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_hip_parity_gpu.py tests/test_edge_cases_gpu.py tests/test_golden.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/ab_tests.log
python3 bench.py --graph twitter2010 --with-idx --topk 500 --queries 125 --steps 2 --warmup 1 > gpurun_out/topk_tw.json 2>/dev/null
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/topk_tw.json").read().strip().splitlines()[-1])
print("tw topk %.1f q/s" % d["value"], {k: round(v / d["steps"], 1) for k, v in d["phases"].items()})
PY
python3 tools/pushbench.py --graph twitter2010 --mode idx --queries 32 --reps 1 | cut -c1-400
python3 tools/pushbench.py --mode query --reps 3 | cut -c1-400
