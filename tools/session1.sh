#!/bin/bash
# GPU box: whole GPU test suite, the default bench line (with the extra configs), then A/B of the wide chunk sizes.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/s1_tests.log 2>&1; echo "tests rc=$? in $(( $(date +%s) - T0 )) s"; tail -4 gpurun_out/s1_tests.log
T0=$(date +%s)
timeout 900 python3 bench.py > gpurun_out/s1_bench_default.json 2> gpurun_out/s1_bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/s1_bench_default.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/s1_bench_default.json").read().strip().splitlines()[-1])
    print("headline %.0f q/s, ms/step %.1f, roofline %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
    print("variants", json.dumps(d.get("variants", {}))[:900])
    for k, v in d.get("configs", {}).items():
        print(k, json.dumps(v)[:700])
except Exception as e:
    print("bench parse failed", e)
PY
LIBS="fora_amd/libfora_hip.so variants/lib_ept12.so variants/lib_ept16.so variants/lib_ept16c.so"
echo "== LJ idx"; timeout 900 python3 tools/pushbench.py --graph livejournal --mode idx --queries 280 --reps 2 $LIBS | cut -c1-600
echo "== TW idx"; timeout 1500 python3 tools/pushbench.py --graph twitter2010 --mode idx --queries 28 --reps 1 $LIBS | cut -c1-600
