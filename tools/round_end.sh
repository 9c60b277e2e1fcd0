#!/bin/bash
# GPU box: whole GPU test suite, smoke, the default bench line, ws profiles (kernel stats + PMC traffic)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/re_tests.log 2>&1; echo "tests rc=$? in $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/re_tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
T0=$(date +%s)
timeout 900 python3 bench.py > gpurun_out/re_bench_default.json 2> gpurun_out/re_bench_default.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/re_bench_default.json").read().strip().splitlines()[-1])
print("headline %.0f q/s, ms/step %.1f, roofline %.3f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
print({k: round(v / d["steps"], 1) for k, v in d["phases"].items() if k.endswith("_ms")})
print("dangling", {k: v for k, v in d["variants"]["dangling_rmat"].items() if k.startswith("value")})
for k, v in d.get("configs", {}).items():
    if isinstance(v, dict): print(k, "%.1f q/s" % v.get("value", -1), "frac", (v.get("roofline") or {}).get("frac"), v.get("error"))
print(d["accuracy"]["holds"], d["accuracy"]["linf_abs_err"], d["cpu_baseline"]["value"], d.get("cpu_baseline_all_cores", {}).get("value"))
PY
bash tools/profile_bench.sh r06_ws > gpurun_out/prof_r06_ws.log 2>&1; head -9 gpurun_out/prof_r06_ws/kernel_stats.csv | cut -c1-150; cat gpurun_out/prof_r06_ws/pmc_summary.txt | cut -c1-200
