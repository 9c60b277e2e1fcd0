#!/bin/bash
# GPU box, builder run: configs 3 and 4 as stand-alone bench lines WITH their CPU baselines (one oracle query on the
# Twitter-2010-sized graph is more than a minute of one core, so the default run leaves it out)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
python3 bench.py --graph livejournal --with-idx --steps 2 --warmup 1 --cpu-seconds 20 --cpu-threads 0 --no-accuracy > gpurun_out/bl_lj_idx.json 2> gpurun_out/bl_lj_idx.err
python3 bench.py --graph twitter2010 --with-idx --steps 1 --warmup 1 --cpu-seconds 100 --cpu-threads 0 --no-accuracy > gpurun_out/bl_tw_idx.json 2> gpurun_out/bl_tw_idx.err
python3 - <<'PY'
import json
for f in ("lj", "tw"):
    try:
        d = json.loads(open(f"gpurun_out/bl_{f}_idx.json").read().strip().splitlines()[-1])
        print(f, "%.1f q/s" % d["value"], "frac %.3f" % d["roofline"]["frac"], d["roofline"]["algorithmic_counts"], d.get("cpu_baseline"))
    except Exception as e:
        print(f, "failed", e)
PY
