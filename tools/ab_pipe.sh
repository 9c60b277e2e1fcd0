#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
for P in 0 1; do
  FORA_HIP_PIPELINE=$P python3 bench.py --graph livejournal --with-idx --steps 2 --warmup 1 --no-cpu --no-accuracy > gpurun_out/pipe_lj_$P.json 2>/dev/null
  FORA_HIP_PIPELINE=$P python3 bench.py --graph twitter2010 --with-idx --steps 1 --warmup 1 --queries 64 --no-cpu --no-accuracy > gpurun_out/pipe_tw_$P.json 2>/dev/null
done
python3 - <<'PY'
import json
for g in ("lj", "tw"):
    for p in (0, 1):
        try:
            d = json.loads(open(f"gpurun_out/pipe_{g}_{p}.json").read().strip().splitlines()[-1])
            print(g, "pipeline", p, "q/s %.1f" % d["value"], "batch", d["config"]["batch"])
        except Exception as e:
            print(g, p, "failed", e)
PY
