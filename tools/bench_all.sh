#!/bin/bash
# Experiment tooling (GPU box): the BASELINE configs through bench.py, one JSON file each under gpurun_out/<tag>_*.json
TAG="${1:-r02}"
O=gpurun_out
mkdir -p $O
python bench.py --steps 5 --warmup 2 > $O/${TAG}_ws.json 2> $O/${TAG}_ws.err
python bench.py --graph livejournal --with-idx --steps 2 --warmup 1 > $O/${TAG}_lj_idx.json 2> $O/${TAG}_lj_idx.err
python bench.py --graph twitter2010 --with-idx --steps 1 --warmup 1 --queries 48 --no-cpu > $O/${TAG}_tw_idx.json 2> $O/${TAG}_tw_idx.err
python bench.py --graph twitter2010 --with-idx --topk 500 --steps 1 --warmup 1 --queries 32 > $O/${TAG}_tw_topk.json 2> $O/${TAG}_tw_topk.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu --no-accuracy --no-variants > $O/${TAG}_ws_torchrun1.json 2> $O/${TAG}_ws_torchrun1.err
for f in ws lj_idx tw_idx tw_topk ws_torchrun1; do echo "== $f"; tail -c 600 $O/${TAG}_$f.err | grep -v amdgpu.ids | tail -3; python - <<PY
import json
try:
    d = json.loads(open("$O/${TAG}_$f.json").read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus", "scaling")}, "batch", d["config"].get("batch"), "frac", r.get("frac"), r.get("avg_ms_by_kernel"))
    print("  phases", {k: round(v, 1) for k, v in (d.get("phases") or {}).items() if k.endswith("_ms") or k.endswith("_s")})
    print("  cpu", d.get("cpu_baseline"), d.get("cpu_baseline_note"), "acc", d.get("accuracy"))
except Exception as e:
    print("no json:", e)
PY
done
