#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" python3 tools/pushbench.py --mode query --reps 3 $ARGS $LIBS | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('%-16s bin %.1f acc %.1f tail %.1f push %.1f | walk %.1f wacc %.1f alloc %.1f | batch %.1f other %.1f | launches %.0f batchsz %d' % (d['lib'], d['bin_ms'], d['accum_ms'], d['tail_ms'], d['push_ms'], d['walk_ms'], d['walk_accum_ms'], d['walk_alloc_ms'], d['batch_ms'], d['other_ms'], d['launches'], d['batch']))
"; }
LIBS=""
ARGS="--queries 2000"
run FORA_HIP_PIPELINE=0
run FORA_HIP_PIPELINE=1
python3 - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, fora_amd
from fora_amd import synth
n, m, rp, col = synth.preset("webstanford")
for pipe in (0, 1):
    os.environ["FORA_HIP_PIPELINE"] = str(pipe)
    eng = fora_amd.Engine(0)
    eng.set_graph(n, m, rp, col); eng.set_params(alpha=0.2, epsilon=0.5, seed=1)
    src = synth.query_set(n, 2000, 20261001)
    eng.query(src, want_ppr=False)
    t0 = time.perf_counter()
    for _ in range(3): eng.query(src, want_ppr=False)
    dt = time.perf_counter() - t0
    print("pipeline", pipe, "wall q/s %.0f" % (6000 / dt), "batch", eng.get_batch())
    eng.close()
PY
