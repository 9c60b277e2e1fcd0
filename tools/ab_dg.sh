#!/bin/bash
# GPU box: A/B of k_walk_dg variants (options and builds) on the headline workload
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" python3 tools/pushbench.py --mode query --reps 3 $LIBS | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('%-16s walk %.1f walk_accum %.1f alloc %.1f push %.1f batch %.1f' % (d['lib'], d['walk_ms'], d['walk_accum_ms'], d['walk_alloc_ms'], d['push_ms'], d['batch_ms']))
"; }
LIBS="fora_amd/libfora_hip.so variants/lib_tile512.so variants/lib_tile128.so"
run FORA_HIP_WALK_DG=2
LIBS=""
run FORA_HIP_DG_HUBS=512
run FORA_HIP_XB=12
run FORA_HIP_XB=24
