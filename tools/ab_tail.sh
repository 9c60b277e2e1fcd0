#!/bin/bash
# GPU box: A/B of the frontier size from which k_push_tail takes over, and of its launch shape (headline workload)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" python3 tools/pushbench.py --mode push --reps 3 $LIBS | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('%-16s bin %.1f acc %.1f tail %.1f push %.1f | launches %.0f' % (d['lib'], d['bin_ms'], d['accum_ms'], d['tail_ms'], d['push_ms'], d['launches']))
"; }
LIBS=""
for T in 32768 40000 50000 65536 80000; do run FORA_HIP_TAIL=$T; done
LIBS="variants/lib_tept8.so variants/lib_tept2.so variants/lib_tt512.so"
run FORA_HIP_TAIL=32768
run FORA_HIP_TAIL=50000
