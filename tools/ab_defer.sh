#!/bin/bash
# GPU box: deferral tests, then A/B of the bounded deferral (options defer / defer_min) on the headline workload.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "deferral or push_paths or query_bit" > gpurun_out/ab_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/ab_tests.log
run() { echo "== $*"; python3 tools/pushbench.py --mode query --reps 3 $(for kv in "$@"; do echo --option $kv; done) $LIBS | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('%-16s bin %.1f acc %.1f tail %.1f push %.1f | walk %.1f wacc %.1f alloc %.1f | batch %.1f | launches %.0f relax/q %.0f pops/q %.0f walks/q %.0f' % (d['lib'], d['bin_ms'], d['accum_ms'], d['tail_ms'], d['push_ms'], d['walk_ms'], d['walk_accum_ms'], d['walk_alloc_ms'], d['batch_ms'], d['launches'], d['relax_per_q'], d['pops_per_q'], d['walks_per_q']))
"; }
LIBS=""
run defer=0
run defer=1 defer_min=8192
run defer=1 defer_min=16384
run defer=1 defer_min=32768
run defer=1 defer_min=50000
run defer=2 defer_min=32768
