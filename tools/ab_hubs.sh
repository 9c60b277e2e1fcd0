#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
FORA_HIP_HUB_MIN=16 timeout 1500 python3 -m pytest tests/test_hip_parity_gpu.py tests/test_edge_cases_gpu.py tests/test_golden.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1; echo "tests(hub_min=16) rc=$?"; tail -5 gpurun_out/ab_tests.log
run() { echo "== $*"; env "$@" python3 tools/pushbench.py --mode push --reps 3 $ARGS $LIBS | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print('%-16s bin %.1f acc %.1f tail %.1f push %.1f | batch %.1f | launches %.0f' % (d['lib'], d['bin_ms'], d['accum_ms'], d['tail_ms'], d['push_ms'], d['batch_ms'], d['launches']))
"; }
LIBS=""; ARGS=""
run FORA_HIP_HUBS=0
run FORA_HIP_HUBS=1024
run FORA_HIP_HUBS=2048
run FORA_HIP_HUBS=4096
run FORA_HIP_HUBS=4096 FORA_HIP_HUB_MIN=16384
run FORA_HIP_HUBS=4096 FORA_HIP_HUB_MIN=1024
run FORA_HIP_HUBS=6144
