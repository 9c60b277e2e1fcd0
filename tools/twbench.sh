#!/bin/bash
# Experiment tooling (GPU box): Twitter-2010-sized indexed queries with one or more library builds.
# usage: tools/twbench.sh <queries> lib1.so [lib2.so ...]
Q="$1"; shift
for L in "$@"; do
  FORA_HIP_LIB=$(realpath $L) python bench.py --graph twitter2010 --with-idx --steps 1 --warmup 1 --queries $Q --no-cpu --no-accuracy 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases']; print('$L', round(d['value'],2), d['config']['batch'], {k: round(v,1) for k,v in p.items() if k.endswith('_ms')}, d['roofline']['launches'])"
done
