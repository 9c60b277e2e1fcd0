#!/bin/bash
# A/B of the threshold-round schedules (options rounds / round_div) within one run: tools/roundbench.sh <graph> <mode> <queries> "R:D R:D ..."
G=${1:-webstanford}; M=${2:-query}; Q=${3:-1000}; shift 3
for rd in ${@:-1:0 2:4}; do
  R=${rd%%:*}; D=${rd##*:}
  echo -n "rounds=$R div=$D "
  python tools/pushbench.py --graph $G --mode $M --queries $Q --reps 2 --option rounds=$R --option round_div=$D fora_amd/libfora_hip.so | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: d[k] for k in ('bin_ms', 'accum_ms', 'tail_ms', 'push_ms', 'launches', 'walk_ms', 'batch_ms', 'relax_per_q', 'walks_per_q', 'other_ms')})"
done
