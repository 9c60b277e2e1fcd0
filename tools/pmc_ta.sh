#!/bin/bash
# GPU box: texture-addresser / L1 counters of the headline step's two hot kernels (one batch of 1000 ws-sized queries) -> gpurun_out/pmc_ta/summary.txt
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/pmc_ta"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TA_[A-Z0-9_]+|TCP_[A-Z0-9_]+|GRBM_GUI_ACTIVE|GRBM_COUNT)\b" | sort -u > "$OUT/avail.txt"
P1="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr"
# (TA_FLAT_*_WAVEFRONTS / TA_ADDR_STALLED_* passes abort inside rocprofv3 on this image and hang: not collected)
i=0
for P in "$P1"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P -d "$OUT/p$i" -o p --output-format csv -- python3 "$R/tools/pushbench.py" --child --reps 1 --mode query > "$OUT/p$i.log" 2>&1
done
python3 "$R/tools/pmc_generic.py" $(find "$OUT" -name '*counter_collection.csv') > "$OUT/summary.txt" 2>&1
grep -A14 -E "k_push_team|k_walk_dg" "$OUT/summary.txt" | head -60; wc -l "$OUT/avail.txt"; grep -c . "$OUT/p1.log"
