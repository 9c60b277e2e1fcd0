#!/bin/bash
# Experiment tooling (GPU box): SQ counter passes over one push of 1000 ws queries; summaries under gpurun_out/pmc/.
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/pmc"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
P2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
P4="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $P -d "$OUT/p$i" -o p --output-format csv -- python3 "$R/tools/pushbench.py" --child --reps 1 "$@" > "$OUT/p$i.log" 2>&1
done
python3 "$R/tools/pmc_generic.py" $(find "$OUT" -name '*counter_collection.csv') > "$OUT/summary.txt" 2>&1
grep -A40 -E "k_pushq_bin|k_accum<false>|k_push_team|k_push_tail" "$OUT/summary.txt" | head -120
