#!/bin/bash
# GPU box: HBM read bytes (FETCH_SIZE) and L2 hits / misses of the LJ-sized step's kernels in round 5's dispatch order (FORA_HIP_SLOT_MAJOR=0) and in the
# slot-major one (5: bin kernel + indexed walks) -> gpurun_out/pmc_sm/summary.txt.  One batch of 143 slots per pass; counters in passes of their own.
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/pmc_sm"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--graph livejournal --with-idx --queries 143 --steps 1 --warmup 0 --no-cpu --no-accuracy --no-variants --no-configs"
for SM in 0 5; do
  export FORA_HIP_SLOT_MAJOR=$SM
  rocprofv3 --pmc FETCH_SIZE -d "$OUT/f$SM" -o f --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/f$SM.log" 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/t$SM" -o t --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/t$SM.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE -d "$OUT/w$SM" -o w --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/w$SM.log" 2>&1
  rocprofv3 --kernel-trace --stats -d "$OUT/k$SM" -o k --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/k$SM.log" 2>&1; cp $(find "$OUT/k$SM" -name "*kernel_stats.csv" | head -1) "$OUT/kstats$SM.csv"
  python3 "$R/tools/pmc_generic.py" $(find "$OUT/f$SM" "$OUT/t$SM" "$OUT/w$SM" -name '*counter_collection.csv') > "$OUT/sm$SM.txt" 2>&1
done
{ for SM in 0 5; do echo "== FORA_HIP_SLOT_MAJOR=$SM"; grep -A5 -E "k_pushq_bin|k_accum<false, true>|k_walk_idx|k_accum<true, true>" "$OUT/sm$SM.txt"; done; } > "$OUT/summary.txt"
cat "$OUT/summary.txt"
for SM in 0 5; do echo "== kernel stats FORA_HIP_SLOT_MAJOR=$SM"; head -6 "$OUT/kstats$SM.csv" | cut -c1-140; done | tee -a "$OUT/summary.txt"
rm -rf "$OUT"/f0 "$OUT"/f5 "$OUT"/t0 "$OUT"/t5 "$OUT"/w0 "$OUT"/w5 "$OUT"/k0 "$OUT"/k5
