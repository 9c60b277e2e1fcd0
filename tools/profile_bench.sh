#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes (separate runs) of ONE bench.py batch.
# usage: tools/profile_bench.sh <tag> <bench.py args...>   -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_traffic.json}
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu --no-accuracy --no-variants --no-configs $*"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- python3 "$R/bench.py" $ARGS > "$OUT/write.log" 2>&1
cp $(find "$OUT/kt" -name '*kernel_stats.csv' | head -1) "$OUT/kernel_stats.csv" 2>/dev/null
python3 "$R/tools/pmc_summary.py" "$OUT/pmc_traffic.json" $(find "$OUT/fetch" -name '*counter_collection.csv') $(find "$OUT/write" -name '*counter_collection.csv') > "$OUT/pmc_summary.txt" 2>&1
head -12 "$OUT/kernel_stats.csv"; grep -E "pushq_bin|k_accum|k_walk" "$OUT/pmc_summary.txt" | head
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write"
