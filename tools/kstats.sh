#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of ONE bench.py step.  usage: tools/kstats.sh <tag> <bench.py args...> -> gpurun_out/kstats_<tag>.csv
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/kstats_tmp_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --no-accuracy --no-variants "$@" > "$OUT/kt.log" 2>&1
cp $(find "$OUT/kt" -name '*kernel_stats.csv' | head -1) "$R/gpurun_out/kstats_$TAG.csv" 2>/dev/null
rm -rf "$OUT"
cut -d, -f1-4,7 "$R/gpurun_out/kstats_$TAG.csv" | head -24
