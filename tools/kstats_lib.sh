#!/bin/bash
# GPU box: per-kernel totals (rocprofv3 --kernel-trace --stats) of one bench.py step with a given library build.
# usage: tools/kstats_lib.sh <tag> <lib.so> <bench.py args...>
TAG="$1"; LIB="$2"; shift 2
R="${GRAFT_REPO_ROOT:-/root/repo}"
export FORA_HIP_LIB="$(realpath $LIB)"
OUT="$R/gpurun_out/kstats_tmp_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --no-accuracy --no-variants --no-configs "$@" > "$OUT/kt.log" 2>&1
cp $(find "$OUT/kt" -name '*kernel_stats.csv' | head -1) "$R/gpurun_out/kstats_$TAG.csv" 2>/dev/null
rm -rf "$OUT"
python3 - "$R/gpurun_out/kstats_$TAG.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-58s calls %5s total %9.1f ms avg %9.3f ms" % (r['Name'][:58], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6))
PY
