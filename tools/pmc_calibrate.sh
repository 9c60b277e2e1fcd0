#!/bin/bash
# GPU box: calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE on known byte counts -> gpurun_out/pmc_calibration.txt
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/calib"; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$OUT/run_bench" "$R/tools/run_bench.hip" || exit 1
cd /tmp && export TMPDIR=/tmp
"$OUT/run_bench" calib > "$OUT/known.txt"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/f" -o f --output-format csv -- "$OUT/run_bench" calib > "$OUT/f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/w" -o w --output-format csv -- "$OUT/run_bench" calib > "$OUT/w.log" 2>&1
python3 "$R/tools/pmc_calibrate.py" "$OUT/known.txt" $(find "$OUT/f" -name '*counter_collection.csv') $(find "$OUT/w" -name '*counter_collection.csv') > "$R/gpurun_out/pmc_calibration.txt" 2>&1
cat "$R/gpurun_out/pmc_calibration.txt"
rm -rf "$OUT/f" "$OUT/w"
