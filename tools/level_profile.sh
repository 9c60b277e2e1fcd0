#!/bin/bash
# GPU box: per-dispatch durations of the push kernels for one 1000-query ws push (kernel trace), in launch order.
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/levels"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/kt" -o kt --output-format csv -- python3 "$R/tools/pushbench.py" --child --reps 1 "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("void fora::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "fora::" in r["Kernel_Name"]]
# second half = the timed repetition
names = [n for n, _ in seq]
half = len(seq) // 2
seq = seq[half:]
with open(out + "/levels.txt", "w") as fo:
    for n, us in seq:
        fo.write(f"{n}\t{us:.1f}\n")
bins = [us for n, us in seq if n.startswith("k_pushq_bin")]
accs = [us for n, us in seq if n.startswith("k_accum<false>")]
print("bin us:", [round(x) for x in bins])
print("acc us:", [round(x) for x in accs])
PY
rm -rf "$OUT/kt"
