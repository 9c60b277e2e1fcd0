#!/bin/bash
# GPU box: kernel stats + PMC traffic of the three workloads with the current kernels
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
bash tools/profile_bench.sh r03_ws > gpurun_out/prof_r03_ws.log 2>&1
bash tools/profile_bench.sh r03_lj --graph livejournal --with-idx > gpurun_out/prof_r03_lj.log 2>&1
bash tools/profile_bench.sh r03_tw --graph twitter2010 --with-idx --queries 32 > gpurun_out/prof_r03_tw.log 2>&1
for t in ws lj tw; do echo "== $t"; head -8 gpurun_out/prof_r03_$t/kernel_stats.csv | cut -c1-160; cat gpurun_out/prof_r03_$t/pmc_summary.txt | cut -c1-250; done
