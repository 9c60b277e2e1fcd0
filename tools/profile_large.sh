#!/bin/bash
# GPU box: kernel stats + PMC traffic of the LJ- and Twitter-2010-sized workloads (tools/profile_bench.sh) -> gpurun_out/prof_r04_{lj,tw}/
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
bash tools/profile_bench.sh r05_lj --graph livejournal --with-idx > gpurun_out/prof_r05_lj.log 2>&1
bash tools/profile_bench.sh r05_tw --graph twitter2010 --with-idx --queries 32 > gpurun_out/prof_r05_tw.log 2>&1
for t in lj tw; do echo "== $t"; head -7 gpurun_out/prof_r05_$t/kernel_stats.csv | cut -c1-140; grep -E "pushq_bin|k_accum|k_walk_idx" gpurun_out/prof_r05_$t/pmc_summary.txt | cut -c1-200; done
