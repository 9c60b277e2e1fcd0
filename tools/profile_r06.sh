#!/bin/bash
# GPU box, round 6: every profile the round's DESIGN numbers come from -> gpurun_out/r06/ (copied to profiles/r06_* afterwards)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; O=gpurun_out/r06; mkdir -p $O
# 1. headline workload: kernel stats + FETCH_SIZE / WRITE_SIZE (separate passes)
bash tools/profile_bench.sh r06_ws > $O/prof_ws.log 2>&1
cp gpurun_out/prof_r06_ws/kernel_stats.csv $O/ws_kernel_stats.csv; cp gpurun_out/prof_r06_ws/pmc_traffic.json $O/ws_pmc_traffic.json
# 2. SQ / TCC / TCP counters of the push (tools/pmc_push.sh -> gpurun_out/pmc/summary.txt)
bash tools/pmc_push.sh > $O/pmc_push.log 2>&1; cp gpurun_out/pmc/summary.txt $O/pmc_team.txt
# 3. TA / SQ counters of the walk kernel and the push (tools/pmc_ta.sh -> gpurun_out/pmc_ta/summary.txt)
bash tools/pmc_ta.sh > $O/pmc_ta.log 2>&1; cp gpurun_out/pmc_ta/summary.txt $O/pmc_ta.txt
# 4. configs 3 / 4: kernel stats + traffic
bash tools/profile_bench.sh r06_lj --graph livejournal --with-idx > $O/prof_lj.log 2>&1
cp gpurun_out/prof_r06_lj/kernel_stats.csv $O/lj_kernel_stats.csv; cp gpurun_out/prof_r06_lj/pmc_traffic.json $O/lj_pmc_traffic.json
bash tools/profile_bench.sh r06_tw --graph twitter2010 --with-idx --queries 32 > $O/prof_tw.log 2>&1
cp gpurun_out/prof_r06_tw/kernel_stats.csv $O/tw_kernel_stats.csv; cp gpurun_out/prof_r06_tw/pmc_traffic.json $O/tw_pmc_traffic.json
# 5. config 5: kernel stats of one 125-query top-k step
bash tools/kstats_lib.sh r06_tw_topk fora_amd/libfora_hip.so --graph twitter2010 --with-idx --topk 500 --queries 125 > $O/kstats_topk.log 2>&1
cp gpurun_out/kstats_r06_tw_topk.csv $O/tw_topk_kernel_stats.csv
ls -la $O; head -6 $O/ws_kernel_stats.csv | cut -c1-160; grep -A8 -E "k_walk_dg|k_push_team" $O/pmc_ta.txt | head -40
