#!/bin/bash
# GPU box, round 5: every profile the round's DESIGN numbers come from -> gpurun_out/r05/ (copied to profiles/r05_* afterwards)
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; O=gpurun_out/r05; mkdir -p $O
# 1. headline workload: kernel stats + FETCH_SIZE / WRITE_SIZE (separate passes)
bash tools/profile_bench.sh r05_ws > $O/prof_ws.log 2>&1
cp gpurun_out/prof_r05_ws/kernel_stats.csv $O/ws_kernel_stats.csv; cp gpurun_out/prof_r05_ws/pmc_traffic.json $O/ws_pmc_traffic.json
# 2. SQ / TCC / TCP counters of the push (tools/pmc_push.sh -> gpurun_out/pmc/summary.txt)
bash tools/pmc_push.sh > $O/pmc_push.log 2>&1; cp gpurun_out/pmc/summary.txt $O/pmc_team.txt
# 3. phase / level stamps and the marginal-cost probes of k_push_team (tools/pushbench.py, 1000 ws-sized queries, push only)
python tools/pushbench.py --reps 5 --mode push variants/libfora_hip_r04.so fora_amd/libfora_hip.so variants/lib_stamps.so variants/lib_levels.so variants/lib_small.so variants/lib_pstore.so variants/lib_pgather.so > $O/team_probes.jsonl 2>&1
# 4. configs 3 / 4: kernel stats + traffic
bash tools/profile_bench.sh r05_lj --graph livejournal --with-idx > $O/prof_lj.log 2>&1
cp gpurun_out/prof_r05_lj/kernel_stats.csv $O/lj_kernel_stats.csv; cp gpurun_out/prof_r05_lj/pmc_traffic.json $O/lj_pmc_traffic.json
bash tools/profile_bench.sh r05_tw --graph twitter2010 --with-idx --queries 32 > $O/prof_tw.log 2>&1
cp gpurun_out/prof_r05_tw/kernel_stats.csv $O/tw_kernel_stats.csv; cp gpurun_out/prof_r05_tw/pmc_traffic.json $O/tw_pmc_traffic.json
# 5. config 5: kernel stats of one 125-query top-k step
bash tools/kstats_lib.sh r05_tw_topk fora_amd/libfora_hip.so --graph twitter2010 --with-idx --topk 500 --queries 125 > $O/kstats_topk.log 2>&1
cp gpurun_out/kstats_r05_tw_topk.csv $O/tw_topk_kernel_stats.csv
ls -la $O; head -6 $O/ws_kernel_stats.csv | cut -c1-160; cat $O/team_probes.jsonl | cut -c1-600
