timeout 900 python3 -m pytest tests/test_hip_parity_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu -k "walk or query or topk or index" 2>&1 | tail -2
python3 tools/pushbench.py --reps 3 --mode query | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('walk_ms','walk_alloc_ms','walk_accum_ms','push_ms','batch_ms','walks_per_q')})"
