timeout 1200 python3 -m pytest tests/test_hip_parity_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -2
python3 tools/pushbench.py --reps 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('tail_ms','team_ms','push_ms')})"
python3 tools/pushbench.py --reps 3 --option tail_hubs=0 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('tail_hubs=0', {k:d[k] for k in ('tail_ms','team_ms','push_ms')})"
