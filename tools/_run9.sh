timeout 900 python3 -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "team or push_bit" 2>&1 | tail -2
python3 tools/pushbench.py --reps 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('tail_ms','team_ms','push_ms')})"
