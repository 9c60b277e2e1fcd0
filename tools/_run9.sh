for t in 2048 3072 4096 5120 6144 8192; do python3 tools/pushbench.py --reps 3 --option team_tail=$t | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print($t, {k:d[k] for k in ('tail_ms','team_ms','push_ms')})"; done
