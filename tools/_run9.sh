python3 tools/pushbench.py --reps 2 --graph livejournal --queries 143 variants/lib_stamps.so | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l[:500]); continue
    print({k:d.get(k) for k in ('bin_ms','accum_ms','tail_ms','push_ms','launches','stamps_bin_Mcyc','stamps_acc_Mcyc','relax_per_q','pops_per_q')})"
