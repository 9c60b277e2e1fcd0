python3 tools/pushbench.py --reps 2 variants/lib_sbase.so variants/lib_frb.so variants/lib_fhub.so variants/lib_ffill.so variants/lib_fstore.so | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print({k:d.get(k) for k in ('lib','team_ms','stamps_bin_Mcyc')})"
