timeout 600 python -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "team_push" 2>&1 | tail -5
echo "== stamps build (Mcyc): consume sweep pop chunks wait+heavy+drain+prebar barrier final slotacq"
timeout 120 python tools/pushbench.py --reps 3 variants/lib_stamps.so
for t in 1024 4096; do echo "== team_tail $t"; FORA_HIP_TEAM_TAIL=$t timeout 120 python tools/pushbench.py --reps 3; done
echo "== ept4"; timeout 120 python tools/pushbench.py --reps 3 variants/lib_ept4.so
echo "== cu4"; timeout 120 python tools/pushbench.py --reps 3 variants/lib_cu4.so
