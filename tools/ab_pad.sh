#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
FORA_HIP_LIB=$R/variants/lib_pad4.so timeout 1500 python3 -m pytest tests/test_hip_parity_gpu.py tests/test_large_gpu.py -x -q -m gpu -k "wide or medium or push_paths or deferral" > gpurun_out/ab_tests.log 2>&1; echo "tests(pad4) rc=$?"; tail -4 gpurun_out/ab_tests.log
LIBS="fora_amd/libfora_hip.so variants/lib_pad2.so variants/lib_pad4.so variants/lib_pad8.so"
echo "== LJ idx"; timeout 900 python3 tools/pushbench.py --graph livejournal --mode idx --queries 280 --reps 2 $LIBS | cut -c1-330
echo "== TW idx"; timeout 1500 python3 tools/pushbench.py --graph twitter2010 --mode idx --queries 32 --reps 1 $LIBS | cut -c1-330
