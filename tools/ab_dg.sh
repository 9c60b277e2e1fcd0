#!/bin/bash
# GPU box: parity tests, then A/B of k_walk_dg variants (options and builds) on the headline workload.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_hip_parity_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/ab_tests.log
FORA_HIP_WALK_DG=1 timeout 1500 python3 -m pytest tests/test_hip_parity_gpu.py -x -q -m gpu -k "query or walk or topk or batching" > gpurun_out/ab_tests1.log 2>&1; echo "tests(dg=1) rc=$?"; tail -2 gpurun_out/ab_tests1.log
