#!/bin/bash
# round 5: Q6_K on k_gemm_r8 through a resident Q6_K8 image — its tests, then the A/B against the per-call route on the same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt
timeout -k 10 300 python -m pytest tests/test_gpu_resident.py -m gpu -q --tb=short -p no:cacheprovider -k "q8_0 or q4_0" > gpurun_out/pytest_q6k8.log 2>&1
echo "pytest q6_K/q8_0/q4_0 resident rc=$?" >> gpurun_out/summary.txt; tail -4 gpurun_out/pytest_q6k8.log >> gpurun_out/summary.txt
AB_TYPE=14 timeout -k 10 250 python scripts/relayout_resident_ab.py 32768x8192x512 8192x8192x2048 16384x4096x1024 16384x8192x512 14336x4096x512 8192x8192x512 4096x14336x512 8192x4096x512 4096x4096x512 > gpurun_out/q6k_resident_ab.txt 2> gpurun_out/q6k_resident_ab.err; echo "ab rc=$?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt; cut -c1-330 gpurun_out/q6k_resident_ab.txt; tail -5 gpurun_out/q6k_resident_ab.err
