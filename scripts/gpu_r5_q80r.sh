#!/bin/bash
# round 5: Q8_0 on k_gemm_r8 through a resident Q8_0R image — its tests, then the A/B against the per-call route on the same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt
timeout -k 10 300 python -m pytest tests/test_gpu_resident.py -m gpu -q --tb=short -p no:cacheprovider -k "q8_0 or q4_0" > gpurun_out/pytest_q80r.log 2>&1
echo "pytest q8_0 resident rc=$?" >> gpurun_out/summary.txt; tail -4 gpurun_out/pytest_q80r.log >> gpurun_out/summary.txt
AB_TYPE=8 timeout -k 10 200 python scripts/relayout_resident_ab.py 8192x8192x2048 16384x4096x1024 32768x8192x512 8192x4096x4096 4096x4096x512 > gpurun_out/q80_resident_ab.txt 2> gpurun_out/q80_resident_ab.err; echo "ab rc=$?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt; cat gpurun_out/q80_resident_ab.txt; tail -5 gpurun_out/q80_resident_ab.err
