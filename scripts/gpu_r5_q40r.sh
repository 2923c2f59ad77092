#!/bin/bash
# round 5: Q4_0 on Q4_K's kernels through a resident Q4_0R image — its tests on the GPU, then the A/B against the per-call route on the same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt
timeout -k 10 400 python -m pytest tests/test_gpu_resident.py -m gpu -q --tb=short -p no:cacheprovider -k "q4_0 or q4_K or iq4_xs" > gpurun_out/pytest_q40r.log 2>&1
echo "pytest q4_0 resident rc=$?" >> gpurun_out/summary.txt; tail -4 gpurun_out/pytest_q40r.log >> gpurun_out/summary.txt
timeout -k 10 300 python scripts/relayout_resident_ab.py > gpurun_out/q40_resident_ab.txt 2> gpurun_out/q40_resident_ab.err; echo "ab rc=$?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt; cat gpurun_out/q40_resident_ab.txt; tail -5 gpurun_out/q40_resident_ab.err
