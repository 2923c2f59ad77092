#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt
timeout 120 tools/microbench/l2_stream > gpurun_out/l2_stream.csv 2>&1; echo "micro rc=$?" >> gpurun_out/summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_gpt2.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpt2.log 2>&1
echo "gpt2 rc=$?" >> gpurun_out/summary.txt; tail -6 gpurun_out/pytest_gpt2.log >> gpurun_out/summary.txt
cat gpurun_out/l2_stream.csv; cat gpurun_out/summary.txt; cat gpurun_out/gpt2_parity.jsonl | tail -5
