#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench.txt
timeout 120 tools/microbench/gemm_bench 4096 4096 512 55 >> gpurun_out/gemm_bench.txt 2>&1
timeout 120 tools/microbench/gemm_bench 8192 4096 512 2>&1 | grep -E "variant (23|55) splitk 1" >> gpurun_out/gemm_bench.txt
cat gpurun_out/gemm_bench.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm_variants or full_size_prefill or golden" > gpurun_out/pytest_variants.log 2>&1
echo "parity rc=$?"; tail -4 gpurun_out/pytest_variants.log
