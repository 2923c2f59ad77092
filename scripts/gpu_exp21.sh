#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench21.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm_variants and 4119" > gpurun_out/pytest_exp.log 2>&1
echo "parity rc=$?"; tail -6 gpurun_out/pytest_exp.log
GB_SPLITKS="0,1" GB_VARIANTS="2071,4119" timeout 200 tools/microbench/gemm_bench 4096 4096 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench21.txt
GB_SPLITKS="0" GB_VARIANTS="2071,4119" timeout 200 tools/microbench/gemm_bench 8192 4096 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench21.txt
GB_SPLITKS="0" GB_VARIANTS="2071,4119" timeout 200 tools/microbench/gemm_bench 4096 10752 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench21.txt
GB_ROUNDS=2 GB_SPLITKS="0" GB_VARIANTS="2071,4119,1031" timeout 200 tools/microbench/gemm_bench 32768 8192 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench21.txt
cat gpurun_out/gemm_bench21.txt
