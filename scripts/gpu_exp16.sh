#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench16.txt
for K in 2048 4096 8192; do
GB_SPLITKS="1,2" GB_VARIANTS="2071" timeout 200 tools/microbench/gemm_bench 4096 $K 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench16.txt
done
GB_SPLITKS="1" GB_VARIANTS="2071" timeout 200 tools/microbench/gemm_bench 8192 2048 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench16.txt
GB_SPLITKS="1" GB_VARIANTS="2071" timeout 200 tools/microbench/gemm_bench 8192 8192 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench16.txt
cat gpurun_out/gemm_bench16.txt
