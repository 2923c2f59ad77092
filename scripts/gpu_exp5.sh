#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench5.txt
GB_TRACE_SPLITK=2 GB_VARIANTS="407,663,919" timeout 120 tools/microbench/gemm_bench 4096 4096 512 663 2>&1 | grep -vE "^wave|^trace" >> gpurun_out/gemm_bench5.txt
GB_VARIANTS="407,663,919" timeout 120 tools/microbench/gemm_bench 8192 4096 512 2>&1 | grep -E "^M=|^variant .* splitk 1" >> gpurun_out/gemm_bench5.txt
GB_VARIANTS="407,663,919" timeout 120 tools/microbench/gemm_bench 4096 10752 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench5.txt
GB_VARIANTS="407,663" timeout 120 tools/microbench/gemm_bench 32768 8192 512 2>&1 | grep -E "^M=|^variant .* splitk 1" >> gpurun_out/gemm_bench5.txt
cat gpurun_out/gemm_bench5.txt
