#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench15.txt
GB_SPLITKS="0" GB_VARIANTS="23,119,407,663,919,2071,0" timeout 200 tools/microbench/gemm_bench 4096 4096 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench15.txt
CDNA4_TUNE=0 GB_SPLITKS="0" GB_VARIANTS="663,2071" timeout 200 tools/microbench/gemm_bench 4096 4096 512 2>&1 | grep -E "^variant" | sed 's/^/tune0 /' >> gpurun_out/gemm_bench15.txt
GB_SPLITKS="0" GB_VARIANTS="663,2071,1031" timeout 200 tools/microbench/gemm_bench 8192 4096 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench15.txt
GB_SPLITKS="0" GB_VARIANTS="663,2071,1031" timeout 200 tools/microbench/gemm_bench 4096 10752 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench15.txt
GB_SPLITKS="0" GB_VARIANTS="663,2071,1031" timeout 200 tools/microbench/gemm_bench 4096 8192 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench15.txt
GB_ROUNDS=2 GB_SPLITKS="0" GB_VARIANTS="663,2071,1031" timeout 200 tools/microbench/gemm_bench 32768 8192 512 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench15.txt
cat gpurun_out/gemm_bench15.txt
