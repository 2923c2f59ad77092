#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench14.txt
GB_TRACE_SPLITK=1 GB_SPLITKS="0" GB_VARIANTS="2071" timeout 120 tools/microbench/gemm_bench 4096 4096 512 2071 > gpurun_out/gemm_bench14.txt 2>&1
cat gpurun_out/gemm_bench14.txt
