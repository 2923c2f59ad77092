#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench17.txt
(rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk|fclk" | head -4) >> gpurun_out/gemm_bench17.txt
GB_TRACE_SPLITK=2 GB_ROUNDS=1 GB_SPLITKS="0" GB_VARIANTS="2071" timeout 120 tools/microbench/gemm_bench 4096 4096 512 2071 2>&1 | grep -E "block 0|^variant" >> gpurun_out/gemm_bench17.txt
GB_TRACE_SPLITK=1 GB_ROUNDS=1 GB_SPLITKS="1" GB_VARIANTS="2071" timeout 120 tools/microbench/gemm_bench 4096 4096 512 2071 2>&1 | grep -E "block 0|^variant" >> gpurun_out/gemm_bench17.txt
GB_TRACE_SPLITK=1 GB_ROUNDS=1 GB_SPLITKS="1" GB_VARIANTS="2071" timeout 120 tools/microbench/gemm_bench 8192 8192 512 2071 2>&1 | grep -E "block 0|^variant" >> gpurun_out/gemm_bench17.txt
cat gpurun_out/gemm_bench17.txt
timeout 300 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-900
