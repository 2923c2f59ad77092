#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench20.txt
for t in 0 8 4 8; do
  echo "== CDNA4_TUNE=$t" >> gpurun_out/gemm_bench20.txt
  CDNA4_TUNE=$t GB_SPLITKS="0,1" GB_VARIANTS="2071" timeout 200 tools/microbench/gemm_bench 4096 4096 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench20.txt
  CDNA4_TUNE=$t GB_SPLITKS="1" GB_VARIANTS="2071" timeout 200 tools/microbench/gemm_bench 8192 8192 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench20.txt
done
cat gpurun_out/gemm_bench20.txt
