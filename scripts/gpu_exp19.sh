#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench19.txt
for t in 2 0 2 0; do
  echo "== CDNA4_TUNE=$t" >> gpurun_out/gemm_bench19.txt
  CDNA4_TUNE=$t GB_SPLITKS="0,1" GB_VARIANTS="2071" timeout 200 tools/microbench/gemm_bench 4096 4096 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench19.txt
  CDNA4_TUNE=$t GB_SPLITKS="0" GB_VARIANTS="2071" timeout 200 tools/microbench/gemm_bench 8192 4096 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench19.txt
done
cat gpurun_out/gemm_bench19.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm_parity_auto or golden or full_size" > gpurun_out/pytest_exp.log 2>&1
echo "parity rc=$?"; tail -3 gpurun_out/pytest_exp.log
