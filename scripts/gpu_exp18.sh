#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench18.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm or golden or full_size or repacked" > gpurun_out/pytest_exp.log 2>&1
echo "parity rc=$?"; tail -4 gpurun_out/pytest_exp.log
for l2 in 0 1 0 1; do
  echo "== CDNA4_XCHG_L2=$l2" >> gpurun_out/gemm_bench18.txt
  CDNA4_XCHG_L2=$l2 GB_SPLITKS="0" GB_VARIANTS="2071,663" timeout 200 tools/microbench/gemm_bench 4096 4096 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench18.txt
done
CDNA4_XCHG_L2=1 GB_SPLITKS="0" GB_VARIANTS="2071" timeout 200 tools/microbench/gemm_bench 4096 10752 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench18.txt
CDNA4_XCHG_L2=0 GB_SPLITKS="0" GB_VARIANTS="2071" timeout 200 tools/microbench/gemm_bench 4096 10752 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench18.txt
cat gpurun_out/gemm_bench18.txt
