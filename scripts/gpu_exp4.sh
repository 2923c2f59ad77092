#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench4.txt
for n in 8 9; do
  echo "== CDNA4_SPLIT_NUM=$n" >> gpurun_out/gemm_bench4.txt
  CDNA4_SPLIT_NUM=$n GB_TRACE_SPLITK=2 GB_VARIANTS="407,663" timeout 120 tools/microbench/gemm_bench 4096 4096 512 407 2>&1 | grep -E "splitk 2|consumer|producer" >> gpurun_out/gemm_bench4.txt
done
cat gpurun_out/gemm_bench4.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm or prefill or golden" > gpurun_out/pytest_exp.log 2>&1
echo "parity rc=$?"; tail -3 gpurun_out/pytest_exp.log
