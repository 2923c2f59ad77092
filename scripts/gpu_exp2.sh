#!/bin/bash
# uneven hand-off split sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench2.txt
for n in 8 9 10 11; do
  echo "== CDNA4_SPLIT_NUM=$n" >> gpurun_out/gemm_bench2.txt
  CDNA4_SPLIT_NUM=$n GB_VARIANTS="23,407,663" timeout 120 tools/microbench/gemm_bench 4096 4096 512 2>&1 | grep -E "^variant .* splitk 2" >> gpurun_out/gemm_bench2.txt
done
echo "== K=11008->10752 (C3), default split" >> gpurun_out/gemm_bench2.txt
GB_VARIANTS="23,407" timeout 120 tools/microbench/gemm_bench 4096 10752 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench2.txt
cat gpurun_out/gemm_bench2.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gemm or prefill or golden" > gpurun_out/pytest_exp.log 2>&1
echo "parity rc=$?"; tail -5 gpurun_out/pytest_exp.log
