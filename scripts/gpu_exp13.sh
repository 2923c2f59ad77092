#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gemm_bench13.txt
for t in 0 1 0 1; do
  echo "== CDNA4_TUNE=$t" >> gpurun_out/gemm_bench13.txt
  CDNA4_TUNE=$t GB_SPLITKS="0,1" GB_VARIANTS="2071" timeout 120 tools/microbench/gemm_bench 4096 4096 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench13.txt
  CDNA4_TUNE=$t GB_SPLITKS="0" GB_VARIANTS="2071" timeout 120 tools/microbench/gemm_bench 4096 10752 512 2>&1 | grep -E "^variant" >> gpurun_out/gemm_bench13.txt
done
cat gpurun_out/gemm_bench13.txt
for v in "A=1" "CDNA4_GEMV_NT=1" "A=1" "CDNA4_GEMV_NT=1"; do echo "== $v"; env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); dd=d['decode']; print('fused cold', dd['us_per_step'], 'warm', dd['us_per_step_cache_warm'], 'wall', dd['us_per_step_host_wall'])"; done
