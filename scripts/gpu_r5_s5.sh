#!/bin/bash
# round 5, session 5: grouped MUL_MAT_ID with the plan's tile order + fragment counts (128-row tiles) vs round 4's launch (CDNA4_MOE_PLAN=0: 256-row tiles), parity tests first
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out/s5; rm -f gpurun_out/s5/*
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -k "mul_mat_id" > gpurun_out/s5/pytest_moe.log 2>&1
echo "pytest moe rc=$?" > gpurun_out/s5/summary.txt; tail -4 gpurun_out/s5/pytest_moe.log >> gpurun_out/s5/summary.txt
for rep in 1 2; do
  AB_TAG=plan_tm128 timeout 120 python scripts/moe_ab.py >> gpurun_out/s5/moe_ab.txt 2>> gpurun_out/s5/err.txt
  AB_TAG=r4_tm256 CDNA4_MOE_PLAN=0 timeout 120 python scripts/moe_ab.py >> gpurun_out/s5/moe_ab.txt 2>> gpurun_out/s5/err.txt
  AB_TAG=plan_off_tm128 CDNA4_MOE_PLAN=0 CDNA4_MOE_TM=128 timeout 120 python scripts/moe_ab.py >> gpurun_out/s5/moe_ab.txt 2>> gpurun_out/s5/err.txt
done
cat gpurun_out/s5/summary.txt gpurun_out/s5/moe_ab.txt; tail -3 gpurun_out/s5/err.txt
