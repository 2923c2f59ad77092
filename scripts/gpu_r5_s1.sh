#!/bin/bash
# round 5, session 1: the one-launch step — parity tests first, then the A/B (two launches | one launch, hand-off | one launch, ticketed), alternating on this box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out/s1
timeout -k 10 900 python -m pytest tests/test_gpu_fused_step.py -q --tb=short -p no:cacheprovider -x > gpurun_out/s1/pytest_fused.log 2>&1
echo "pytest fused rc=$?" > gpurun_out/s1/summary.txt; tail -15 gpurun_out/s1/pytest_fused.log >> gpurun_out/s1/summary.txt
for rep in 1 2; do
  AB_TAG=two_launches CDNA4_NO_FUSEQ=1 timeout 300 python scripts/step_ab.py >> gpurun_out/s1/step_ab.txt 2>> gpurun_out/s1/step_ab.err
  AB_TAG=one_launch_handoff timeout 300 python scripts/step_ab.py >> gpurun_out/s1/step_ab.txt 2>> gpurun_out/s1/step_ab.err
  AB_TAG=one_launch_ticketed CDNA4_FQ_TICKETED=1 timeout 300 python scripts/step_ab.py >> gpurun_out/s1/step_ab.txt 2>> gpurun_out/s1/step_ab.err
done
timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > gpurun_out/s1/bench.log 2> gpurun_out/s1/bench.err; echo "bench rc=$?" >> gpurun_out/s1/summary.txt
cat gpurun_out/s1/summary.txt; cat gpurun_out/s1/step_ab.txt; tail -c 1500 gpurun_out/s1/bench.log; tail -5 gpurun_out/s1/step_ab.err
