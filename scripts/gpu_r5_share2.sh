#!/bin/bash
# round 5: the NORM chain leaves the activation image (no quantizer launch for any MUL_MAT of its rows) — tests, then the layer front with the hand-off on / off on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt gpurun_out/split_report.jsonl
timeout -k 10 400 python -m pytest tests/test_gpu_act_share.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_act_share.log 2>&1
echo "pytest act_share rc=$?" >> gpurun_out/summary.txt; tail -5 gpurun_out/pytest_act_share.log >> gpurun_out/summary.txt
H=oracle/_ref/split_harness; P=ggml_amd/lib/libggml-cdna4.so
: > gpurun_out/act_share_ab2.txt
for rep in 1 2; do
  for shape in "q4_K 4096 14336 512" "q4_K 4096 14336 64"; do
    HARNESS_NO_CPU=1 timeout 200 $H $P $shape shared >> gpurun_out/act_share_ab2.txt 2>> gpurun_out/act_share_ab2.err
    HARNESS_NO_CPU=1 GGML_CDNA4_NO_ACT_SHARE=1 timeout 200 $H $P $shape shared | sed 's/^{/{"share":"off",/' >> gpurun_out/act_share_ab2.txt 2>> gpurun_out/act_share_ab2.err
  done
done
cat gpurun_out/summary.txt; cut -c1-230 gpurun_out/act_share_ab2.txt
