#!/bin/bash
# round 5: hand-off of quantized activations between MUL_MATs of one src1 — tests, then the layer graph at llama-8B sizes with the hand-off on / off, alternating on one box;
# then every -m gpu test, smoke and the driver's bench command on this tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt gpurun_out/split_report.jsonl gpurun_out/parity_report.jsonl
timeout -k 10 600 python -m pytest tests/test_gpu_act_share.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_act_share.log 2>&1
echo "pytest act_share rc=$?" >> gpurun_out/summary.txt; tail -4 gpurun_out/pytest_act_share.log >> gpurun_out/summary.txt
H=oracle/_ref/split_harness; P=ggml_amd/lib/libggml-cdna4.so
: > gpurun_out/act_share_ab.txt
for rep in 1 2; do
  for shape in "q4_K 4096 14336 512" "q4_K 4096 14336 64" "q4_0 4096 11008 512"; do
    HARNESS_NO_CPU=1 timeout 200 $H $P $shape shared >> gpurun_out/act_share_ab.txt 2>> gpurun_out/act_share_ab.err
    HARNESS_NO_CPU=1 GGML_CDNA4_NO_ACT_SHARE=1 timeout 200 $H $P $shape shared | sed 's/^{/{"share":"off",/' >> gpurun_out/act_share_ab.txt 2>> gpurun_out/act_share_ab.err
  done
done
echo "ab done" >> gpurun_out/summary.txt
timeout -k 10 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" >> gpurun_out/summary.txt; tail -6 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_driver_cmd.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt; cat gpurun_out/act_share_ab.txt; head -c 700 gpurun_out/bench_driver_cmd.log; echo; grep -i 'fail\|error' gpurun_out/pytest_gpu.log | head -20
