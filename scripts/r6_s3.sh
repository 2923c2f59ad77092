#!/bin/bash
# round 6, session 3: stream-k MUL_MAT_ID with the faster planner — per-kernel durations, the per-segment trace (instrumented twin), cost weights; the shared-device tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/r6s3; mkdir -p $O; rm -rf $O/*
export GGML_CDNA4_OWNED_DEVICE=1
timeout 120 python scripts/moe_trace.py > $O/moe_trace.txt 2> $O/moe_trace.err
CDNA4_SK_CW=10,10,10,10 timeout 120 python scripts/moe_trace.py > $O/moe_trace_flat.txt 2>> $O/moe_trace.err
for rep in 1 2; do
  AB_TAG=sk_default timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk_cw_flat CDNA4_SK_CW=10,10,10,10 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk_cw_9_10_10_10 CDNA4_SK_CW=9,10,10,10 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk_cw_10_10_10_9 CDNA4_SK_CW=10,10,10,9 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=per_tile CDNA4_MOE_SK=0 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
done
cd /tmp
CDNA4_SK_CW=10,10,10,10 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/moe_prof -o moe -- python $R/scripts/moe_ab.py > $O/moe_prof.txt 2>&1
cd $R
python - <<'PY' > $O/moe_kernels.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/r6s3/moe_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $O/moe_prof
timeout 600 python -m pytest tests/test_gpu_shared_device.py -q -m gpu -p no:cacheprovider > $O/pytest_shared.log 2>&1; echo "pytest shared rc=$?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_step.py -q -m gpu -k "mul_mat_id or fused_step or one_launch or shared" -p no:cacheprovider > $O/pytest_moe.log 2>&1; echo "pytest parity rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/moe_ab.txt $O/moe_kernels.txt; head -12 $O/moe_trace.txt; head -12 $O/moe_trace_flat.txt; tail -5 $O/moe_trace.err; tail -15 $O/pytest_shared.log; tail -5 $O/pytest_moe.log; grep shared gpurun_out/parity_report.jsonl | tail -3
