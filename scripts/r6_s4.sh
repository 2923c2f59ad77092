#!/bin/bash
# round 6, session 4: stream-k MUL_MAT_ID — planner with its stores behind the last barrier, no activation DMA for fragments without rows: cost weights, trace, per-kernel durations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/r6s4; mkdir -p $O; rm -rf $O/*
for rep in 1 2; do
  for cw in 10,10,10,10 5,7,9,10 6,8,9,10 7,8,9,10 8,9,10,10 4,6,8,10; do
    AB_TAG=sk_cw_$cw CDNA4_SK_CW=$cw timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  done
  AB_TAG=per_tile CDNA4_MOE_SK=0 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
done
CDNA4_SK_CW=10,10,10,10 timeout 120 python scripts/moe_trace.py > $O/moe_trace_flat.txt 2> $O/moe_trace.err
CDNA4_SK_CW=6,8,9,10 timeout 120 python scripts/moe_trace.py > $O/moe_trace_6_8_9_10.txt 2>> $O/moe_trace.err
cd /tmp
CDNA4_SK_CW=10,10,10,10 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/moe_prof -o moe -- python $R/scripts/moe_ab.py > $O/moe_prof.txt 2>&1
cd $R
python - <<'PY' > $O/moe_kernels.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/r6s4/moe_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $O/moe_prof
timeout 300 python scripts/moe_stability.py 100 >> $O/summary.txt 2>> $O/stability.err
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "mul_mat_id" -p no:cacheprovider > $O/pytest_moe.log 2>&1; echo "pytest parity rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/moe_ab.txt $O/moe_kernels.txt; head -8 $O/moe_trace_flat.txt; head -8 $O/moe_trace_6_8_9_10.txt; tail -3 $O/moe_trace.err; tail -4 $O/pytest_moe.log
