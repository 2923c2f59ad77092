#!/bin/bash
# round 6, session 7: k_gemm_r8_sk (256-row tiles) against k_gemm_kq_sk (128-row tiles) — parity, A/B, cost weights, trace, per-kernel durations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/r6s7; mkdir -p $O; rm -rf $O/*
timeout 300 python scripts/moe_stability.py 50 >> $O/summary.txt 2>> $O/stability.err
for rep in 1 2; do
  AB_TAG=sk128 CDNA4_SK_TILE=128 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk256_flat timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk256_lin CDNA4_SK_CW8=6,7,8,9,10,11,12,13 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk256_steep CDNA4_SK_CW8=3,5,7,9,11,13,15,17 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=per_tile CDNA4_MOE_SK=0 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
done
timeout 120 python scripts/moe_trace.py > $O/moe_trace_256_flat.txt 2> $O/moe_trace.err
CDNA4_SK_CW8=6,7,8,9,10,11,12,13 timeout 120 python scripts/moe_trace.py > $O/moe_trace_256_lin.txt 2>> $O/moe_trace.err
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/moe_prof -o moe -- python $R/scripts/moe_ab.py > $O/moe_prof.txt 2>&1
cd $R
python - <<'PY' > $O/moe_kernels.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/r6s7/moe_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $O/moe_prof
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "mul_mat_id" -p no:cacheprovider > $O/pytest_moe.log 2>&1; echo "pytest parity rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/moe_ab.txt $O/moe_kernels.txt; head -9 $O/moe_trace_256_flat.txt; head -7 $O/moe_trace_256_lin.txt; tail -3 $O/moe_trace.err; tail -4 $O/pytest_moe.log
