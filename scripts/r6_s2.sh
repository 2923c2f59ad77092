#!/bin/bash
# round 6, session 2: the stream-k grouped MUL_MAT_ID on hardware — parity tests, bit-stability, A/B against the per-tile launches, cost weights, per-kernel durations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/r6s2; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "mul_mat_id" -x -p no:cacheprovider > $O/pytest_moe.log 2>&1; echo "pytest parity rc=$?" >> $O/summary.txt
timeout 300 python scripts/moe_stability.py 200 >> $O/summary.txt 2>> $O/stability.err
for rep in 1 2; do
  AB_TAG=sk_default timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=per_tile CDNA4_MOE_SK=0 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk_cw_flat CDNA4_SK_CW=10,10,10,10 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk_cw_8_12_16_20 CDNA4_SK_CW=8,12,16,20 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk_cw_14_16_18_20 CDNA4_SK_CW=14,16,18,20 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=sk_spans_128 CDNA4_SK_SPANS=128 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
done
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/moe_prof -o moe -- python $R/scripts/moe_ab.py > $O/moe_prof.txt 2>&1
cd $R
python - <<'PY' > $O/moe_kernels.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/r6s2/moe_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $O/moe_prof
timeout 600 python -m pytest tests/test_gpu_backend_plugin.py tests/test_gpu_resident.py -q -m gpu -k "MUL_MAT_ID or expert_stack" -p no:cacheprovider > $O/pytest_plugin.log 2>&1; echo "pytest plugin rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/moe_ab.txt $O/moe_kernels.txt; tail -5 $O/pytest_moe.log; tail -5 $O/pytest_plugin.log
