#!/bin/bash
# round 6, session 8: stream-k MUL_MAT_ID as it ships (128-row tiles, lean planner with hoisted id loads, padding made in the LDS copy): A/B, per-kernel durations, trace, tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/r6s8; mkdir -p $O; rm -rf $O/*
for rep in 1 2 3; do
  AB_TAG=sk timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
  AB_TAG=per_tile CDNA4_MOE_SK=0 timeout 120 python scripts/moe_ab.py >> $O/moe_ab.txt 2>> $O/moe_ab.err
done
timeout 120 python scripts/moe_trace.py > $O/moe_trace.txt 2> $O/moe_trace.err
cd /tmp
for abl in 0 1; do
CDNA4_SK_FRONT_ABL=$abl timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/moe_prof$abl -o moe -- python $R/scripts/moe_ab.py > $O/moe_prof$abl.txt 2>&1
done
cd $R
python - <<'PY' > $O/moe_kernels.txt 2>&1
import csv, glob
for abl in (0, 1):
  for f in glob.glob("gpurun_out/r6s8/moe_prof%d/**/*kernel_stats.csv" % abl, recursive=True):
    for r in csv.DictReader(open(f)):
        if "front" in r["Name"] or "gemm_kq_sk" in r["Name"]: print("front_abl", abl, r["Name"][:80], r["Calls"], r["AverageNs"])
PY
rm -rf $O/moe_prof*/
timeout 300 python scripts/moe_stability.py 200 >> $O/summary.txt 2>> $O/stability.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_resident.py tests/test_gpu_backend_plugin.py -q -m gpu -k "mul_mat_id or MUL_MAT_ID or expert_stack" -p no:cacheprovider > $O/pytest_moe.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/moe_ab.txt $O/moe_kernels.txt; head -8 $O/moe_trace.txt; tail -4 $O/pytest_moe.log
