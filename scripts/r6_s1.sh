#!/bin/bash
# round 6, session 1: baselines for the round's kernel work — per-kernel durations of the grouped MUL_MAT_ID prefill (plan / gather-quantize / GEMM), the batch sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$PWD; O=$R/gpurun_out/r6s1; mkdir -p $O; rm -rf $O/*
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/moe_prof -o moe -- python $R/scripts/moe_ab.py > $O/moe.txt 2>&1
cd $R
python - <<'PY' > $O/moe_kernels.txt 2>&1
import csv, glob
for f in glob.glob("gpurun_out/r6s1/moe_prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $O/moe_prof
AB_TAG=tm128 CDNA4_MOE_TM=128 timeout 120 python scripts/moe_ab.py >> $O/moe.txt 2>&1
timeout 900 python scripts/gpu_batch_sweep.py > $O/batch_sweep.txt 2> $O/batch_sweep.err
cat $O/moe.txt $O/moe_kernels.txt; head -60 $O/batch_sweep.txt
