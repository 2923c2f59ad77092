#!/bin/bash
# round 6, session 13: kernel-only durations of the small-batch route (quantizer + k_mmq) per row count, and k_gemm_r8 unsplit / split in two at C3
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/r6s13; mkdir -p $O; rm -rf $O/*
for sh in 4096x14336 4096x4096; do for b in 1 2 4 8 16 32; do
  ( cd /tmp && BATCH_SHAPES=$sh BATCH_ROWS=$b timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${sh}_$b -o p -- python $R/scripts/batch_q4k.py > $O/run_${sh}_$b.txt 2>&1 )
  f=$(find $O/prof_${sh}_$b -name "*kernel_stats.csv" | head -1)
  echo "== $sh rows $b: $(grep tag $O/run_${sh}_$b.txt | cut -c1-120)" >> $O/kernels.txt
  python - "$f" >> $O/kernels.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print("   %-110s calls %6s avg_ns %10s" % (r["Name"][:110], r["Calls"], r["AverageNs"]))
PY
done; done
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete
cat $O/kernels.txt | cut -c1-200
