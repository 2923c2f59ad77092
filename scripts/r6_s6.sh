#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/r6s6; mkdir -p $O; rm -rf $O/*


cd /tmp
for abl in 0 1; do
  CDNA4_SK_FRONT_ABL=$abl timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$abl -o moe -- python $R/scripts/moe_ab.py > $O/prof_$abl.txt 2>&1
done
cd $R
python - <<'PY' > $O/front_kernels.txt 2>&1
import csv, glob
for abl in (0, 1):
    for f in glob.glob("gpurun_out/r6s6/prof_%d/**/*kernel_stats.csv" % abl, recursive=True):
        for r in csv.DictReader(open(f)):
            if "front" in r["Name"] or "gemm_kq_sk" in r["Name"]: print("abl", abl, r["Name"][:80], r["Calls"], r["AverageNs"])
PY
rm -rf $O/prof_*/
cat $O/front_kernels.txt; grep prefill $O/prof_0.txt
