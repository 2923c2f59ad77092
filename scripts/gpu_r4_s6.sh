#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
L=268435456; W=$((L+33554432))
cd tools/microbench
for shape in "32768 8192 512" "16384 8192 512" "16384 4096 512" "4096 4096 2048" "8192 8192 1024" "32768 4096 512"; do
  GB_VARIANTS="0,$W" GB_SPLITKS=0 GB_ROUNDS=5 timeout 120 ./gemm_bench $shape
done > ../../$O/gemm_bench.txt 2>&1
cd /tmp
GB_ROUNDS=1 GB_VARIANTS="$W" GB_SPLITKS=0 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -o p -- $GRAFT_REPO_ROOT/tools/microbench/gemm_bench 32768 8192 512 "" > $GRAFT_REPO_ROOT/$O/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > $O/pmc_summary.txt
import csv, glob, collections, os
for f in glob.glob('gpurun_out/s6/pmc/**/p_counter_collection.csv', recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,cs in agg.items():
        if 'gemm' in k: print(k[:60], {c: round(sum(v)/len(v)) for c,v in cs.items()})
PY
rm -rf $O/pmc
cat $O/pmc_summary.txt $O/gemm_bench.txt
