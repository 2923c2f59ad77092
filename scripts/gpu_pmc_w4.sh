#!/bin/bash
# PMC passes over gemm_bench at C5: k_gemm_kq_t64 (variant 0) beside k_gemm_w4 (bits 28 + 25) — wave-cycle split, instruction mix, LDS conflicts
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/pmc_w4"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
W=$((268435456+33554432))
run() { # name, pmc list
  local name=$1 pmc=$2
  GB_ROUNDS=1 GB_VARIANTS="0,$W" GB_SPLITKS=0 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$O/$name" -o p -- "$R/tools/microbench/gemm_bench" 32768 8192 512 "" > "$O/$name.log" 2>&1
}
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
run c "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_LDS"
run d "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
cd "$R"; python3 - <<'PY' > gpurun_out/pmc_w4/summary.txt
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc_w4/*')):
    f=os.path.join(d,'p_counter_collection.csv')
    if not os.path.isdir(d) or not os.path.exists(f): continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,cs in agg.items():
        if 'gemm' not in k: continue
        print(os.path.basename(d), k[:60], {c: round(sum(v)/len(v)) for c,v in cs.items()}, 'n=%d' % len(next(iter(cs.values()))))
PY
rm -rf gpurun_out/pmc_w4/a gpurun_out/pmc_w4/b gpurun_out/pmc_w4/c gpurun_out/pmc_w4/d
cat gpurun_out/pmc_w4/summary.txt
