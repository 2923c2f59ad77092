#!/bin/bash
# PMC passes over gemm_bench (product kernels): where do the waves of the t64 / w12 kernels spend their cycles?
R="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$R/gpurun_out"; cd /tmp; export TMPDIR=/tmp
run() { # name, pmc list, M K B variants splitk
  local name=$1 pmc=$2; shift 2
  GB_ROUNDS=1 GB_VARIANTS="$4" GB_SPLITKS="$5" timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmcg_$name" -o p -- "$R/tools/microbench/gemm_bench" $1 $2 $3 "" > "$R/gpurun_out/pmcg_$name.log" 2>&1
}
for shape in "c5 32768 8192 512 40967 1" "head 4096 4096 512 24583,4119 2"; do
  set -- $shape; tag=$1; shift
  run "${tag}_a" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "$@"
  run "${tag}_b" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" "$@"
  run "${tag}_c" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_LDS" "$@"
  run "${tag}_d" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "$@"
done
cd "$R"; python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmcg_*')):
    f=os.path.join(d,'p_counter_collection.csv')
    if not os.path.isdir(d) or not os.path.exists(f): print(d,'no csv'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,cs in agg.items():
        if 'gemm' not in k: continue
        print(os.path.basename(d), k[:60], {c: round(sum(v)/len(v)) for c,v in cs.items()}, 'n=%d' % len(next(iter(cs.values()))))
PY
