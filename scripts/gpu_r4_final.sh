#!/bin/bash
# round 4, full session: every -m gpu test, smoke, bench, gemm_bench (t64 / w4 / r8 on one box), rocprofv3 kernel stats + PMC passes (separate runs, kernel-trace only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -rf gpurun_out/summary.txt gpurun_out/parity_report.jsonl gpurun_out/gpt2_parity.jsonl gpurun_out/split_report.jsonl gpurun_out/prof_* gpurun_out/pmc_* gpurun_out/pmcg_*
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest -m gpu rc=$?" >> gpurun_out/summary.txt; tail -5 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt
timeout -k 10 700 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/summary.txt
# one box, one process per shape: k_gemm_kq_t64 (its 128- / 256-row forms), k_gemm_w4, k_gemm_r8, AUTO
L=268435456; W=$((L+33554432)); R8=$((L+67108864))
( cd tools/microbench
  for shape in "4096 4096 512" "4096 11008 512" "8192 4096 512" "8192 8192 512"; do GB_SPLITKS=0 GB_VARIANTS="0,24583,$W,$R8" GB_ROUNDS=5 timeout 180 ./gemm_bench $shape ""; done
  for shape in "32768 8192 512" "16384 8192 512" "16384 4096 512" "4096 4096 2048" "8192 8192 1024" "32768 4096 512"; do GB_SPLITKS=0 GB_VARIANTS="0,40967,$W,$R8" GB_ROUNDS=5 timeout 180 ./gemm_bench $shape ""; done
) 2>&1 | grep -E "^M=|^variant" > gpurun_out/gemm_bench.txt
if [ -x tools/microbench/gemm_bench_abl ]; then
  V=""; for a in 0 1 2 3 4 8 16 32 15; do V="$V,$((R8 + a*65536))"; done
  (cd tools/microbench; GB_VARIANTS="${V:1}" GB_SPLITKS=0 GB_ROUNDS=3 timeout 300 ./gemm_bench_abl 32768 8192 512) > gpurun_out/r8_ablations.txt 2>&1
fi
timeout 600 python scripts/gpu_decode_ab.py > gpurun_out/decode_ab.txt 2>&1
cd /tmp
# rocprofv3: kernel stats of the default bench command (headline) and of the C5 shape through bench.py --config c5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 200 --warmup 20 --lean > "$R/gpurun_out/rocprof_stats.log" 2>&1
echo "rocprof stats rc=$?" >> "$R/gpurun_out/summary.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats_c5" -o r1 -- python "$R/bench.py" --config c5 --steps 40 --warmup 5 > "$R/gpurun_out/rocprof_stats_c5.log" 2>&1
echo "rocprof stats c5 rc=$?" >> "$R/gpurun_out/summary.txt"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmc_$name" -o p -- python "$R/bench.py" --steps 100 --warmup 10 --lean > "$R/gpurun_out/rocprof_pmc_$name.log" 2>&1
  echo "pmc [$pmc] rc=$?" >> "$R/gpurun_out/summary.txt"
done
# C5: k_gemm_kq_t64<256> beside k_gemm_r8 under the SQ counters and the HBM-side traffic
run() { local name=$1 pmc=$2
  GB_ROUNDS=1 GB_VARIANTS="40967,$R8" GB_SPLITKS=0 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmcg_$name" -o p -- "$R/tools/microbench/gemm_bench" 32768 8192 512 "" > "$R/gpurun_out/pmcg_$name.log" 2>&1; }
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
run c "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_LDS"
run d "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run f "FETCH_SIZE"
run w "WRITE_SIZE"
cd "$R"
python3 - > gpurun_out/gemm_c5_pmc.txt <<'PY'
import csv, glob, collections, os
print("# rocprofv3 --kernel-trace --pmc <pass> -- tools/microbench/gemm_bench 32768 8192 512 (variants: k_gemm_kq_t64<256> = 40967, k_gemm_r8); per-launch averages")
for d in sorted(glob.glob('gpurun_out/pmcg_*')):
    f=os.path.join(d,'p_counter_collection.csv')
    if not os.path.isdir(d) or not os.path.exists(f): continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for r in csv.DictReader(open(os.path.join(d,'p_kernel_trace.csv'))): dur[r['Kernel_Name']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,cs in agg.items():
        if 'gemm' not in k: continue
        print(os.path.basename(d)[5:], k[:48], 'n=%d' % len(next(iter(cs.values()))), 'avg_dur_us(profiled)=%.2f' % (sum(dur[k])/max(1,len(dur[k]))), {c: round(sum(v)/len(v)) for c,v in cs.items()})
PY
if [ -f gpurun_out/prof_stats_c5/r1_kernel_stats.csv ]; then cp gpurun_out/prof_stats_c5/r1_kernel_stats.csv gpurun_out/rocprofv3_kernel_stats_c5.csv; fi
rocminfo > gpurun_out/rocminfo.txt 2>&1; nproc > gpurun_out/nproc.txt
python tools/summarize_prof.py gpurun_out gpurun_out/profile_summary > gpurun_out/summarize.log 2>&1
rm -rf gpurun_out/prof_stats gpurun_out/prof_stats_c5 gpurun_out/pmc_* gpurun_out/pmcg_a gpurun_out/pmcg_c gpurun_out/pmcg_d gpurun_out/pmcg_f gpurun_out/pmcg_w
cat gpurun_out/summary.txt; cat gpurun_out/gemm_bench.txt | cut -c1-140; cat gpurun_out/gemm_c5_pmc.txt | cut -c1-400; tail -c 600 gpurun_out/bench.log
