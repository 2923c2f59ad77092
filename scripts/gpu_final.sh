#!/bin/bash
# a round's full session on HEAD: every -m gpu test, smoke, bench.py (defaults, then the driver's own command), rocprofv3 kernel stats + four PMC passes over `bench.py --lean`
# (separate runs, kernel-trace only), the same for the decode loop at 4096 x 14336 and (stats only) the grouped MUL_MAT_ID loop, gemm_bench over the BASELINE shapes; condensed on
# the box (tools/summarize_prof.py) -> gpurun_out/profile_summary/, to be copied into profiles/rNN/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -rf gpurun_out/summary.txt gpurun_out/parity_report.jsonl gpurun_out/gpt2_parity.jsonl gpurun_out/split_report.jsonl gpurun_out/prof_* gpurun_out/pmc_* gpurun_out/profile_summary
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest -m gpu rc=$?" >> gpurun_out/summary.txt; tail -5 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt
timeout -k 10 700 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/summary.txt
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_driver_cmd.log 2>> gpurun_out/bench.err; echo "bench (driver's command, no extras) rc=$?" >> gpurun_out/summary.txt
( cd tools/microbench
  for shape in "4096 4096 512" "4096 11008 512" "8192 4096 512"; do GB_SPLITKS=0 GB_VARIANTS="0,24583" GB_ROUNDS=5 timeout 120 ./gemm_bench $shape ""; done
  L=268435456; R8=$((L+67108864))
  for shape in "32768 8192 512" "16384 8192 512"; do GB_SPLITKS=0 GB_VARIANTS="0,40967,$R8" GB_ROUNDS=5 timeout 120 ./gemm_bench $shape ""; done
) 2>&1 | grep -E "^M=|^variant" > gpurun_out/gemm_bench.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 200 --warmup 20 --lean > "$R/gpurun_out/rocprof_stats.log" 2>&1
echo "rocprof stats rc=$?" >> "$R/gpurun_out/summary.txt"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmc_$name" -o p -- python "$R/bench.py" --steps 100 --warmup 10 --lean > "$R/gpurun_out/rocprof_pmc_$name.log" 2>&1
  echo "pmc [$pmc] rc=$?" >> "$R/gpurun_out/summary.txt"
done
cd "$R"
rocminfo > gpurun_out/rocminfo.txt 2>&1; nproc > gpurun_out/nproc.txt
python tools/summarize_prof.py gpurun_out gpurun_out/profile_summary > gpurun_out/summarize.log 2>&1
# the decode kernel at 4096 x 14336: kernel stats + instruction / wave-cycle counters + traffic (the one-launch k_gemv_q_fused<Q4_K, 16, 1>)
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_decode" -o r1 -- python "$R/scripts/decode_loop.py" > "$R/gpurun_out/rocprof_decode.log" 2>&1
for pmc in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo "$pmc" | tr ' ' '+' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmcd_$name" -o p -- python "$R/scripts/decode_loop.py" > "$R/gpurun_out/rocprof_pmcd_$name.log" 2>&1
done
cd "$R"
python3 - > gpurun_out/profile_summary/pmc_decode_counters.txt <<'PY'
import csv, glob, collections, os
print("# rocprofv3 --kernel-trace --pmc <pass> -- python scripts/decode_loop.py (Q4_K 4096 x 14336, one launch per token, rotating matrices); per-launch averages")
for d in sorted(glob.glob('gpurun_out/pmcd_*')):
    f=os.path.join(d,'p_counter_collection.csv')
    if not os.path.isdir(d) or not os.path.exists(f): continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for r in csv.DictReader(open(os.path.join(d,'p_kernel_trace.csv'))): dur[r['Kernel_Name']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,cs in agg.items():
        if 'gemv' not in k: continue
        print(os.path.basename(d)[6:], k[:60], 'n=%d' % len(next(iter(cs.values()))), 'avg_dur_us(profiled)=%.2f' % (sum(dur[k])/max(1,len(dur[k]))), {c: round(sum(v)/len(v)) for c,v in cs.items()})
PY
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_moe" -o r1 -- python "$R/scripts/moe_ab.py" > "$R/gpurun_out/rocprof_moe.log" 2>&1 )
MS=$(find gpurun_out/prof_moe -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$MS" ]; then head -6 "$MS" > gpurun_out/profile_summary/rocprofv3_moe_kernel_stats.csv; grep tag gpurun_out/rocprof_moe.log > gpurun_out/profile_summary/moe_under_profiler.txt; fi
if [ -f gpurun_out/prof_decode/r1_kernel_stats.csv ]; then cp gpurun_out/prof_decode/r1_kernel_stats.csv gpurun_out/profile_summary/rocprofv3_decode_kernel_stats.csv; fi
cp gpurun_out/pytest_gpu.log gpurun_out/profile_summary/pytest_gpu_final.log 2>/dev/null
cp gpurun_out/bench_driver_cmd.log gpurun_out/profile_summary/ 2>/dev/null
cp gpurun_out/bench.log gpurun_out/profile_summary/bench_final.log 2>/dev/null; cp gpurun_out/summary.txt gpurun_out/profile_summary/summary_final.txt
rm -rf gpurun_out/prof_stats gpurun_out/prof_decode gpurun_out/prof_moe gpurun_out/pmc_* gpurun_out/pmcd_*
cat gpurun_out/summary.txt; cat gpurun_out/gemm_bench.txt | cut -c1-160; cat gpurun_out/profile_summary/pmc_decode_counters.txt | cut -c1-400; head -c 1200 gpurun_out/bench_driver_cmd.log; echo; grep -v '^$' gpurun_out/profile_summary/pmc_summary.txt | cut -c1-330 | head -40
