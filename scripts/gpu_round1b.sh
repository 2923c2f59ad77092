#!/bin/bash
# second GPU session: full -m gpu suite (per file), smoke, bench, variant sweep, rocprof stats + PMC passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt gpurun_out/variants.txt gpurun_out/parity_report.jsonl
export PYTHONUNBUFFERED=1
for f in test_gpu_parity test_gpu_backend_plugin test_gpu_gpt2; do
  timeout -k 10 1200 python -m pytest tests/$f.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_$f.log 2>&1
  echo "$f rc=$?" >> gpurun_out/summary.txt; tail -4 gpurun_out/pytest_$f.log >> gpurun_out/summary.txt
done
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/summary.txt
for v in 5 13 7 4; do for sk in 1 2; do
  timeout -k 10 120 python bench.py --steps 100 --warmup 10 --variant $v --splitk $sk --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('variant',$v,'splitk',$sk,'step_tflops',j['value'],'gemm_us',j['roofline']['us_per_launch'],'gemm_tflops',j['roofline']['achieved'],'gemv_cold_us',j['decode']['us_per_gemv_cold_hbm'],'gemv_warm_us',j['decode']['us_per_gemv_cache_warm'])" >> gpurun_out/variants.txt 2>&1
done; done
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$R/gpurun_out/rocprof_stats.log" 2>&1
echo "rocprof stats rc=$?" >> "$R/gpurun_out/summary.txt"
rocprofv3 -L > "$R/gpurun_out/rocprof_counters.txt" 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/prof_pmc_$name" -o p -- python "$R/bench.py" --steps 20 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/rocprof_pmc_$name.log" 2>&1
  echo "pmc [$pmc] rc=$?" >> "$R/gpurun_out/summary.txt"
done
cd "$R"; cat gpurun_out/summary.txt; cat gpurun_out/variants.txt; tail -1 gpurun_out/bench.log
