#!/bin/bash
# full GPU session: every -m gpu test, smoke, bench, rocprofv3 kernel stats + PMC passes (separate runs, kernel-trace only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -rf gpurun_out/summary.txt gpurun_out/parity_report.jsonl gpurun_out/gpt2_parity.jsonl gpurun_out/prof_* gpurun_out/pmc_*
export PYTHONUNBUFFERED=1
timeout -k 10 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" >> gpurun_out/summary.txt; tail -5 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/summary.txt
GB_TRACE_SPLITK=2 GB_SPLITKS="0,1,2" GB_VARIANTS="0,5,23,407,663,1031,2071" timeout 180 tools/microbench/gemm_bench 4096 4096 512 663 > gpurun_out/gemm_bench.txt 2>&1
for shape in "8192 4096 512" "4096 10752 512" "4096 11008 512" "4096 8192 512" "32768 8192 512"; do   # 11008 = the true C3 K (43 superblocks, odd: no split unless CDNA4_ODD_SPLIT=1)
  GB_SPLITKS="0" GB_VARIANTS="0,663,1031" timeout 180 tools/microbench/gemm_bench $shape 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench.txt
done
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$R/gpurun_out/rocprof_stats.log" 2>&1
echo "rocprof stats rc=$?" >> "$R/gpurun_out/summary.txt"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmc_$name" -o p -- python "$R/bench.py" --steps 30 --warmup 3 --no-cpu-baseline > "$R/gpurun_out/rocprof_pmc_$name.log" 2>&1
  echo "pmc [$pmc] rc=$?" >> "$R/gpurun_out/summary.txt"
done
cd "$R"; cat gpurun_out/summary.txt; grep -E "^M=|^variant" gpurun_out/gemm_bench.txt | head -40; tail -1 gpurun_out/bench.log
