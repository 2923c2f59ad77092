#!/bin/bash
# full GPU session: every -m gpu test, smoke, bench, rocprofv3 kernel stats + PMC passes (separate runs, kernel-trace only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -rf gpurun_out/summary.txt gpurun_out/parity_report.jsonl gpurun_out/gpt2_parity.jsonl gpurun_out/prof_* gpurun_out/pmc_*
export PYTHONUNBUFFERED=1
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest -m gpu rc=$?" >> gpurun_out/summary.txt; tail -5 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt
timeout -k 10 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/summary.txt
# gemm_bench (product kernels): auto route, the round-1 loader-wave kernel (4119), k_gemm_kq_t64 with 128-row (24583) and 256-row (40967) tiles
GB_SPLITKS="0" GB_VARIANTS="0,4119,24583" timeout 180 tools/microbench/gemm_bench 4096 4096 512 "" 2>&1 | grep -E "^M=|^variant" > gpurun_out/gemm_bench.txt
for shape in "4096 11008 512" "8192 4096 512" "8192 8192 512" "4096 14336 512"; do
  GB_SPLITKS="0" GB_VARIANTS="0,4119,24583" timeout 180 tools/microbench/gemm_bench $shape "" 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench.txt
done
GB_SPLITKS="0" GB_VARIANTS="0,4119,24583,40967" timeout 180 tools/microbench/gemm_bench 32768 8192 512 "" 2>&1 | grep -E "^M=|^variant" >> gpurun_out/gemm_bench.txt
# timing-only ablations + in-kernel stage trace of k_gemm_kq_t64 (tools/microbench: make abl), the MFMA / VALU probe
if [ -x tools/microbench/gemm_bench_abl ]; then
  B=40967; V="$B"; for a in 1 3 4 8 15 32; do V="$V,$((B + a*65536))"; done
  (cd tools/microbench; GB_VARIANTS="$V" GB_SPLITKS=1 GB_TRACE_REPS=30 timeout 300 ./gemm_bench_abl 32768 8192 512 "$((B + 256*65536))") > gpurun_out/t64_ablations.txt 2>&1
  B=24583; V="4119,$B"; for a in 1 3 4 8 15 32; do V="$V,$((B + a*65536))"; done
  (cd tools/microbench; GB_VARIANTS="$V" GB_SPLITKS=2 GB_TRACE_SPLITK=2 GB_TRACE_REPS=30 timeout 300 ./gemm_bench_abl 4096 4096 512 "$((B + 256*65536))") >> gpurun_out/t64_ablations.txt 2>&1
fi
timeout 120 tools/microbench/mfma_valu 2000 > gpurun_out/mfma_valu_probe.txt 2>&1
bash scripts/gpu_pmc1.sh > gpurun_out/pmc_gemm_bench_sq.txt 2>&1
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 200 --warmup 20 --lean > "$R/gpurun_out/rocprof_stats.log" 2>&1
echo "rocprof stats rc=$?" >> "$R/gpurun_out/summary.txt"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmc_$name" -o p -- python "$R/bench.py" --steps 100 --warmup 10 --lean > "$R/gpurun_out/rocprof_pmc_$name.log" 2>&1
  echo "pmc [$pmc] rc=$?" >> "$R/gpurun_out/summary.txt"
done
cd "$R"
# only summaries travel back (gpurun merges at most 64 MiB): condense, then drop the raw rocprofv3 output
rocminfo > gpurun_out/rocminfo.txt 2>&1; nproc > gpurun_out/nproc.txt
python tools/summarize_prof.py gpurun_out gpurun_out/profile_summary > gpurun_out/summarize.log 2>&1
rm -rf gpurun_out/prof_stats gpurun_out/pmc_* gpurun_out/pmcg_*
cat gpurun_out/summary.txt; grep -E "^M=|^variant" gpurun_out/gemm_bench.txt | head -60; tail -1 gpurun_out/bench.log
