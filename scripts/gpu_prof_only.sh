#!/bin/bash
# rocprofv3 passes of bench.py --lean only (kernel stats + the four PMC passes), condensed on the box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -rf gpurun_out/prof_* gpurun_out/pmc_*
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o r1 -- python "$R/bench.py" --steps 200 --warmup 20 --lean > "$R/gpurun_out/rocprof_stats.log" 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$R/gpurun_out/pmc_$name" -o p -- python "$R/bench.py" --steps 100 --warmup 10 --lean > "$R/gpurun_out/rocprof_pmc_$name.log" 2>&1
done
cd "$R"; python tools/summarize_prof.py gpurun_out gpurun_out/profile_summary2 > gpurun_out/summarize.log 2>&1
rm -rf gpurun_out/prof_stats gpurun_out/pmc_* gpurun_out/pmcg_*
cat gpurun_out/profile_summary2/rocprofv3_kernel_stats.csv | head -6; grep t64 gpurun_out/profile_summary2/pmc_summary.txt | cut -c1-300; tail -1 gpurun_out/rocprof_stats.log | cut -c1-600
