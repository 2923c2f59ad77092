cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -rf gpurun_out/prof_final
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_final" -o r1 -- python "$R/bench.py" --steps 200 --warmup 20 --lean > "$R/gpurun_out/rocprof_final_stats.log" 2>&1
cd "$R"; f=$(find gpurun_out/prof_final -name "r1_kernel_stats.csv" | head -1); cp "$f" gpurun_out/rocprofv3_kernel_stats_final.csv; head -4 gpurun_out/rocprofv3_kernel_stats_final.csv
grep -E "^\{" gpurun_out/rocprof_final_stats.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench --lean under rocprofv3: step_us %.2f kernel_us(HIP events) %.2f frac %.4f' % (d['ms_per_step']*1e3, d['roofline']['us_per_launch'], d['roofline']['frac']))"
rm -rf gpurun_out/prof_final
