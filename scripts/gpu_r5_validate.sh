#!/bin/bash
# round 5, validation of HEAD after the final profiling session (instantiation prunes, buffer_from_host_ptr): every -m gpu test, smoke, bench.py as the driver runs it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
rm -f gpurun_out/summary.txt gpurun_out/parity_report.jsonl gpurun_out/split_report.jsonl
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" >> gpurun_out/summary.txt; tail -6 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_driver_cmd.log 2> gpurun_out/bench.err; echo "bench (driver's command, no extras) rc=$?" >> gpurun_out/summary.txt
python tools/summarize_prof.py gpurun_out gpurun_out/profile_summary_v > gpurun_out/summarize.log 2>&1
cat gpurun_out/summary.txt; grep -h 'hostptr\|declined' gpurun_out/split_report.jsonl | tail -3; head -c 600 gpurun_out/bench_driver_cmd.log; echo; grep -i 'fail\|error' gpurun_out/pytest_gpu.log | head -20
