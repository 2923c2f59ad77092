#!/bin/bash
# round 5, last GPU session (9.9 minutes left): smoke, the driver's bench command, the resident-image bench leg, then the -m gpu tests in order of what changed since the
# last full run (890 passed on 4527c2d; tools/isa_manifest.py: no kernel that ran then has changed — new: k_gemm_r8<Q8_0R | Q6_K8>, k_repack_q6_K8, k_norm<.., Q8K>) until
# the time budget is spent; what did not run is listed
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
T0=$(date +%s); BUDGET=${BUDGET:-470}
left() { echo $(( BUDGET - ($(date +%s) - T0) )); }
mkdir -p gpurun_out; rm -f gpurun_out/summary.txt gpurun_out/split_report.jsonl gpurun_out/parity_report.jsonl
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary.txt
timeout -k 5 90 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_driver_cmd.log 2> gpurun_out/bench.err; echo "bench (driver's command, no extras) rc=$?" >> gpurun_out/summary.txt
timeout -k 5 90 python bench.py --leg resident_images > gpurun_out/bench_resident_images.log 2>> gpurun_out/bench.err; echo "bench --leg resident_images rc=$?" >> gpurun_out/summary.txt
run() {   # name, pytest args...
  local name=$1; shift
  local l=$(left)
  if [ "$l" -lt 25 ]; then echo "NOT RUN (time): $name" >> gpurun_out/summary.txt; return; fi
  timeout -k 5 "$l" python -m pytest "$@" -m gpu -q --tb=short -p no:cacheprovider > "gpurun_out/pytest_$name.log" 2>&1
  echo "pytest $name rc=$? : $(tail -1 gpurun_out/pytest_$name.log)" >> gpurun_out/summary.txt
}
run act_share tests/test_gpu_act_share.py
run resident tests/test_gpu_resident.py -k "not soak"
run cabi_ops tests/test_gpu_cabi_ops.py
run backend_plugin tests/test_gpu_backend_plugin.py
run sched tests/test_gpu_sched.py
run gpt2 tests/test_gpu_gpt2.py
run fused_step_exact tests/test_gpu_fused_step.py tests/test_gpu_exact.py tests/test_gpu_bench.py
run parity tests/test_gpu_parity.py
run widening tests/test_gpu_widening.py
run resident_soak tests/test_gpu_resident.py -k "soak"
cat gpurun_out/summary.txt; head -c 500 gpurun_out/bench_driver_cmd.log; echo; cat gpurun_out/bench_resident_images.log | cut -c1-1500; grep -h -i 'failed\|error' gpurun_out/pytest_*.log | head -20
