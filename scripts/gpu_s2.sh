#!/bin/bash
# round 3, session 2: the tests that failed in session 1 + the new ones, the determinism diagnosis, and the write-through store experiments
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "q8_1 or fused or other_quantizations or q4_1_q5_1_iq4_nl_prefill_gemm or two_part or smoke" > gpurun_out/s2_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)" gpurun_out/s2_pytest.log | head; tail -2 gpurun_out/s2_pytest.log
timeout 600 python scripts/gpu_diag_determinism.py > gpurun_out/s2_determinism.log 2>&1; echo "determinism rc=$?"; cat gpurun_out/s2_determinism.log | cut -c1-400
for ts in 0 1; do for qs in 0 1; do
  CDNA4_T64_STORE=$ts CDNA4_QUANT_STORE=$qs timeout 300 python bench.py --lean --steps 400 > gpurun_out/s2_bench_$ts$qs.log 2>&1
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/s2_bench_$ts$qs.log').read().strip().splitlines()[-1])
    print("T64_STORE=$ts QUANT_STORE=$qs step_us %.2f gemm_us %.2f value %.1f" % (d['ms_per_step']*1e3, d['roofline']['us_per_launch'], d['value']))
except Exception as e: print("bench $ts$qs failed", e)
PY
done; done
