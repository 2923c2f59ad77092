#!/bin/bash
# One short, torch-free GPU session for the widening rows: C-ABI checks against the oracle, the stock harness on the plug-in, a few timings.
#   gpurun --timeout 200 -- bash scripts/gpu_widening.sh
export GGML_BACKEND_PATH=$PWD/ggml_amd/lib/libggml-cdna4.so
mkdir -p gpurun_out; rm -f gpurun_out/widening_check.jsonl
timeout 60 python scripts/gpu_check_widening.py formats fattn > gpurun_out/widening_check.log 2>&1; echo "check rc=$?"; tail -3 gpurun_out/widening_check.log
for op in FLASH_ATTN_EXT MUL_MAT GET_ROWS CPY; do
    timeout 35 oracle/_ref/test-backend-ops test -o $op -b CDNA40 > gpurun_out/tbo_$op.log 2>&1
    echo "$op rc=$? ok=$(grep -c ': .*OK' gpurun_out/tbo_$op.log) fail=$(grep -c FAIL gpurun_out/tbo_$op.log)"
done
timeout 30 python scripts/gpu_check_widening.py timings > gpurun_out/widening_timings.log 2>&1; echo "timings rc=$?"; grep time_ gpurun_out/widening_timings.log | cut -c1-220
