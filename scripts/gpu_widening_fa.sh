#!/bin/bash
# FLASH_ATTN_EXT only: C-ABI vs oracle, stock harness, timings (torch-free, ~1 min)
export GGML_BACKEND_PATH=$PWD/ggml_amd/lib/libggml-cdna4.so
mkdir -p gpurun_out; rm -f gpurun_out/widening_check.jsonl
timeout 50 python scripts/gpu_check_widening.py fattn > gpurun_out/widening_check_fa.log 2>&1; echo "check rc=$?"; grep -c '"ok": true' gpurun_out/widening_check_fa.log; grep '"ok": false' gpurun_out/widening_check_fa.log | cut -c1-300; tail -1 gpurun_out/widening_check_fa.log
timeout 30 oracle/_ref/test-backend-ops test -o FLASH_ATTN_EXT -b CDNA40 > gpurun_out/tbo_FLASH_ATTN_EXT.log 2>&1
echo "FLASH_ATTN_EXT rc=$? ok=$(grep -c ': .*OK' gpurun_out/tbo_FLASH_ATTN_EXT.log) fail=$(grep -c FAIL gpurun_out/tbo_FLASH_ATTN_EXT.log)"
timeout 25 python scripts/gpu_check_widening.py timings_fa > gpurun_out/widening_timings_fa.log 2>&1; echo "timings rc=$?"; grep time_ gpurun_out/widening_timings_fa.log | cut -c1-220
