#!/bin/bash
# First GPU minutes of the next round: the formats added after round 2's last hardware session (Q2_K prefill, Q4_1 / Q5_1 / IQ4_NL / IQ4_XS) —
# C-ABI vs oracle over ctypes (torch-free), then the stock harness on the plug-in.   ~1.5 min
#   gpurun --timeout 240 -- bash scripts/gpu_widening2.sh
export GGML_BACKEND_PATH=$PWD/ggml_amd/lib/libggml-cdna4.so
mkdir -p gpurun_out; rm -f gpurun_out/widening_check.jsonl
timeout 120 python scripts/gpu_check_widening.py formats2 fattn2 > gpurun_out/widening_check2.log 2>&1; echo "check rc=$?"; grep -c '"ok": true' gpurun_out/widening_check2.log; grep '"ok": false' gpurun_out/widening_check2.log | cut -c1-300; tail -1 gpurun_out/widening_check2.log
for op in MUL_MAT MUL_MAT_ID GET_ROWS CPY FLASH_ATTN_EXT; do
    timeout 45 oracle/_ref/test-backend-ops test -o $op -b CDNA40 > gpurun_out/tbo_$op.log 2>&1
    echo "$op rc=$? ok=$(grep -c ': .*OK' gpurun_out/tbo_$op.log) fail=$(grep -c FAIL gpurun_out/tbo_$op.log)"
done
