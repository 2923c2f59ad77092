#!/bin/bash
# round 2, GPU call 3: scheduler integration (sched harness, split buffer type, async/events), C-ABI op tests, MUL_MAT_ID plug-in run
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
( timeout 900 python -m pytest tests/test_gpu_sched.py tests/test_gpu_cabi_ops.py -q -m gpu 2>&1 | tail -30 ) > gpurun_out/r2_sched_tests.txt
( timeout 900 python -m pytest tests/test_gpu_gpt2.py -q -m gpu 2>&1 | tail -30 ) > gpurun_out/r2_gpt2_tests.txt
( timeout 900 python -m pytest tests/test_gpu_backend_plugin.py -q -m gpu -k "MUL_MAT" 2>&1 | tail -15 ) > gpurun_out/r2_plugin_mm.txt
cat gpurun_out/r2_sched_tests.txt; cat gpurun_out/r2_gpt2_tests.txt; tail -5 gpurun_out/r2_plugin_mm.txt; tail -5 gpurun_out/gpt2_parity.jsonl | cut -c1-700
