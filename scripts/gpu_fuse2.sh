#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sched.py tests/test_gpu_gpt2.py tests/test_gpu_backend_plugin.py -m gpu -q -x > gpurun_out/graph_tests.log 2>&1
tail -15 gpurun_out/graph_tests.log; grep peepholes gpurun_out/gpt2_parity.jsonl | tail -1 | cut -c1-400
GGML_BACKEND_PATH=ggml_amd/lib/libggml-cdna4.so GGML_CDNA4_STATS=1 timeout 200 oracle/_ref/test-backend-ops perf -o MUL_MAT -b CDNA40 2>&1 | grep -E "q4_K|q4_0.*n=1,|HIP-graph" | head -12
