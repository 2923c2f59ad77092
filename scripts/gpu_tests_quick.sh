#!/bin/bash
# a quick GPU pass: kernel parity + plug-in (stock test-backend-ops) + the stock perf lines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backend_plugin.py tests/test_gpu_cabi_ops.py -m gpu -q -x -p no:cacheprovider > gpurun_out/quick_tests.log 2>&1
echo "rc=$?"; tail -6 gpurun_out/quick_tests.log
GGML_BACKEND_PATH=ggml_amd/lib/libggml-cdna4.so timeout 200 oracle/_ref/test-backend-ops perf -o MUL_MAT -b CDNA40 2>&1 | grep -E "type_a=q4_K|type_a=q4_0.*n=[1-8],|type_a=q8_0.*n=8," | head -20
