#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gpt2_parity.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gpt2.py tests/test_gpu_cabi_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "int8_matrix or gpt2 or fused" > gpurun_out/s12_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/s12_pytest.log | head -20; tail -2 gpurun_out/s12_pytest.log
cat gpurun_out/gpt2_parity.jsonl 2>/dev/null | cut -c1-400 | head -20
