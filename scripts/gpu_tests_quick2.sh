#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backend_plugin.py tests/test_gpu_gpt2.py tests/test_gpu_sched.py -m gpu -q -x -p no:cacheprovider > gpurun_out/quick_tests.log 2>&1
echo "rc=$?"; tail -6 gpurun_out/quick_tests.log
