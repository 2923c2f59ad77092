#!/bin/bash
# every -m gpu test WITHOUT -x (one failure must not hide the rest), per-test results kept: gpurun_out/pytest_gpu.log (-rA lines) + junit xml
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; rm -f gpurun_out/gpt2_parity.jsonl gpurun_out/parity_report.jsonl gpurun_out/split_report.jsonl
timeout -k 10 ${GPU_TESTS_TIMEOUT:-1800} python -m pytest tests -m gpu -q --tb=short -rA -p no:cacheprovider --junitxml=gpurun_out/pytest_gpu.xml "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log | head -40; tail -3 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
