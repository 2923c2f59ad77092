#!/bin/bash
# round 5, session 4: the whole -m gpu suite on the current build, then the decode configurations of Q4_0 / Q8_0 (A/B) and bench.py
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out/s4; rm -f gpurun_out/s4/*
timeout -k 10 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/s4/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?" > gpurun_out/s4/summary.txt; tail -8 gpurun_out/s4/pytest_gpu.log >> gpurun_out/s4/summary.txt
CFGS=default,4 timeout 400 python scripts/gpu_decode_cfg.py 2,8 > gpurun_out/s4/decode_cfg_q40_q80.txt 2>&1
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/s4/bench.log 2> gpurun_out/s4/bench.err; echo "bench rc=$?" >> gpurun_out/s4/summary.txt
cat gpurun_out/s4/summary.txt; cat gpurun_out/s4/decode_cfg_q40_q80.txt; tail -c 3000 gpurun_out/s4/bench.log
