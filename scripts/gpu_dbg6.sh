#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1
tail -8 gpurun_out/gpu_tests.log
timeout 300 python scripts/gpu_dbg5.py > gpurun_out/dbg5.log 2>&1; tail -3 gpurun_out/dbg5.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_default.log 2>&1; tail -c 6000 gpurun_out/bench_default.log
