#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/gpu_dbg5.py > gpurun_out/dbg5.log 2>&1; echo "dbg5 rc $?" >> gpurun_out/dbg5.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "64x128 or huge_grids or mul_mat_id" > gpurun_out/t64_tests.log 2>&1
tail -5 gpurun_out/t64_tests.log
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/bench_dist.log 2>&1
tail -5 gpurun_out/dbg5.log; tail -c 3000 gpurun_out/bench_dist.log
