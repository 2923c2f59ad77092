#!/bin/bash
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "mul_mat_id" 2>&1 | tail -25 ) > gpurun_out/r2_moe_tests.txt
( BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r2_bench_dist1.out 2> gpurun_out/r2_bench_dist1.err )
cut -c1-300 gpurun_out/r2_moe_tests.txt; tail -5 gpurun_out/r2_bench_dist1.err | cut -c1-400; grep '"metric"' gpurun_out/r2_bench_dist1.out | cut -c1-4000
