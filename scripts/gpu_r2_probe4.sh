#!/bin/bash
# round 2, GPU call 4: the new bench.py (default, forced-dist, c5) + the C-ABI op tests after the signed-zero fixes
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
( timeout 600 python -m pytest tests/test_gpu_cabi_ops.py -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r2_cabi.txt
( timeout 900 python bench.py 2>&1 | tail -4 ) > gpurun_out/r2_bench_default.txt
( BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/r2_bench_dist1.txt
( timeout 600 python bench.py --config c5 --steps 30 --warmup 5 2>&1 | tail -3 ) > gpurun_out/r2_bench_c5.txt
cat gpurun_out/r2_cabi.txt; cut -c1-6000 gpurun_out/r2_bench_default.txt; cut -c1-3000 gpurun_out/r2_bench_dist1.txt; cut -c1-2000 gpurun_out/r2_bench_c5.txt
