#!/bin/bash
# round 2, GPU call 2: full GPU suite on the new defaults (reader-reset flags, t64 on huge grids, C-ABI op tests) + bench line
mkdir -p gpurun_out; cd "$(dirname "$0")/.."
export PYTHONPATH=$PWD
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 ) > gpurun_out/r2_pytest_gpu.txt
( timeout 400 python bench.py --no-diagnostics 2>&1 | tail -3 ) > gpurun_out/r2_bench.txt
tail -25 gpurun_out/r2_pytest_gpu.txt; cut -c1-1500 gpurun_out/r2_bench.txt
