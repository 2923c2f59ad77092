#!/bin/bash
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cabi_ops.py -q -m gpu -x -k "byte_exact and 2-0-uniform" 2>&1 | grep -v "^E   *+" | tail -30 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_sched.py -q -m gpu 2>&1 | tail -5 | cut -c1-400
