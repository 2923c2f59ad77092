#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cabi_ops.py tests/test_gpu_gpt2.py tests/test_gpu_backend_plugin.py tests/test_gpu_sched.py -m gpu -q -x > gpurun_out/fuse_tests.log 2>&1
tail -15 gpurun_out/fuse_tests.log; tail -1 gpurun_out/gpt2_parity.jsonl | cut -c1-600
