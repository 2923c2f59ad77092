#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1
tail -8 gpurun_out/gpu_tests.log
