#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sched.py -m gpu -q -x -k "replayed" > gpurun_out/graph_tests.log 2>&1
tail -4 gpurun_out/graph_tests.log; tail -1 gpurun_out/split_report.jsonl | cut -c300-600
