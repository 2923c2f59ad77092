#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 120 tools/microbench/l2_stream > gpurun_out/l2_stream.csv 2>&1; cat gpurun_out/l2_stream.csv
