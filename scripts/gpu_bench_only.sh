#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-1200
