#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s18; mkdir -p $O
timeout 600 python scripts/gpu_r4_s18.py > $O/q5k_r8.txt 2> $O/err.txt
cat $O/q5k_r8.txt | cut -c1-300; tail -3 $O/err.txt
