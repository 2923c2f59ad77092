#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s17; mkdir -p $O
timeout 1500 python scripts/gpu_batch_sweep.py > $O/batch_sweep.txt 2> $O/err.txt
head -c 9000 $O/batch_sweep.txt; tail -3 $O/err.txt
