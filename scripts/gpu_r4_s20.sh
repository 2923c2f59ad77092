#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out/s20
timeout 900 python scripts/gpu_r4_s20.py > gpurun_out/s20/decode_cfg.txt 2> gpurun_out/s20/err.txt
cat gpurun_out/s20/decode_cfg.txt | cut -c1-260; tail -2 gpurun_out/s20/err.txt
