#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_dist1.log 2>&1
echo "dist1 rc=$?"; tail -3 gpurun_out/bench_dist1.log | cut -c1-1500
