#!/bin/bash
# round 4, session 8: whole -m gpu suite after the r8 route + tails in the 128x128-tile kernels; full bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s8; mkdir -p $O
timeout -k 10 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/summary.txt
timeout -k 10 700 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -15 $O/pytest_gpu.log; tail -c 1500 $O/bench.log
