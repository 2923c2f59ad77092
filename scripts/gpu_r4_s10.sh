#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s10; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sched.py -q -m gpu --tb=short > $O/t_sched.log 2>&1
timeout 900 python -m pytest tests/test_gpu_cabi_ops.py -q -m gpu --tb=short -k "fused" > $O/t_fused.log 2>&1
tail -15 $O/t_sched.log; tail -5 $O/t_fused.log; grep -h ksplit gpurun_out/split_report.jsonl | tail -6 | cut -c1-300
