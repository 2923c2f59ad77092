#!/bin/bash
# a short session (rewritten per use; the numbers it produced are under profiles/rNN/): here — the default mode beside a tenant, more routes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
O=$PWD/gpurun_out/session; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_gpu_shared_device.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)" >> $O/summary.txt
cat $O/summary.txt; grep -i "failed\|error\|assert" $O/pytest.log | head; grep "more_routes" gpurun_out/parity_report.jsonl | tail -1 | cut -c1-900
