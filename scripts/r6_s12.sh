#!/bin/bash
# round 6, session 12: the reference-order mode on K-quants (bit-for-bit tests, layer front, gpt-2), and C3 on k_gemm_r8's split forms vs k_gemm_kq_t64
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/r6s12; mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests/test_gpu_exact.py tests/test_gpu_act_share.py tests/test_gpu_gpt2.py -q -m gpu -p no:cacheprovider > $O/pytest_exact.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
( cd tools/microbench
  GB_VARIANTS="0" GB_SPLITKS="0" GB_ROUNDS=4 timeout 200 ./gemm_bench 4096 11008 512
  GB_VARIANTS="8192,335544320" GB_SPLITKS="1,2" GB_ROUNDS=4 timeout 200 ./gemm_bench 4096 11008 512
  GB_VARIANTS="335544320" GB_SPLITKS="4,8" GB_ROUNDS=4 timeout 200 ./gemm_bench 4096 11008 512 ) > $O/c3_r8_splits.txt 2>&1
cat $O/summary.txt; grep "us/call\|failed" $O/c3_r8_splits.txt | head -20; tail -8 $O/pytest_exact.log; cp gpurun_out/split_report.jsonl gpurun_out/gpt2_parity.jsonl $O/ 2>/dev/null
