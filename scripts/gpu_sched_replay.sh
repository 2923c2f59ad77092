#!/bin/bash
# HIP-graph replay through the plug-in, without python: oracle/_ref/split_harness for three formats; prints the harness report
# and the plug-in's capture / replay counters (tests/test_gpu_sched.py asserts [4, 14] and [2, 5])
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for t in q4_K q4_0 q6_K; do
  GGML_CDNA4_STATS=1 timeout 120 oracle/_ref/split_harness ggml_amd/lib/libggml-cdna4.so $t 512 512 16 2>&1 | grep -E "captures|graph_replay|differs|fail" | cut -c1-400
done | tee gpurun_out/sched_replay.txt
