#!/bin/bash
# a short session (rewritten per use): here — per-stage stamps of k_gemm_kq_t64's 128-row tile at C3 and at 8192 x 4096 x 512 (measurement build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp GGML_CDNA4_OWNED_DEVICE=1
O=$PWD/gpurun_out/session; mkdir -p $O; rm -rf $O/*
cd tools/microbench
for shape in "4096 11008 512" "8192 8192 512"; do
  GB_VARIANTS="24583" GB_SPLITKS="2" GB_ROUNDS=3 GB_TRACE_REPS=30 GB_TRACE_SPLITK=2 timeout 200 ./gemm_bench_abl $shape 16801799 >> $O/t64_128_stage_trace.txt 2>&1
done
cut -c1-200 $O/t64_128_stage_trace.txt
