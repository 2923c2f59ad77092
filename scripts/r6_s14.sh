#!/bin/bash
# round 6, session 14: the headline's fixed costs — block 0's milestones in the one-launch step and in the GEMM on prepared activations (measurement build of the library)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp GGML_CDNA4_OWNED_DEVICE=1
O=$PWD/gpurun_out/r6s14; mkdir -p $O; rm -rf $O/*
cd tools/microbench
GB_VARIANTS="0" GB_SPLITKS="0" GB_ROUNDS=3 GB_TRACE_REPS=30 GB_TRACE_SPLITK=0 GB_TRACE_FQ=1 timeout 300 ./gemm_bench_abl 4096 4096 512 > $O/headline_milestones.txt 2>&1
cat $O/headline_milestones.txt | cut -c1-260
