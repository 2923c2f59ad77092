#!/bin/bash
# round 6, session 11: one-row products of one src1 as ONE launch (plug-in, llama-sized layer front at one row, HIP-graph replay), C3 on k_gemm_r8's split forms,
# and the plug-in tests the new graph walk touches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/r6s11; mkdir -p $O; rm -rf $O/*
H=oracle/_ref/split_harness; P=ggml_amd/lib/libggml-cdna4.so
for rep in 1 2 3; do
  for t in q4_K q4_0; do
    echo "grouped $t $(HARNESS_NO_CPU=1 timeout 120 $H $P $t 4096 14336 1 shared 2>/dev/null | tail -1)" >> $O/group_ab.txt
    echo "node_by_node $t $(GGML_CDNA4_NO_GROUP=1 HARNESS_NO_CPU=1 timeout 120 $H $P $t 4096 14336 1 shared 2>/dev/null | tail -1)" >> $O/group_ab.txt
  done
done
timeout 300 $H $P q4_K 4096 14336 1 shared > $O/group_full_check.txt 2>&1
( cd tools/microbench && GB_VARIANTS="0,335544320" GB_SPLITKS="0,1,2,4,8" GB_ROUNDS=4 timeout 300 ./gemm_bench 4096 11008 512 ) > $O/c3_r8_splits.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_act_share.py tests/test_gpu_sched.py tests/test_gpu_gpt2.py -q -m gpu -p no:cacheprovider > $O/pytest_plugin.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/group_ab.txt | cut -c1-260; tail -2 $O/group_full_check.txt | cut -c1-300; cat $O/c3_r8_splits.txt | grep -v "^$" | head -20; tail -5 $O/pytest_plugin.log
