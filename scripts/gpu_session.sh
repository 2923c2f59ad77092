#!/bin/bash
# a short A/B session (rewritten per use; the numbers it produced are under profiles/rNN/): here — the MoE FFN with / without the shared front, the tests that go with it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/session; mkdir -p $O; rm -rf $O/*
H=oracle/_ref/split_harness; P=ggml_amd/lib/libggml-cdna4.so
timeout 900 python -m pytest tests/test_gpu_moe_front.py tests/test_gpu_group.py tests/test_gpu_resident.py tests/test_gpu_act_share.py -q -m gpu -p no:cacheprovider -k "moe or group or host_ptr or stack" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)" >> $O/summary.txt
for rep in 1 2 3; do
  echo "shared   $(HARNESS_NO_CPU=1 timeout 200 $H $P q4_K 4096 14336 512 moeffn 2>/dev/null | tail -1)" >> $O/moeffn_ab.txt
  echo "separate $(GGML_CDNA4_NO_ACT_SHARE=1 HARNESS_NO_CPU=1 timeout 200 $H $P q4_K 4096 14336 512 moeffn 2>/dev/null | tail -1)" >> $O/moeffn_ab.txt
done
timeout 300 python bench.py --leg mul_mat_id > $O/bench_moe.txt 2>&1
cat $O/summary.txt; cut -c1-250 $O/moeffn_ab.txt; tail -3 $O/bench_moe.txt | cut -c1-900; grep -i "failed\|error" $O/pytest.log | head
