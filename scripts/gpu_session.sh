#!/bin/bash
# a short session (rewritten per use): here — k_mmq_q4_K with the weights prefetched three superblocks ahead and the minimum term on the matrix core, against the previous
# kernel (a twin library built from the previous source), alternating, device time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/session; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "small_batches or mul_mat_id" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)" >> $O/summary.txt
for rep in 1 2; do
  AB_TAG=new BATCH_ROWS=1,3,4,8,16,24,32,48,64 timeout 200 python scripts/batch_q4k.py 2>/dev/null | tail -1 >> $O/mmq_ab.txt
  AB_TAG=prev CDNA4_KERNELS_LIB=$R/tools/microbench/ab/libcdna4_kernels_prev_mmq.so BATCH_ROWS=1,3,4,8,16,24,32,48,64 timeout 200 python scripts/batch_q4k.py 2>/dev/null | tail -1 >> $O/mmq_ab.txt
done
cat $O/summary.txt; cat $O/mmq_ab.txt
