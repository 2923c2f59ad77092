#!/bin/bash
# round 6, session 9: small batches — the K-sliced one-launch int8 matrix-core kernel (k_mmq_ks_q4_K) and the 16-wave forms of the 2- / 4-row GEMV against round 5's routes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/r6s9; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "small_batches or batch" -x -p no:cacheprovider > $O/pytest_small.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
for rep in 1 2; do
  AB_TAG=r5_routes CDNA4_MMQ_KS_MODE=0 CDNA4_FUSED_NB_CFG=0 timeout 300 python scripts/batch_q4k.py >> $O/batch.txt 2>> $O/batch.err
  AB_TAG=auto timeout 300 python scripts/batch_q4k.py >> $O/batch.txt 2>> $O/batch.err
  AB_TAG=ks_everywhere CDNA4_MMQ_KS_MODE=2 timeout 300 python scripts/batch_q4k.py >> $O/batch.txt 2>> $O/batch.err
  AB_TAG=ks_off_nb16 CDNA4_MMQ_KS_MODE=0 timeout 300 python scripts/batch_q4k.py >> $O/batch.txt 2>> $O/batch.err
done
for ks in 1 2 3 4; do AB_TAG=ks_everywhere_KS$ks CDNA4_MMQ_KS=$ks CDNA4_MMQ_KS_MODE=2 timeout 300 python scripts/batch_q4k.py >> $O/batch.txt 2>> $O/batch.err; done
cat $O/summary.txt; tail -3 $O/pytest_small.log; cat $O/batch.txt
