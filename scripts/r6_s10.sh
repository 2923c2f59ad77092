#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/r6s10; mkdir -p $O; rm -rf $O/*
for rep in 1 2; do
  AB_TAG=ks_off CDNA4_MMQ_KS_MODE=0 timeout 300 python scripts/batch_q4k.py >> $O/batch.txt 2>> $O/batch.err
  AB_TAG=ks_everywhere CDNA4_MMQ_KS_MODE=2 timeout 300 python scripts/batch_q4k.py >> $O/batch.txt 2>> $O/batch.err
done
for ks in 2 3; do AB_TAG=ks_everywhere_KS$ks CDNA4_MMQ_KS=$ks CDNA4_MMQ_KS_MODE=2 timeout 300 python scripts/batch_q4k.py >> $O/batch.txt 2>> $O/batch.err; done
cat $O/batch.txt
