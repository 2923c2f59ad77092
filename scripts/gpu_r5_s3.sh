#!/bin/bash
# round 5, session 3: two-level grid barrier vs the one-word barrier (CDNA4_FQ_ABL=16), alternating; every process bounded
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out/s3; rm -f gpurun_out/s3/*
for rep in 1 2 3; do
  for abl in 0 16; do
    AB_TAG=fq_abl_$abl CDNA4_FQ_ABL=$abl timeout 60 python scripts/step_ab.py 4096x4096x512 4096x11008x512 8192x4096x512 >> gpurun_out/s3/step_ab.txt 2>> gpurun_out/s3/step_ab.err
  done
done
AB_TAG=two_launches CDNA4_NO_FUSEQ=1 timeout 60 python scripts/step_ab.py 4096x4096x512 4096x11008x512 8192x4096x512 >> gpurun_out/s3/step_ab.txt 2>> gpurun_out/s3/step_ab.err
cat gpurun_out/s3/step_ab.txt
