#!/bin/bash
# round 5, session 2: where the one-launch step's prologue spends its time — timing-only ablations (CDNA4_FQ_ABL: 4 no grid barrier, 8 no quantizer, 12 neither)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out/s2; rm -f gpurun_out/s2/*
for rep in 1 2; do
  for abl in 0 16 4 8 12; do
    AB_TAG=fq_abl_$abl CDNA4_FQ_ABL=$abl timeout 300 python scripts/step_ab.py 4096x4096x512 4096x11008x512 >> gpurun_out/s2/step_ab.txt 2>> gpurun_out/s2/step_ab.err
  done
  AB_TAG=two_launches CDNA4_NO_FUSEQ=1 timeout 300 python scripts/step_ab.py 4096x4096x512 4096x11008x512 >> gpurun_out/s2/step_ab.txt 2>> gpurun_out/s2/step_ab.err
done
cat gpurun_out/s2/step_ab.txt; tail -3 gpurun_out/s2/step_ab.err
