#!/bin/bash
# a short session (rewritten per use): here — k_mmq_q4_K with sixteen waves per work-group (CDNA4_MMQ_NW=16) against eight, alternating, device time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1 GGML_CDNA4_OWNED_DEVICE=1
R=$PWD; O=$R/gpurun_out/session; mkdir -p $O; rm -rf $O/*
CDNA4_MMQ_NW=16 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "small_batches and q4_K" > $O/pytest.log 2>&1; echo "pytest (16 waves) rc=$? $(tail -1 $O/pytest.log)" >> $O/summary.txt
for rep in 1 2; do
  AB_TAG=nw16 CDNA4_MMQ_NW=16 BATCH_ROWS=1,3,4,8,12,16 timeout 200 python scripts/batch_q4k.py 2>/dev/null | tail -1 >> $O/mmq_ab.txt
  AB_TAG=nw8 BATCH_ROWS=1,3,4,8,12,16 timeout 200 python scripts/batch_q4k.py 2>/dev/null | tail -1 >> $O/mmq_ab.txt
done
cat $O/summary.txt; cat $O/mmq_ab.txt
