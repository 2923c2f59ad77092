#!/bin/bash
# round 4, session 11: grouped MUL_MAT_ID with the ticketed K split (A/B), its parity tests, K-split buffer type with the RCCL counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s11; mkdir -p $O
for sk in 1 0; do
CDNA4_MOE_SPLITK=$sk timeout 300 python - >> $O/moe_ab.txt 2>> $O/moe_ab.err <<PY
import json, os, torch, bench as B
dev = torch.device("cuda", 0)
from ggml_amd import native; native.lib()
r = B.moe_row(dev, 100)
print("CDNA4_MOE_SPLITK=%s" % os.environ.get("CDNA4_MOE_SPLITK"), json.dumps(r["prefill_512_tokens"]))
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_widening.py -q -m gpu --tb=short -k "mul_mat_id or moe or grouped" > $O/t_moe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_sched.py -q -m gpu --tb=short -k "ksplit" > $O/t_ksplit.log 2>&1
cat $O/moe_ab.txt | cut -c1-400; tail -4 $O/t_moe.log; tail -4 $O/t_ksplit.log; tail -3 $O/moe_ab.err
