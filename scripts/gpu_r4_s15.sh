#!/bin/bash
# round 4, session 15: the 128x128-tile kernels' split in two — hand-off vs ticketed (formats leg, one process per setting, alternating), their parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s15; mkdir -p $O
for h in 1 0 1 0; do
CDNA4_W8_HANDOFF=$h timeout 300 python - >> $O/w8_split_ab.txt 2>> $O/ab.err <<PY
import json, os, torch, bench as B
dev = torch.device("cuda", 0)
from ggml_amd import native; native.lib()
r = B.format_rows(dev, 100)
print("CDNA4_W8_HANDOFF=%s" % os.environ.get("CDNA4_W8_HANDOFF"), {k: (v["gemm_b512_us"], v["step_b512_us"]) for k, v in r.items()})
PY
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi_ops.py tests/test_gpu_widening.py tests/test_gpu_sched.py -q -m gpu --tb=short -k "not gpt2 and not flash" > $O/t.log 2>&1
cat $O/w8_split_ab.txt; tail -4 $O/t.log
