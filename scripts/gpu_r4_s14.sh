#!/bin/bash
# round 4, session 14: grouped MUL_MAT_ID on 128- vs 256-row tiles; t64 tiles at 3/4 of a tile per CU
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s14; mkdir -p $O
for tm in 128 256 128 256; do
CDNA4_MOE_TM=$tm timeout 300 python - >> $O/moe_tm_ab.txt 2>> $O/moe.err <<PY
import json, os, torch, bench as B
dev = torch.device("cuda", 0)
from ggml_amd import native; native.lib()
r = B.moe_row(dev, 100)
print("CDNA4_MOE_TM=%s" % os.environ.get("CDNA4_MOE_TM"), r["prefill_512_tokens"]["us_per_call"])
PY
done
( cd tools/microbench
  for shape in "12288 4096 512" "12288 8192 512" "6144 4096 1024" "24576 4096 512"; do GB_SPLITKS=0 GB_VARIANTS="24583,40967" GB_ROUNDS=5 timeout 180 ./gemm_bench $shape ""; done
) 2>&1 | grep -E "^M=|^variant" > $O/t64_tiles_075.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short -k "mul_mat_id_grouped" > $O/t_moe.log 2>&1
cat $O/moe_tm_ab.txt; cat $O/t64_tiles_075.txt | cut -c1-110; tail -3 $O/t_moe.log
