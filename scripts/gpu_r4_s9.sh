#!/bin/bash
# round 4, session 9: decode A/B (register form vs DMA form), C5 leg with the same-box t64 number, the re-aimed auto-route test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s9; mkdir -p $O
timeout 900 python scripts/gpu_decode_ab.py > $O/decode_ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "auto_picks or decode or fused_decode or gemv" > $O/t_parity.log 2>&1
timeout 300 python - > $O/c5_ab.json 2> $O/c5_ab.err <<'PY'
import json, torch, bench as B
dev = torch.device("cuda", 0)
from ggml_amd import native; native.lib()
print(json.dumps({"c5": B.shape_row(dev, B.Q4_K, 32768, 8192, 512, 50), "m16k8k": B.shape_row(dev, B.Q4_K, 16384, 8192, 512, 50)}))
PY
tail -3 $O/t_parity.log; head -12 $O/decode_ab.txt | cut -c1-250; cat $O/c5_ab.json | cut -c1-1500; tail -3 $O/c5_ab.err
