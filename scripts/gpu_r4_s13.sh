#!/bin/bash
# round 4, session 13: hand-off vs ticketed split in two (kernel only, same box); few-rows MoE on the int8 matrix cores vs the padded fp16 grouped GEMM
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/s13; mkdir -p $O
( cd tools/microbench
  for shape in "4096 4096 512" "4096 11008 512" "8192 4096 512" "4096 14336 512" "2048 4096 512"; do
    for sh in 0 1 0 1; do echo "GGML_CDNA4_SHARED_DEVICE=$sh"; GGML_CDNA4_SHARED_DEVICE=$sh GB_SPLITKS=0 GB_VARIANTS="0" GB_ROUNDS=5 timeout 120 ./gemm_bench $shape ""; done
  done ) 2>&1 | grep -E "^GGML|^M=|^variant" > $O/handoff_vs_ticketed.txt
for no in 0 1; do
CDNA4_NO_MMQ_IDS=$no timeout 300 python - >> $O/moe64_ab.txt 2>> $O/moe64.err <<PY
import os, json, numpy as np, torch, sys
sys.path.insert(0, "tests")
import refutil as R
from ggml_amd import ops
out = {}
for (t, name) in ((R.Q4_K, "q4_K"), (R.Q6_K, "q6_K"), (R.Q8_0, "q8_0")):
  for n_tok in (32, 64, 128):
    ne, nu, m, k = 8, 2, 4096, 4096
    rng = np.random.default_rng(1)
    w = R.random_block_bytes(t, ne * m, k, rng) if hasattr(R, "random_block_bytes") else R.random_weights(t, ne * m, k, seed=1)
    a = ops.QTensor.from_host_bytes(t, k, ne * m, w)
    xd = torch.from_numpy(rng.uniform(-1, 1, (n_tok, nu, k)).astype(np.float32)).cuda()
    idd = torch.from_numpy(np.stack([rng.permutation(ne)[:nu] for _ in range(n_tok)]).astype(np.int32)).cuda()
    for _ in range(10): ops.mul_mat_id(a, xd, idd, n_expert=ne)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): ops.mul_mat_id(a, xd, idd, n_expert=ne)
    e1.record(); e1.synchronize()
    out["%s n_tok=%d" % (name, n_tok)] = round(e0.elapsed_time(e1) * 1e3 / 40, 2)
print("CDNA4_NO_MMQ_IDS=%s" % os.environ.get("CDNA4_NO_MMQ_IDS"), json.dumps(out))
PY
done
cat $O/handoff_vs_ticketed.txt | cut -c1-120; cat $O/moe64_ab.txt; tail -3 $O/moe64.err
