"""grouped MUL_MAT_ID prefill (bench.moe_row's shape: 8 experts x 2 used x 512 tokens x 4096^2 Q4_K): N calls on one workspace, every output hashed — the launch must be
bit-reproducible (the planner's ranking is a stable sort of the ids; the cut tiles are summed in span order) — and a sample of the rows against the oracle"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import refutil as R
from ggml_amd import native, ops
L = native.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_expert, n_used, n_tok, m, k = 8, 2, 512, 4096, 4096
rng = np.random.default_rng(7)
w = R.random_weights(R.Q4_K, n_expert * m, k, seed=5)
a = ops.QTensor.from_host_bytes(R.Q4_K, k, n_expert * m, w, device="cuda:0")
xb = rng.uniform(-1, 1, (n_tok, n_used, k)).astype(np.float32)
ids = np.stack([rng.permutation(n_expert)[:n_used] for _ in range(n_tok)]).astype(np.int32)
xd, idd = torch.from_numpy(xb).cuda(), torch.from_numpy(ids).cuda()
hashes = set()
for i in range(n):
    y = ops.mul_mat_id(a, xd, idd, n_expert=n_expert)
    hashes.add(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())
sel = rng.choice(n_tok, 16, replace=False)
e = R.rel_l2(y.cpu().numpy()[sel], R.o_mul_mat_id(R.Q4_K, w, xb[sel], ids[sel], m, k, n_expert))
print(json.dumps({"tag": os.environ.get("AB_TAG", ""), "calls": n, "distinct_outputs": len(hashes), "rel_l2_vs_oracle_sample": e}), flush=True)
