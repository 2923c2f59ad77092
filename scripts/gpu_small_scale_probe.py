"""Probe: MUL_MAT on weights quantized from SMALL values (the fp16 block scales of Q6_K / Q4_K fall into fp16's subnormal range below ~6e-5 — where real LLM tensors live:
weights of ~0.02 give Q6_K d = max / 32 / 128 ~ 1e-5) against the oracle, per format x weight scale x batch regime.  One JSON line per case."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import refutil as R  # noqa: E402


def main():
    from ggml_amd import ops
    types = {"q6_K": R.Q6_K, "q4_K": R.Q4_K, "q5_K": R.Q5_K, "q4_0": R.Q4_0, "q8_0": R.Q8_0}
    shapes = [(1024, 1024, 200), (256, 1024, 200), (512, 2048, 1), (512, 2048, 16), (4096, 4096, 512)]
    for name, t in types.items():
        for scale in (1.0, 0.0625, 0.004):
            for (m, k, b) in shapes:
                if (m, k, b) == (4096, 4096, 512) and scale == 1.0:
                    continue
                rng = np.random.default_rng(m + b)
                w = R.r_quantize(t, (rng.uniform(-1, 1, (m, k)) * scale).astype(np.float32))
                x = rng.uniform(-1, 1, (b, k)).astype(np.float32)
                a = ops.QTensor.from_host_bytes(t, k, m, w, device="cuda:0")
                y = ops.mul_mat(a, torch.from_numpy(x).cuda()).cpu().numpy()
                ref = R.o_mul_mat(t, w, x, m, k)
                print(json.dumps({"type": name, "scale": scale, "shape": [m, k, b], "rel_l2": R.rel_l2(y, ref), "norm_ratio": float(np.linalg.norm(y) / max(np.linalg.norm(ref), 1e-30))}), flush=True)


if __name__ == "__main__":
    main()
