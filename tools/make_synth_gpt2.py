#!/usr/bin/env python
"""Writes a synthetic GPT-2 117M-shaped model in the legacy `ggml` .bin format that examples/gpt-2 reads
(magic 0x67676d6c; layout of examples/gpt-2/convert-ckpt-to-ggml.py:91-154).  No network / checkpoints here, so
weights are random: N(0, 0.02) matrices, LayerNorm gains ~1, small biases — the architecture, shapes and
tensor names are the real ones.  Quantize afterwards with the reference's own gpt-2-quantize.

    python tools/make_synth_gpt2.py out_f32.bin [--layers 12] [--seed 0]
"""
import argparse
import struct
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--n-vocab", type=int, default=50257)
    ap.add_argument("--n-ctx", type=int, default=1024)
    ap.add_argument("--n-embd", type=int, default=768)
    ap.add_argument("--n-head", type=int, default=12)
    ap.add_argument("--wte-std", type=float, default=0.02, help="std of the (tied) embedding / output matrix: 0.08 gives logits of std ~2, a spread like a trained model's")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    E = a.n_embd
    with open(a.out, "wb") as f:
        f.write(struct.pack("i", 0x67676d6c))
        for v in (a.n_vocab, a.n_ctx, E, a.n_head, a.layers, 0):       # ftype 0 = all f32
            f.write(struct.pack("i", v))
        f.write(struct.pack("i", a.n_vocab))
        for i in range(a.n_vocab):                                      # placeholder vocabulary (unique byte strings)
            w = ("t%d" % i).encode()
            f.write(struct.pack("i", len(w))); f.write(w)

        def tensor(name, arr):
            arr = np.ascontiguousarray(arr, np.float32)
            nb = name.encode()
            f.write(struct.pack("iii", arr.ndim, len(nb), 0))
            for d in range(arr.ndim):
                f.write(struct.pack("i", arr.shape[arr.ndim - 1 - d]))
            f.write(nb)
            arr.tofile(f)

        def mat(rows, cols, std=0.02):
            return (rng.standard_normal((rows, cols)) * std).astype(np.float32)

        def vec(n, mean=0.0, std=0.02):
            return (mean + rng.standard_normal(n) * std).astype(np.float32)

        tensor("model/wte", mat(a.n_vocab, E, a.wte_std))
        tensor("model/wpe", mat(a.n_ctx, E, 0.01))
        for l in range(a.layers):
            p = "model/h%d/" % l
            tensor(p + "ln_1/g", vec(E, 1.0)); tensor(p + "ln_1/b", vec(E))
            tensor(p + "attn/c_attn/w", mat(3 * E, E)); tensor(p + "attn/c_attn/b", vec(3 * E))      # stored transposed: [out][in]
            tensor(p + "attn/c_proj/w", mat(E, E)); tensor(p + "attn/c_proj/b", vec(E))
            tensor(p + "ln_2/g", vec(E, 1.0)); tensor(p + "ln_2/b", vec(E))
            tensor(p + "mlp/c_fc/w", mat(4 * E, E)); tensor(p + "mlp/c_fc/b", vec(4 * E))
            tensor(p + "mlp/c_proj/w", mat(E, 4 * E)); tensor(p + "mlp/c_proj/b", vec(E))
        tensor("model/ln_f/g", vec(E, 1.0)); tensor("model/ln_f/b", vec(E))
    print("wrote", a.out)


if __name__ == "__main__":
    main()
