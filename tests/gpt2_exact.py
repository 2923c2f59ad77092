"""An fp64 forward pass of the gpt-2 graph the reference builds (examples/gpt-2/main-backend.cpp:444-719) on the DEQUANTIZED weights of
a legacy-format model file (examples/gpt-2/quantize.cpp output): the neutral yardstick for the logits of two backends that both
round on the way (the CPU backend to Q8_0 activations and an fp16 GELU table, ours to fp16 activations in the MFMA path).
Test infrastructure only."""
import struct

import numpy as np

import refutil as R


def load_model(path):
    """-> (hparams dict, {name: float64 array in numpy order [rows][cols]})"""
    with open(path, "rb") as f:
        magic, = struct.unpack("i", f.read(4))
        assert magic == 0x67676d6c
        n_vocab, n_ctx, n_embd, n_head, n_layer, ftype = struct.unpack("6i", f.read(24))
        nv, = struct.unpack("i", f.read(4))
        for _ in range(nv):
            ln, = struct.unpack("i", f.read(4)); f.read(ln)
        tensors = {}
        while True:
            h = f.read(12)
            if len(h) < 12:
                break
            n_dims, name_len, ttype = struct.unpack("3i", h)
            ne = list(struct.unpack("%di" % n_dims, f.read(4 * n_dims)))            # ne[0] fastest
            name = f.read(name_len).decode()
            n = int(np.prod(ne))
            if ttype == 0:
                a = np.frombuffer(f.read(4 * n), np.float32).astype(np.float64)
            elif ttype == 1:
                a = np.frombuffer(f.read(2 * n), np.float16).astype(np.float64)
            elif ttype == 2:                                                         # q4_0: the reference's own to_float (oracle == _ref, tests/test_oracle_vs_ref.py)
                raw = np.frombuffer(f.read(n // 32 * 18), np.uint8)
                a = R.o_dequantize(R.Q4_0, raw, ne[0]).astype(np.float64).reshape(-1)
            else:
                raise ValueError("tensor type %d" % ttype)
            tensors[name] = a.reshape(list(reversed(ne)))
    return dict(n_vocab=n_vocab, n_ctx=n_ctx, n_embd=n_embd, n_head=n_head, n_layer=n_layer), tensors


def _norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def _gelu(x):                                                                        # ggml_gelu_f32's closed form (ggml-cpu.c:1753-1755) in fp64
    return 0.5 * x * (1.0 + np.tanh(0.79788456080286535587989211986876 * x * (1.0 + 0.044715 * x * x)))


def forward(hp, T, tokens):
    """logits [n_vocab] of the LAST token of `tokens`, everything in fp64"""
    E, H = hp["n_embd"], hp["n_head"]
    n = len(tokens)
    x = T["model/wte"][tokens] + T["model/wpe"][np.arange(n)]
    mask = np.triu(np.ones((n, n), bool), 1)
    for l in range(hp["n_layer"]):
        p = "model/h%d/" % l
        cur = _norm(x, T[p + "ln_1/g"], T[p + "ln_1/b"])
        qkv = cur @ T[p + "attn/c_attn/w"].T + T[p + "attn/c_attn/b"]
        q, k, v = (qkv[:, i * E:(i + 1) * E].reshape(n, H, E // H).transpose(1, 0, 2) for i in range(3))
        s = q @ k.transpose(0, 2, 1) / np.sqrt(E // H)
        s[:, mask] = -np.inf
        s = np.exp(s - s.max(-1, keepdims=True)); s /= s.sum(-1, keepdims=True)
        a = (s @ v).transpose(1, 0, 2).reshape(n, E)
        x = a @ T[p + "attn/c_proj/w"].T + T[p + "attn/c_proj/b"] + x
        cur = _norm(x, T[p + "ln_2/g"], T[p + "ln_2/b"])
        cur = _gelu(cur @ T[p + "mlp/c_fc/w"].T + T[p + "mlp/c_fc/b"])
        x = cur @ T[p + "mlp/c_proj/w"].T + T[p + "mlp/c_proj/b"] + x
    x = _norm(x[-1:], T["model/ln_f/g"], T["model/ln_f/b"])
    return (x @ T.get("model/lm_head", T["model/wte"]).T)[0]


def harness_tokens(n_vocab, n):
    """the fixed LCG token stream of oracle/gpt2_harness.cpp"""
    st, out = 12345, []
    for _ in range(n):
        st = (st * 1664525 + 1013904223) & 0xFFFFFFFF
        out.append((st >> 8) % n_vocab)
    return np.array(out, np.int64)
