"""Generates tests/golden/small.gguf, small_align64.gguf and gguf_expected.json with the UNMODIFIED reference
(oracle/_ref/libggml-base.so, built by oracle/ref.mk from /root/reference): small.gguf is written by the reference's GGUF
writer (gguf_set_val_* / gguf_add_tensor / gguf_write_to_file, src/gguf.cpp:917-1320), small_align64.gguf (general.alignment
= 64, which that writer cannot produce) by tests/ggufref.py's serializer, and the expectations for BOTH are what the
reference's READER (gguf_init_from_file + getters, include/gguf.h:80-126) reports for them.  Run in the build container:

    make -C oracle && python tests/golden/make_gguf_golden.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refutil as R  # noqa: E402
import ggufref as G  # noqa: E402

assert R.have_ref(), "build oracle/_ref first (make -C oracle)"


def build(path, alignment=None):
    kv, tensors = G.fixture_content(alignment)
    if alignment is None:
        G.write_with_reference(path, kv, tensors)
    else:   # the reference's writer ignores general.alignment (src/gguf.cpp:1102,1289 use the default): serialize by hand
        open(path, "wb").write(G.py_serialize(kv, tensors, alignment=alignment))


build(os.path.join(HERE, "small.gguf"))
build(os.path.join(HERE, "small_align64.gguf"), alignment=64)
exp = {name: G.read_with_reference(os.path.join(HERE, name)) for name in ("small.gguf", "small_align64.gguf")}
# what the reference's reader says about every malformed file of ggufref.malformed_cases(): metadata only (no_alloc) and
# with the tensor data loaded; null = the reference process died on it (GGML_ASSERT / division by zero)
import tempfile
with tempfile.TemporaryDirectory() as d:
    paths = []
    for label, data in G.malformed_cases():
        paths.append(os.path.join(d, label + ".gguf"))
        open(paths[-1], "wb").write(data)
    meta = G.reference_accepts_many(paths, with_data=False)
    full = G.reference_accepts_many(paths, with_data=True)
exp["malformed"] = {label: [m, f] for (label, _), m, f in zip(G.malformed_cases(), meta, full)}
json.dump(exp, open(os.path.join(HERE, "gguf_expected.json"), "w"), indent=1, sort_keys=True)
print({k: (len(v["kv"]), len(v["tensors"]), v["data_offset"], v["data_size"]) for k, v in exp.items() if k != "malformed"})
print({k: v for k, v in exp["malformed"].items() if not k.startswith("cut_")})
