"""The GGUF reader of libcdna4_kernels.so (include/ggml_cdna4_gguf.h) against the reference's
reader (src/gguf.cpp:319-705): golden files written / judged by the unmodified reference (tests/golden/make_gguf_golden.py),
every acceptance check from both sides on deliberately malformed files, and — when oracle/_ref is present — the reference
run live on freshly generated files.  CPU only: the reader is host code."""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

import ggufref as G
import refutil as R

ROOT = R.ROOT
GOLD = os.path.join(ROOT, "tests", "golden")
EXPECTED = json.load(open(os.path.join(GOLD, "gguf_expected.json")))


@pytest.fixture(scope="module")
def gg():
    import __graft_entry__ as g
    g.build()
    import ggml_amd.gguf as m
    m._lib()
    return m


def test_header_symbols_all_exported(gg):
    import ggml_amd.native as n
    hdr = open(os.path.join(ROOT, "include", "ggml_cdna4_gguf.h")).read()
    declared = set(re.findall(r"\b(ggml_cdna4_gguf_[a-z0-9_]+)\s*\(", hdr))
    assert declared == {s[0] for s in gg.SYMBOLS}, declared ^ {s[0] for s in gg.SYMBOLS}
    out = subprocess.run(["nm", "-D", "--defined-only", n.LIB_PATH], capture_output=True, text=True).stdout
    assert declared <= set(re.findall(r" T (ggml_cdna4_gguf_\w+)", out))


def as_expected(f):
    """a GGUFFile in the JSON shape of ggufref.read_with_reference"""
    kv = []
    for key, val in f.kv.items():
        t, et = f.kv_types[key]
        if t == 9:
            kv.append([key, 9, et, val if et == 8 else val.tolist()])
        else:
            kv.append([key, t, None, val])
    raw = f._raw
    tensors = []
    for t in f.tensors:
        lo = f.data_offset + t.offset
        tensors.append({"name": t.name, "type": t.type, "offset": t.offset, "size": t.size,
                        "sha256": hashlib.sha256(raw[lo:lo + t.size]).hexdigest() if lo + t.size <= len(raw) else None})
    return {"version": f.version, "alignment": f.alignment, "data_offset": f.data_offset, "data_size": f.data_size, "kv": kv, "tensors": tensors}


def open_file(gg, path, require_data=True):
    f = gg.GGUFFile(path, require_data=require_data)
    f._raw = open(path, "rb").read()
    return f


@pytest.mark.parametrize("name", ["small.gguf", "small_align64.gguf"])
def test_golden_file_reads_like_the_reference(gg, name):
    with open_file(gg, os.path.join(GOLD, name)) as f:
        got, exp = as_expected(f), EXPECTED[name]
        assert got["kv"] == exp["kv"]
        assert got["tensors"] == exp["tensors"]
        for k in ("version", "alignment", "data_offset", "data_size"):
            assert got[k] == exp[k], k
        # payload views: zero-copy, read-only, and exactly the bytes in the file
        for t in f.tensors:
            a = f.tensor_bytes(t.name)
            assert a.dtype == np.uint8 and a.size == t.size and not a.flags.writeable
            assert hashlib.sha256(a.tobytes()).hexdigest() == exp["tensors"][f.tensor_id(t.name)]["sha256"]
        assert f.tensors[0].ne == (256, 8, 1, 1) and f.tensors[-1].ne == (2, 3, 2, 2)


def test_payload_is_what_mul_mat_consumes(gg):
    """the Q4_K tensor of the fixture, straight from the mapping, through the CPU oracle's dequantizer and MUL_MAT"""
    with open_file(gg, os.path.join(GOLD, "small.gguf")) as f:
        t = f.tensors[f.tensor_id("blk.0.attn_q.weight")]
        w = np.array(f.tensor_bytes(t.name))
        k, m = t.ne[0], t.ne[1]
        assert w.size == gg._lib().ggml_cdna4_row_size(t.type, k) * m
        deq = R.o_dequantize(t.type, w, k)
        assert np.isfinite(deq).all() and 0.2 < np.abs(deq).mean() < 0.8          # uniform(-1,1) weights
        x = np.random.default_rng(3).uniform(-1, 1, (2, k)).astype(np.float32)
        y = R.o_mul_mat(t.type, w, x, m, k)
        assert R.rel_l2(y, x.astype(np.float64) @ deq.astype(np.float64).T) < 1e-2


def test_type_table(gg):
    L = gg._lib()
    for t, (blck, size) in {0: (1, 4), 1: (1, 2), 2: (32, 18), 8: (32, 34), 12: (256, 144), 13: (256, 176), 14: (256, 210), 15: (256, 292), 4: (0, 0), 39: (0, 0), -1: (0, 0)}.items():
        assert (L.ggml_cdna4_gguf_blck_size(t), L.ggml_cdna4_gguf_type_size(t)) == (blck, size)
    if R.have_ref():
        b = G.base()
        b.ggml_type_size.restype = C.c_size_t; b.ggml_blck_size.restype = C.c_int64
        for t in range(39):
            assert (L.ggml_cdna4_gguf_blck_size(t), L.ggml_cdna4_gguf_type_size(t)) == (b.ggml_blck_size(t), b.ggml_type_size(t)), t


def ours_accepts(gg, path, require_data):
    h = gg._lib().ggml_cdna4_gguf_open(path.encode(), 1 if require_data else 0)
    if h:
        gg._lib().ggml_cdna4_gguf_close(h)
        return True
    assert gg._lib().ggml_cdna4_last_error().startswith(b"gguf: ")
    return False


def test_malformed_files_same_verdict_as_the_reference(gg, tmp_path):
    cases = G.malformed_cases()
    assert [c[0] for c in cases] == list(EXPECTED["malformed"].keys()) or set(c[0] for c in cases) == set(EXPECTED["malformed"])
    bad = []
    for label, data in cases:
        p = str(tmp_path / (label + ".gguf"))
        open(p, "wb").write(data)
        for mode, want in zip((False, True), EXPECTED["malformed"][label]):
            want = bool(want)                      # None = the reference process DIED on this file: we must reject, not die
            if ours_accepts(gg, p, mode) != want:
                bad.append((label, "with data" if mode else "metadata only", "reference accepts" if want else "reference rejects"))
    assert not bad, bad


@pytest.mark.skipif(not R.have_ref(), reason="needs oracle/_ref (the compiled reference)")
def test_recorded_verdicts_are_the_reference_s(tmp_path):
    cases = G.malformed_cases()[:60]
    paths = []
    for label, data in cases:
        paths.append(str(tmp_path / (label + ".gguf")))
        open(paths[-1], "wb").write(data)
    live = G.reference_accepts_many(paths, with_data=False)
    assert live == [EXPECTED["malformed"][label][0] for label, _ in cases]


@pytest.mark.skipif(not R.have_ref(), reason="needs oracle/_ref (the compiled reference)")
def test_random_files_live_against_the_reference(gg, tmp_path):
    """files with random keys, value types, array lengths, tensor shapes and alignments: our reader == the reference's"""
    rng = np.random.default_rng(11)
    scal = ["u8", "i8", "u16", "i16", "u32", "i32", "f32", "u64", "i64", "f64", "bool"]
    lo_hi = {"u8": (0, 255), "i8": (-128, 127), "u16": (0, 65535), "i16": (-32768, 32767), "u32": (0, 2 ** 32 - 1), "i32": (-2 ** 31, 2 ** 31 - 1),
             "u64": (0, 2 ** 63 - 1), "i64": (-2 ** 62, 2 ** 62)}

    def value(kind):
        if kind == "bool":
            return bool(rng.integers(0, 2))
        if kind in ("f32", "f64"):
            return float(np.float32(rng.standard_normal()))
        lo, hi = lo_hi[kind]
        return int(rng.integers(lo, hi, endpoint=True))
    types = [(0, 1, 4), (1, 1, 2), (2, 32, 18), (8, 32, 34), (12, 256, 144), (13, 256, 176), (14, 256, 210), (26, 1, 4)]
    for it in range(12):
        align = int(rng.choice([1, 8, 32, 64, 256]))
        kv = [] if align == 32 else [("general.alignment", "u32", align)]
        for j in range(int(rng.integers(0, 12))):
            kind = str(rng.choice(scal + ["str", "arr:str"] + ["arr:" + s for s in scal if s != "bool"]))
            key = "k%d.%s" % (j, "x" * int(rng.integers(0, 40)))
            if kind == "str":
                kv.append((key, kind, "".join(chr(int(c)) for c in rng.integers(32, 127, int(rng.integers(0, 50))))))
            elif kind == "arr:str":
                kv.append((key, kind, ["s%d" % i * int(rng.integers(0, 4)) for i in range(int(rng.integers(0, 6)))]))
            elif kind.startswith("arr:"):
                kv.append((key, kind, [value(kind[4:]) for _ in range(int(rng.integers(0, 9)))]))
            else:
                kv.append((key, kind, value(kind)))
        tensors = []
        for j in range(int(rng.integers(0, 7))):
            t, blck, size = types[int(rng.integers(0, len(types)))]
            nd = int(rng.integers(1, 5))
            ne = tuple([blck * int(rng.integers(1, 4))] + [int(rng.integers(1, 4)) for _ in range(nd - 1)])
            nbytes = int(np.prod(ne)) // blck * size
            tensors.append(("t%d" % j, t, ne, rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()))
        p = str(tmp_path / ("r%d.gguf" % it))
        open(p, "wb").write(G.py_serialize(kv, tensors, alignment=align))
        exp = G.read_with_reference(p)
        assert exp is not None, (it, kv, [(n, t, ne) for n, t, ne, _ in tensors])
        with open_file(gg, p) as f:
            assert as_expected(f) == exp


@pytest.mark.skipif(not R.have_ref(), reason="needs oracle/_ref (the compiled reference)")
def test_serializer_of_the_tests_equals_the_reference_writer(tmp_path):
    """the hand serializer that builds the malformed / aligned files writes the reference writer's bytes for a valid one"""
    kv, tensors = G.fixture_content()
    assert G.py_serialize(kv, tensors) == open(os.path.join(GOLD, "small.gguf"), "rb").read()


def test_getter_misuse_is_an_error_not_an_abort(gg):
    L = gg._lib()
    h = L.ggml_cdna4_gguf_open(os.path.join(GOLD, "small.gguf").encode(), 1)
    assert h
    try:
        i = L.ggml_cdna4_gguf_find_key(h, b"test.u32")
        out = C.c_uint64(0)
        assert L.ggml_cdna4_gguf_val(h, i, 4, C.byref(out)) == 0 and out.value == 4000000000
        assert L.ggml_cdna4_gguf_val(h, i, 5, C.byref(out)) == -1 and b"does not hold" in L.ggml_cdna4_last_error()
        assert L.ggml_cdna4_gguf_val_str(h, i) is None
        assert L.ggml_cdna4_gguf_arr_data(h, i) is None
        j = L.ggml_cdna4_gguf_find_key(h, b"tokenizer.ggml.tokens")
        assert L.ggml_cdna4_gguf_arr_n(h, j) == 5 and L.ggml_cdna4_gguf_arr_str(h, j, 4) == b"x" * 70
        assert L.ggml_cdna4_gguf_arr_str(h, j, 5) is None
        assert L.ggml_cdna4_gguf_arr_data(h, j) is None                    # string arrays have no packed data
        assert L.ggml_cdna4_gguf_key(h, 999) is None and L.ggml_cdna4_gguf_kv_type(h, -1) == -1
        assert L.ggml_cdna4_gguf_find_key(h, b"nope") == -1 and L.ggml_cdna4_gguf_find_tensor(h, b"nope") == -1
        assert L.ggml_cdna4_gguf_tensor_name(h, 99) is None and L.ggml_cdna4_gguf_tensor_data(h, 99) is None
        assert L.ggml_cdna4_gguf_find_tensor(h, b"output.weight") == 4
    finally:
        L.ggml_cdna4_gguf_close(h)
    assert L.ggml_cdna4_gguf_open(b"/nonexistent/file.gguf", 0) is None and b"failed to open" in L.ggml_cdna4_last_error()


def test_metadata_only_open_of_a_truncated_file(gg, tmp_path):
    raw = open(os.path.join(GOLD, "small.gguf"), "rb").read()
    p = str(tmp_path / "cut.gguf")
    exp = EXPECTED["small.gguf"]
    cut = exp["data_offset"] + exp["tensors"][3]["offset"] + 10           # the file ends inside the 4th tensor
    open(p, "wb").write(raw[:cut])
    with pytest.raises(gg.GGUFError, match="tensor data"):
        gg.GGUFFile(p, require_data=True)
    with gg.GGUFFile(p, require_data=False) as f:
        assert [t.name for t in f.tensors] == [t["name"] for t in exp["tensors"]]
        assert hashlib.sha256(f.tensor_bytes(2).tobytes()).hexdigest() == exp["tensors"][2]["sha256"]
        with pytest.raises(gg.GGUFError, match="ends before"):
            f.tensor_bytes(3)


@pytest.mark.gpu
def test_gguf_weights_through_the_hip_path(gg):
    """GGUF payload -> HBM (GGUFFile.qtensor) -> ggml_cdna4_mul_mat, against the CPU oracle on the same bytes"""
    import torch
    from ggml_amd import ops
    rng = np.random.default_rng(5)
    with gg.GGUFFile(os.path.join(GOLD, "small.gguf")) as f:
        for name in ("blk.0.attn_q.weight", "output.weight"):              # Q4_K [256 x 8], Q6_K [256 x 3]
            t = f.tensors[f.tensor_id(name)]
            k, m = t.ne[0], t.ne[1]
            a = f.qtensor(name, device="cuda")
            x = rng.uniform(-1, 1, (2, k)).astype(np.float32)
            y = ops.mul_mat(a, torch.from_numpy(x).cuda(), path=ops.PATH_GEMV).cpu().numpy()
            yo = R.o_mul_mat(t.type, np.array(f.tensor_bytes(name)), x, m, k)
            assert np.isfinite(y).all() and R.rel_l2(y, yo) < 1e-5, name


@pytest.mark.gpu
def test_gguf_upload_through_pinned_staging(gg, tmp_path):
    """ggml_cdna4_gguf_upload: payloads larger and smaller than the 16-MiB staging chunk arrive in HBM byte for byte; the rate of the 256-MiB
    payload (mapping -> two pinned staging buffers -> HBM) goes to the parity report"""
    import time
    import torch
    rng = np.random.default_rng(9)
    big = rng.integers(0, 256, 40 * (1 << 20) + 4096, dtype=np.uint8)           # 2.5 chunks of I8
    small = rng.integers(0, 256, 144 * 8, dtype=np.uint8)                       # one Q4_K row set
    p = str(tmp_path / "up.gguf")
    open(p, "wb").write(G.py_serialize([], [("big", 24, (big.size,), big.tobytes()), ("w", 12, (256, 8), small.tobytes())]))
    with gg.GGUFFile(p) as f:
        for name, ref in (("big", big), ("w", small)):
            dst = torch.zeros(ref.size, dtype=torch.uint8, device="cuda")
            f.upload(name, dst, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(dst.cpu().numpy(), ref), name
        with pytest.raises(gg.GGUFError, match="too small"):
            f.upload("big", torch.zeros(16, dtype=torch.uint8, device="cuda"))
    huge = rng.integers(0, 256, 256 << 20, dtype=np.uint8)
    p2 = str(tmp_path / "huge.gguf")
    open(p2, "wb").write(G.py_serialize([], [("huge", 24, (huge.size,), huge.tobytes())]))
    with gg.GGUFFile(p2) as f:
        dst = torch.zeros(huge.size, dtype=torch.uint8, device="cuda")
        f.upload("huge", dst, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()      # first pass: page cache + staging buffers warm
        t0 = time.perf_counter()
        f.upload("huge", dst, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert np.array_equal(dst.cpu().numpy(), huge)
    import gpu_util
    gpu_util.report(test="gguf_upload", bytes=int(huge.size), seconds=dt, GBps=huge.size / dt / 1e9)


def test_gpt2_model_as_gguf_on_the_cpu_backend(tmp_path):
    """BASELINE configs[3] says "gpt-2 117M GGUF Q4_0": the reference's .bin model written as GGUF by the reference's own writer (oracle/gpt2_harness
    TOGGUF), read back through THIS library's reader into the reference's unmodified gpt-2 graph — logits bit-identical to the .bin-loaded run.  CPU
    backend and a 2-layer model here (the -m gpu twin runs all 12 layers on the plug-in, uploading through ggml_cdna4_gguf_upload)."""
    import subprocess
    import sys
    H = os.path.join(R.REF_DIR, "gpt2_harness")
    if not os.path.exists(H):
        pytest.skip("oracle/_ref not built")
    import ggml_amd.native as N
    f32, q4, gguf = (str(tmp_path / n) for n in ("f32.bin", "q4_0.bin", "model.gguf"))
    subprocess.run([sys.executable, os.path.join(R.ROOT, "tools", "make_synth_gpt2.py"), f32, "--layers", "2"], check=True, timeout=600)
    subprocess.run([os.path.join(R.REF_DIR, "gpt-2-quantize"), f32, q4, "q4_0"], check=True, timeout=600, capture_output=True)
    os.remove(f32)
    subprocess.run([H, q4, "CPU", "-", "TOGGUF:" + gguf, "0", "0", "1"], check=True, timeout=600, capture_output=True)
    env = dict(os.environ, CDNA4_KERNELS_SO=N.LIB_PATH)
    for model, out in ((q4, "a.bin"), (gguf, "b.bin")):
        r = subprocess.run([H, model, "CPU", "-", str(tmp_path / out), "8", "2", "4"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    a, b = np.fromfile(str(tmp_path / "a.bin"), np.uint32), np.fromfile(str(tmp_path / "b.bin"), np.uint32)
    assert a.size == 3 * 50257 and np.array_equal(a, b)


def test_header_with_many_keys_opens_in_linear_time(gg, tmp_path):
    """duplicate detection and find_key are hashed: 200,000 keys open in well under a second (a linear scan per key, as in the
    reference, would be 2e10 string compares)"""
    import time
    p = str(tmp_path / "many.gguf")
    open(p, "wb").write(G.py_serialize([("k%06d" % i, "u32", i) for i in range(200000)], []))
    L = gg._lib()
    t0 = time.time()
    h = L.ggml_cdna4_gguf_open(p.encode(), 1)
    dt = time.time() - t0
    assert h and L.ggml_cdna4_gguf_n_kv(h) == 200000 and L.ggml_cdna4_gguf_find_key(h, b"k199999") == 199999
    L.ggml_cdna4_gguf_close(h)
    assert dt < 5.0, dt
