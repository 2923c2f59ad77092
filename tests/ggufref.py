"""ctypes access to the GGUF writer and reader of the unmodified reference (oracle/_ref/libggml-base.so: src/gguf.cpp) —
test infrastructure only, like refutil.py.  Used to generate tests/golden/*.gguf and to run the reference's reader next
to ours on the same (also corrupted) files."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np

import refutil as R

# enum gguf_type (include/gguf.h:52-67)
T = {"u8": 0, "i8": 1, "u16": 2, "i16": 3, "u32": 4, "i32": 5, "f32": 6, "bool": 7, "str": 8, "arr": 9, "u64": 10, "i64": 11, "f64": 12}
CT = {0: C.c_uint8, 1: C.c_int8, 2: C.c_uint16, 3: C.c_int16, 4: C.c_uint32, 5: C.c_int32, 6: C.c_float, 7: C.c_bool,
      10: C.c_uint64, 11: C.c_int64, 12: C.c_double}
NP = {0: np.uint8, 1: np.int8, 2: np.uint16, 3: np.int16, 4: np.uint32, 5: np.int32, 6: np.float32, 7: np.bool_,
      10: np.uint64, 11: np.int64, 12: np.float64}
SUFFIX = {0: "u8", 1: "i8", 2: "u16", 3: "i16", 4: "u32", 5: "i32", 6: "f32", 7: "bool", 10: "u64", 11: "i64", 12: "f64"}


class InitParams(C.Structure):          # struct gguf_init_params, include/gguf.h:72-77
    _fields_ = [("no_alloc", C.c_bool), ("ctx", C.c_void_p)]


class GgmlInit(C.Structure):            # struct ggml_init_params, include/ggml.h:624-629
    _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]


_b = None


def base():
    global _b
    if _b is None:
        b, _ = R.ref()
        vp, i64, sz = C.c_void_p, C.c_int64, C.c_size_t
        b.gguf_init_empty.restype = vp
        b.gguf_init_from_file.restype = vp
        b.gguf_init_from_file.argtypes = [C.c_char_p, InitParams]
        b.gguf_free.argtypes = [vp]
        b.gguf_write_to_file.restype = C.c_bool
        b.gguf_write_to_file.argtypes = [vp, C.c_char_p, C.c_bool]
        for k, ct in CT.items():
            f = getattr(b, "gguf_set_val_" + SUFFIX[k]); f.argtypes = [vp, C.c_char_p, ct]; f.restype = None
            f = getattr(b, "gguf_get_val_" + SUFFIX[k]); f.argtypes = [vp, i64]; f.restype = ct
        b.gguf_set_val_str.argtypes = [vp, C.c_char_p, C.c_char_p]
        b.gguf_set_arr_data.argtypes = [vp, C.c_char_p, C.c_int, vp, sz]
        b.gguf_set_arr_str.argtypes = [vp, C.c_char_p, C.POINTER(C.c_char_p), sz]
        b.gguf_add_tensor.argtypes = [vp, vp]
        b.ggml_new_tensor.restype = vp
        b.ggml_new_tensor.argtypes = [vp, C.c_int, C.c_int, C.POINTER(i64)]
        b.ggml_set_name.restype = vp
        b.ggml_set_name.argtypes = [vp, C.c_char_p]
        b.ggml_get_data.restype = vp
        b.ggml_get_data.argtypes = [vp]
        b.ggml_nbytes.restype = sz
        b.ggml_nbytes.argtypes = [vp]
        b.ggml_init.restype = vp
        b.ggml_init.argtypes = [GgmlInit]
        for name, res in (("gguf_get_version", C.c_uint32), ("gguf_get_alignment", sz), ("gguf_get_data_offset", sz), ("gguf_get_n_kv", i64),
                          ("gguf_get_n_tensors", i64)):
            f = getattr(b, name); f.restype = res; f.argtypes = [vp]
        for name, res in (("gguf_get_key", C.c_char_p), ("gguf_get_kv_type", C.c_int), ("gguf_get_arr_type", C.c_int), ("gguf_get_arr_n", sz),
                          ("gguf_get_val_str", C.c_char_p), ("gguf_get_arr_data", vp), ("gguf_get_tensor_name", C.c_char_p),
                          ("gguf_get_tensor_type", C.c_int), ("gguf_get_tensor_offset", sz), ("gguf_get_tensor_size", sz)):
            f = getattr(b, name); f.restype = res; f.argtypes = [vp, i64]
        b.gguf_get_arr_str.restype = C.c_char_p
        b.gguf_get_arr_str.argtypes = [vp, i64, sz]
        b.gguf_find_key.restype = i64
        b.gguf_find_key.argtypes = [vp, C.c_char_p]
        _b = b
    return _b


def write_with_reference(path, kv, tensors):
    """kv: [(key, 'u8'|..|'str'|'arr:<elem>', value)], tensors: [(name, ggml_type, ne tuple, payload bytes)]"""
    b = base()
    g = b.gguf_init_empty()
    keep = []
    for key, kind, val in kv:
        k = key.encode()
        if kind == "str":
            b.gguf_set_val_str(g, k, val.encode())
        elif kind == "arr:str":
            arr = (C.c_char_p * len(val))(*[s.encode() for s in val])
            b.gguf_set_arr_str(g, k, arr, len(val))
        elif kind.startswith("arr:"):
            t = T[kind[4:]]
            a = np.asarray(val, NP[t])
            b.gguf_set_arr_data(g, k, t, a.ctypes.data_as(C.c_void_p), a.size)
        else:
            getattr(b, "gguf_set_val_" + kind)(g, k, val)
    total = sum(len(p) for _, _, _, p in tensors) + (len(tensors) + 1) * 1024 + (1 << 16)
    ctx = b.ggml_init(GgmlInit(total, None, False))
    for name, t, ne, payload in tensors:
        ne_c = (C.c_int64 * len(ne))(*ne)
        tens = b.ggml_new_tensor(ctx, t, len(ne), ne_c)
        b.ggml_set_name(tens, name.encode())
        assert b.ggml_nbytes(tens) == len(payload), (name, b.ggml_nbytes(tens), len(payload))
        C.memmove(b.ggml_get_data(tens), payload, len(payload))
        b.gguf_add_tensor(g, tens)
        keep.append(tens)
    ok = b.gguf_write_to_file(g, path.encode(), False)
    b.gguf_free(g)
    b.ggml_free(ctx)
    assert ok


def read_with_reference(path):
    """what the reference's reader reports (JSON-able); tensor payloads as sha256 of the bytes at data_offset + offset"""
    b = base()
    g = b.gguf_init_from_file(path.encode(), InitParams(True, None))
    if not g:
        return None
    raw = open(path, "rb").read()
    out = {"version": b.gguf_get_version(g), "alignment": b.gguf_get_alignment(g), "data_offset": b.gguf_get_data_offset(g), "kv": [], "tensors": []}
    for i in range(b.gguf_get_n_kv(g)):
        key, t = b.gguf_get_key(g, i).decode(), b.gguf_get_kv_type(g, i)
        if t == T["arr"]:
            et, n = b.gguf_get_arr_type(g, i), b.gguf_get_arr_n(g, i)
            if et == T["str"]:
                val = [b.gguf_get_arr_str(g, i, j).decode("utf-8", "surrogateescape") for j in range(n)]
            else:
                dt = np.dtype(NP[et])
                val = np.frombuffer(C.string_at(b.gguf_get_arr_data(g, i), n * dt.itemsize), dt).tolist() if n else []
            out["kv"].append([key, t, et, val])
        elif t == T["str"]:
            out["kv"].append([key, t, None, b.gguf_get_val_str(g, i).decode("utf-8", "surrogateescape")])
        else:
            v = getattr(b, "gguf_get_val_" + SUFFIX[t])(g, i)
            out["kv"].append([key, t, None, v])
    size = 0
    for i in range(b.gguf_get_n_tensors(g)):
        off, n = b.gguf_get_tensor_offset(g, i), b.gguf_get_tensor_size(g, i)
        lo = out["data_offset"] + off
        out["tensors"].append({"name": b.gguf_get_tensor_name(g, i).decode(), "type": b.gguf_get_tensor_type(g, i), "offset": off, "size": n,
                               "sha256": hashlib.sha256(raw[lo:lo + n]).hexdigest() if lo + n <= len(raw) else None})
        size = off + (n + out["alignment"] - 1) // out["alignment"] * out["alignment"]
    out["data_size"] = size
    b.gguf_free(g)
    return out


def reference_accepts(path, with_data=False):
    """does gguf_init_from_file succeed?  Run in a child process: the reference aborts (GGML_ASSERT) or traps on some
    malformed inputs.  Returns True / False, or None if the child died (abort / SIGFPE / ...)."""
    code = ("import sys; sys.path.insert(0, %r); import ctypes as C, ggufref as G; b = G.base(); ctx = C.c_void_p();\n"
            "p = G.InitParams(False, C.cast(C.pointer(ctx), C.c_void_p)) if %r else G.InitParams(True, None)\n"
            "g = b.gguf_init_from_file(%r, p); print('ACCEPT' if g else 'REJECT')") % (os.path.dirname(os.path.abspath(__file__)), with_data, path.encode())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    if "ACCEPT" in r.stdout:
        return True
    if "REJECT" in r.stdout:
        return False
    return None


# ---- a plain-Python GGUF serializer (layout: include/gguf.h:1-31) for the files the reference's writer cannot produce: a
# ---- non-default alignment (its writer ignores general.alignment, src/gguf.cpp:1102,1289) and deliberately malformed files
import struct

_FMT = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<b", 10: "<Q", 11: "<q", 12: "<d"}
_BLCK = {0: (1, 4), 1: (1, 2), 2: (32, 18), 8: (32, 34), 12: (256, 144), 13: (256, 176), 14: (256, 210), 26: (1, 4)}


def _s(x):
    b = x if isinstance(x, bytes) else x.encode()
    return struct.pack("<Q", len(b)) + b


def py_serialize(kv, tensors, alignment=32, version=3, magic=b"GGUF", n_kv=None, n_tensors=None, offsets=None, pad_data=True):
    """kv as for write_with_reference (plus raw entries ('key', 'raw', bytes) = pre-encoded type + value);
    tensors: [(name, type, ne, payload)].  `offsets` overrides the tensor offsets written into the infos."""
    out = bytearray(magic + struct.pack("<I", version) + struct.pack("<q", len(tensors) if n_tensors is None else n_tensors)
                    + struct.pack("<q", len(kv) if n_kv is None else n_kv))
    for key, kind, val in kv:
        out += _s(key)
        if kind == "raw":
            out += val
        elif kind == "str":
            out += struct.pack("<i", 8) + _s(val)
        elif kind == "arr:str":
            out += struct.pack("<i", 9) + struct.pack("<i", 8) + struct.pack("<Q", len(val)) + b"".join(_s(v) for v in val)
        elif kind.startswith("arr:"):
            t = T[kind[4:]]
            out += struct.pack("<i", 9) + struct.pack("<i", t) + struct.pack("<Q", len(val)) + b"".join(struct.pack(_FMT[t], v) for v in val)
        else:
            out += struct.pack("<i", T[kind]) + struct.pack(_FMT[T[kind]], val)
    off, offs = 0, []
    for i, (name, t, ne, payload) in enumerate(tensors):
        offs.append(off if offsets is None else offsets[i])
        off += (len(payload) + alignment - 1) // alignment * alignment
    for (name, t, ne, payload), o in zip(tensors, offs):
        out += _s(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<q", n) for n in ne) + struct.pack("<i", t) + struct.pack("<Q", o)
    if tensors or pad_data:
        out += b"\0" * (-len(out) % alignment)
    for name, t, ne, payload in tensors:
        out += payload + (b"\0" * (-len(payload) % alignment) if pad_data else b"")
    return bytes(out)


def fixture_content(alignment=None):
    """the key/value pairs and tensors of tests/golden/small*.gguf (needs the reference for ggml_quantize_chunk)"""
    rng = np.random.default_rng(2024)
    kv = [
        ("general.architecture", "str", "gpt2"),
        ("general.name", "str", "gguf fixture é中"),
        ("test.u8", "u8", 200), ("test.i8", "i8", -100), ("test.u16", "u16", 60000), ("test.i16", "i16", -30000),
        ("test.u32", "u32", 4000000000), ("test.i32", "i32", -2000000000), ("test.f32", "f32", 3.25),
        ("test.u64", "u64", 2 ** 63 + 5), ("test.i64", "i64", -(2 ** 62)), ("test.f64", "f64", -1.0 / 3.0),
        ("test.bool_t", "bool", True), ("test.bool_f", "bool", False),
        ("test.empty_str", "str", ""),
        ("test.arr_u8", "arr:u8", [0, 1, 255]), ("test.arr_i32", "arr:i32", [-5, 0, 7, 2 ** 31 - 1]),
        ("test.arr_f32", "arr:f32", [0.5, -1.5, 1e-20]), ("test.arr_u64", "arr:u64", [1, 2 ** 40]),
        ("test.arr_f64", "arr:f64", [2.0 ** -40]), ("test.arr_i16", "arr:i16", [-1, 1]),
        ("tokenizer.ggml.tokens", "arr:str", ["<s>", "hello", "", "wörld", "x" * 70]),
    ]
    if alignment is not None:
        kv.append(("general.alignment", "u32", alignment))
    tensors = []
    for name, t, ne in (("blk.0.attn_q.weight", R.Q4_K, (256, 8)), ("blk.0.ffn_up.weight", R.Q8_0, (64, 3)),
                        ("blk.0.attn_k.weight", R.Q4_0, (32, 5)), ("blk.0.attn_v.weight", R.Q5_K, (512, 2)),
                        ("output.weight", R.Q6_K, (256, 3))):
        w = R.r_quantize(t, rng.uniform(-1, 1, (ne[1], ne[0])).astype(np.float32))
        tensors.append((name, t, ne, w.tobytes()))
    tensors.append(("blk.0.attn_norm.bias", R.F32, (7,), rng.standard_normal(7).astype(np.float32).tobytes()))
    tensors.append(("token_embd.f16", R.F16, (5, 3), rng.standard_normal(15).astype(np.float16).tobytes()))
    tensors.append(("ids.i32", 26, (3, 2, 2), np.arange(12, dtype=np.int32).tobytes()))
    tensors.append(("four.d", R.F32, (2, 3, 2, 2), rng.standard_normal(24).astype(np.float32).tobytes()))
    return kv, tensors


def malformed_cases():
    """[(label, file bytes)] — deterministic, reference-free: every acceptance check of src/gguf.cpp:319-617 from both sides.
    tests/golden/gguf_expected.json records what the reference's reader says about each."""
    kv = [("a.u32", "u32", 7), ("b.str", "str", "hello"), ("c.arr", "arr:i16", [1, 2, 3]), ("d.strs", "arr:str", ["x", "yz"])]
    f32 = lambda n: np.arange(n, dtype=np.float32).tobytes()
    q80 = bytes(range(34)) * 2                                      # two Q8_0 blocks = one row of 64
    tens = [("t0", 0, (5,), f32(5)), ("t1", 8, (64, 1), q80), ("t2", 0, (2, 3), f32(6))]
    ok = py_serialize(kv, tens)
    cases = [("valid", ok), ("valid_no_tensors", py_serialize(kv, [])), ("valid_nothing", py_serialize([], [], pad_data=False)),
             ("bad_magic", b"GGUG" + ok[4:]), ("short_magic", ok[:3]), ("empty_file", b""),
             ("version_0", py_serialize(kv, tens, version=0)), ("version_1", py_serialize(kv, tens, version=1)),
             ("version_2", py_serialize(kv, tens, version=2)), ("version_4", py_serialize(kv, tens, version=4)),
             ("n_kv_negative", py_serialize(kv, tens, n_kv=-1)), ("n_kv_too_many", py_serialize(kv, tens, n_kv=len(kv) + 1)),
             ("n_kv_huge", py_serialize(kv, tens, n_kv=2 ** 62)), ("n_tensors_negative", py_serialize(kv, tens, n_tensors=-3)),
             ("n_tensors_too_many", py_serialize(kv, tens, n_tensors=len(tens) + 1)), ("n_tensors_huge", py_serialize(kv, tens, n_tensors=2 ** 61)),
             ("n_tensors_fewer", py_serialize(kv, tens, n_tensors=2)),
             ("dup_key", py_serialize(kv + [("a.u32", "u32", 8)], tens)),
             ("kv_type_13", py_serialize(kv + [("e", "raw", struct.pack("<i", 13) + b"\0" * 8)], tens)),
             ("kv_type_neg", py_serialize(kv + [("e", "raw", struct.pack("<i", -1) + b"\0" * 8)], tens)),
             ("array_of_arrays", py_serialize(kv + [("e", "raw", struct.pack("<iiQ", 9, 9, 1) + b"\0" * 16)], tens)),
             ("array_bad_elem_type", py_serialize(kv + [("e", "raw", struct.pack("<iiQ", 9, 14, 1) + b"\0" * 16)], tens)),
             ("array_n_beyond_eof", py_serialize([("e", "raw", struct.pack("<iiQ", 9, 4, 2 ** 40))], [])),
             ("str_len_beyond_eof", py_serialize([("e", "raw", struct.pack("<iQ", 8, 2 ** 40) + b"abc")], [])),
             ("str_array_n_beyond_eof", py_serialize([("e", "raw", struct.pack("<iiQ", 9, 8, 2 ** 50))], [])),
             ("key_len_beyond_eof", ok[:24] + struct.pack("<Q", 2 ** 45) + ok[32:]),
             ("bool_value_2", py_serialize(kv + [("e", "raw", struct.pack("<ib", 7, 2))], tens)),
             ("empty_array", py_serialize(kv + [("e", "arr:f32", [])], tens)), ("empty_str_array", py_serialize(kv + [("e", "arr:str", [])], tens)),
             ("align_1", py_serialize(kv + [("general.alignment", "u32", 1)], tens, alignment=1)),
             ("align_64", py_serialize(kv + [("general.alignment", "u32", 64)], tens, alignment=64)),
             ("align_0", py_serialize(kv + [("general.alignment", "u32", 0)], tens)),
             ("align_3", py_serialize(kv + [("general.alignment", "u32", 3)], tens)),
             ("align_64_but_offsets_32", py_serialize(kv + [("general.alignment", "u32", 64)], tens, alignment=32)),
             ("dup_tensor", py_serialize(kv, tens + [("t0", 0, (1,), f32(1))])),
             ("name_63", py_serialize(kv, [("n" * 63, 0, (5,), f32(5))])), ("name_64", py_serialize(kv, [("n" * 64, 0, (5,), f32(5))])),
             ("name_empty", py_serialize(kv, [("", 0, (5,), f32(5))])),
             ("name_embedded_nul_dup", py_serialize(kv, [("ab", 0, (1,), f32(1)), ("ab\0cd", 0, (1,), f32(1))])),
             ("dims_0", py_serialize(kv, [("s", 0, (), f32(1))])), ("dims_4", py_serialize(kv, [("s", 0, (1, 2, 1, 3), f32(6))])),
             ("dims_5", py_serialize(kv, [("s", 0, (1, 1, 1, 1, 1), f32(1))])),
             ("ne_negative", py_serialize(kv, [("s", 0, (-4,), b"")])), ("ne0_zero", py_serialize(kv, [("s", 0, (0, 3), b"")])),
             ("ne_overflow", py_serialize(kv, [("s", 24, (2 ** 32, 2 ** 31), b"")])),
             ("ne_just_representable", py_serialize(kv, [("s", 24, (2 ** 31, 2 ** 31), b"")], pad_data=False)),
             ("row_not_multiple_of_block", py_serialize(kv, [("s", 8, (48, 1), b"\0" * 51)])),
             ("tensor_type_39", py_serialize(kv, [("s", 39, (4,), f32(4))])), ("tensor_type_neg", py_serialize(kv, [("s", -2, (4,), f32(4))])),
             ("tensor_type_removed_4", py_serialize(kv, [("s", 4, (32,), b"\0" * 20)])),
             ("offset_gap", py_serialize(kv, tens, offsets=[0, 64, 160])), ("offset_unpadded", py_serialize(kv, tens, offsets=[0, 20, 88])),
             ("offset_first_nonzero", py_serialize(kv, tens, offsets=[32, 64, 160])),
             ("data_truncated", ok[:-40]), ("data_missing", ok[:len(ok) - 32 - 96 - 32]), ("last_padding_missing", ok[:-8])]
    # the metadata cut at every 7th byte
    meta_end = len(ok) - 32 - 96 - 32
    cases += [("cut_%04d" % n, ok[:n]) for n in range(5, meta_end, 7)]
    return cases


def reference_accepts_many(paths, with_data=False):
    """gguf_init_from_file verdicts for many files with few child processes: True / False per path, None where the child died"""
    res, i = [], 0
    here = os.path.dirname(os.path.abspath(__file__))
    while i < len(paths):
        code = ("import sys; sys.path.insert(0, %r); import ctypes as C, ggufref as G; b = G.base()\n"
                "for p in %r:\n"
                "    ctx = C.c_void_p(); ip = G.InitParams(False, C.cast(C.pointer(ctx), C.c_void_p)) if %r else G.InitParams(True, None)\n"
                "    g = b.gguf_init_from_file(p.encode(), ip); print('ACCEPT' if g else 'REJECT', flush=True)\n") % (here, paths[i:], with_data)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        got = [ln == "ACCEPT" for ln in r.stdout.split() if ln in ("ACCEPT", "REJECT")]
        res += got
        i += len(got)
        if i < len(paths) and len(got) < len(paths) - (i - len(got)):
            res.append(None)                       # the child died on this file
            i += 1
    return res
