"""GGUF files through the native reader of libcdna4_kernels.so (include/ggml_cdna4_gguf.h; reference reader:
src/gguf.cpp:319-705, API include/gguf.h:79-126).  The file is mmap'ed by the library; tensor payloads are exposed as
zero-copy numpy views into that mapping — the block arrays ggml_cdna4_mul_mat takes — and `GGUFFile.qtensor()` puts one
into HBM as an `ops.QTensor`."""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import native

_i64, _vp, _sz, _int, _cp = C.c_int64, C.c_void_p, C.c_size_t, C.c_int, C.c_char_p
# every symbol include/ggml_cdna4_gguf.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("ggml_cdna4_gguf_open", _vp, [_cp, _int]),
    ("ggml_cdna4_gguf_close", None, [_vp]),
    ("ggml_cdna4_gguf_version", C.c_uint32, [_vp]),
    ("ggml_cdna4_gguf_alignment", _sz, [_vp]),
    ("ggml_cdna4_gguf_data_offset", _sz, [_vp]),
    ("ggml_cdna4_gguf_data_size", _sz, [_vp]),
    ("ggml_cdna4_gguf_n_kv", _i64, [_vp]),
    ("ggml_cdna4_gguf_find_key", _i64, [_vp, _cp]),
    ("ggml_cdna4_gguf_key", _cp, [_vp, _i64]),
    ("ggml_cdna4_gguf_kv_type", _int, [_vp, _i64]),
    ("ggml_cdna4_gguf_arr_type", _int, [_vp, _i64]),
    ("ggml_cdna4_gguf_arr_n", _sz, [_vp, _i64]),
    ("ggml_cdna4_gguf_val", _int, [_vp, _i64, _int, _vp]),
    ("ggml_cdna4_gguf_val_str", _cp, [_vp, _i64]),
    ("ggml_cdna4_gguf_arr_data", _vp, [_vp, _i64]),
    ("ggml_cdna4_gguf_arr_str", _cp, [_vp, _i64, _sz]),
    ("ggml_cdna4_gguf_n_tensors", _i64, [_vp]),
    ("ggml_cdna4_gguf_find_tensor", _i64, [_vp, _cp]),
    ("ggml_cdna4_gguf_tensor_name", _cp, [_vp, _i64]),
    ("ggml_cdna4_gguf_tensor_type", _int, [_vp, _i64]),
    ("ggml_cdna4_gguf_tensor_ne", _int, [_vp, _i64, _vp]),
    ("ggml_cdna4_gguf_tensor_offset", _sz, [_vp, _i64]),
    ("ggml_cdna4_gguf_tensor_size", _sz, [_vp, _i64]),
    ("ggml_cdna4_gguf_tensor_data", _vp, [_vp, _i64]),
    ("ggml_cdna4_gguf_upload", _int, [_vp, _i64, _vp, _sz, _vp]),
    ("ggml_cdna4_gguf_blck_size", _i64, [_int]),
    ("ggml_cdna4_gguf_type_size", _sz, [_int]),
]

# enum gguf_type (include/gguf.h:52-67) -> numpy dtype of the fixed-size value types
STRING, ARRAY = 8, 9
DTYPES = {0: np.uint8, 1: np.int8, 2: np.uint16, 3: np.int16, 4: np.uint32, 5: np.int32, 6: np.float32, 7: np.bool_,
          10: np.uint64, 11: np.int64, 12: np.float64}

TensorInfo = namedtuple("TensorInfo", "name type ne offset size")

_bound = None


def _lib():
    global _bound
    if _bound is None:
        L = native.lib()
        for name, res, args in SYMBOLS:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _bound = L
    return _bound


class GGUFError(RuntimeError):
    pass


class GGUFFile:
    """One open GGUF file.  `kv` maps keys to Python values (scalars, str, numpy arrays, lists of str), `tensors` lists the
    tensor table in file order.  Views returned by `tensor_bytes` are valid until `close()`."""

    def __init__(self, path, require_data=True):
        L = _lib()
        self._L = L
        self._h = L.ggml_cdna4_gguf_open(str(path).encode(), 1 if require_data else 0)
        if not self._h:
            raise GGUFError(L.ggml_cdna4_last_error().decode())
        h = self._h
        self.version = L.ggml_cdna4_gguf_version(h)
        self.alignment = L.ggml_cdna4_gguf_alignment(h)
        self.data_offset = L.ggml_cdna4_gguf_data_offset(h)
        self.data_size = L.ggml_cdna4_gguf_data_size(h)
        self.kv, self.kv_types = {}, {}
        for i in range(L.ggml_cdna4_gguf_n_kv(h)):
            key = L.ggml_cdna4_gguf_key(h, i).decode()
            t = L.ggml_cdna4_gguf_kv_type(h, i)
            if t == ARRAY:
                et, n = L.ggml_cdna4_gguf_arr_type(h, i), L.ggml_cdna4_gguf_arr_n(h, i)
                if et == STRING:
                    val = [L.ggml_cdna4_gguf_arr_str(h, i, j).decode("utf-8", "surrogateescape") for j in range(n)]
                else:
                    dt = np.dtype(DTYPES[et])
                    buf = C.string_at(L.ggml_cdna4_gguf_arr_data(h, i), n * dt.itemsize) if n else b""
                    val = np.frombuffer(buf, dtype=dt).copy()
                self.kv_types[key] = (ARRAY, et)
            elif t == STRING:
                val = L.ggml_cdna4_gguf_val_str(h, i).decode("utf-8", "surrogateescape")
                self.kv_types[key] = (STRING, None)
            else:
                out = np.zeros(1, DTYPES[t])
                if L.ggml_cdna4_gguf_val(h, i, t, out.ctypes.data_as(_vp)) != 0:
                    raise GGUFError(L.ggml_cdna4_last_error().decode())
                val = out[0].item()
                self.kv_types[key] = (t, None)
            self.kv[key] = val
        self.tensors = []
        ne = (C.c_int64 * 4)()
        for i in range(L.ggml_cdna4_gguf_n_tensors(h)):
            L.ggml_cdna4_gguf_tensor_ne(h, i, ne)
            self.tensors.append(TensorInfo(L.ggml_cdna4_gguf_tensor_name(h, i).decode(), L.ggml_cdna4_gguf_tensor_type(h, i),
                                           tuple(ne), L.ggml_cdna4_gguf_tensor_offset(h, i), L.ggml_cdna4_gguf_tensor_size(h, i)))
        self._index = {t.name: i for i, t in enumerate(self.tensors)}

    def close(self):
        if self._h:
            self._L.ggml_cdna4_gguf_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tensor_id(self, name_or_id):
        if isinstance(name_or_id, int):
            return name_or_id
        if name_or_id not in self._index:
            raise KeyError(name_or_id)
        return self._index[name_or_id]

    def tensor_bytes(self, name_or_id):
        """read-only uint8 view of the tensor's payload inside the mapping (no copy)"""
        if not self._h:
            raise GGUFError("file is closed")
        i = self.tensor_id(name_or_id)
        p = self._L.ggml_cdna4_gguf_tensor_data(self._h, i)
        if not p:
            raise GGUFError(self._L.ggml_cdna4_last_error().decode())
        n = self.tensors[i].size
        a = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p)) if n else np.zeros(0, np.uint8)
        a.flags.writeable = False
        return a

    def upload(self, name_or_id, dst, stream=None):
        """payload -> the uint8 device tensor `dst` through the library's pinned double buffer (ggml_cdna4_gguf_upload)"""
        i = self.tensor_id(name_or_id)
        if self._L.ggml_cdna4_gguf_upload(self._h, i, dst.data_ptr(), dst.numel() * dst.element_size(), stream) != 0:
            raise GGUFError(self._L.ggml_cdna4_last_error().decode())

    def qtensor(self, name_or_id, device="cuda"):
        """the 2-D quantized weight `name` in HBM as an ops.QTensor ([K, M] in ggml order: ne0 = K contiguous, ne1 = M rows)"""
        from . import ops
        t = self.tensors[self.tensor_id(name_or_id)]
        if t.ne[2] != 1 or t.ne[3] != 1:
            raise GGUFError("tensor '%s' is not a matrix" % t.name)
        return ops.QTensor.from_host_bytes(t.type, t.ne[0], t.ne[1], self.tensor_bytes(name_or_id), device=device)
