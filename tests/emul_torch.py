"""TEST INFRASTRUCTURE: run tests written for the GPU library against the whole-library CPU emulation (tools/emul/lib_emul: the product's own
kernel and host sources as a host shared library with the same C-ABI).  activate() swaps the emulated library in for ggml_amd.native.lib() and
makes "cuda" tensors ordinary CPU tensors whose memory is a shared mapping (the emulated work-groups are processes that write into it).
Nothing under ggml_amd/ knows about this; the product still has no CPU path.  Used by tests/test_gpu_tests_on_the_emulator.py."""
import contextlib
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_real_empty = torch.empty
_so = None
_keep = []


def _shared(nbytes):
    p = _so.cdna4_emul_alloc(C.c_size_t(max(int(nbytes), 1)))
    buf = (C.c_uint8 * max(int(nbytes), 1)).from_address(p)
    _keep.append(buf)
    return buf


def _shared_tensor(shape, dtype):
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
    n = int(np.prod(shape)) if len(shape) else 1
    es = _real_empty(0, dtype=dtype).element_size()
    t = torch.frombuffer(_shared(n * es), dtype=dtype, count=n) if n else _real_empty(0, dtype=dtype)
    return t.view(shape)


def activate():
    global _so
    if _so is not None:
        return
    # never on a box with a GPU: the emulation is a lint for sessions without hardware, not a substitute for it — with a device
    # present the -m gpu tests must run the product's library on it
    if torch.cuda.is_available():
        raise RuntimeError("CDNA4_TESTS_ON_EMULATOR is refused on a box with a GPU: run the tests on the device")
    sys.path.insert(0, os.path.join(ROOT, "tools", "emul"))
    import lib_emul_check
    from ggml_amd import native
    so = C.CDLL(lib_emul_check.build_so())
    so.cdna4_emul_alloc.restype = C.c_void_p; so.cdna4_emul_alloc.argtypes = [C.c_size_t]
    for name, res, args in native.SYMBOLS:
        fn = getattr(so, name); fn.restype, fn.argtypes = res, args
    _so = so
    native._lib = so                                  # what native.lib() returns from now on (this process only)

    real_empty, real_zeros = torch.empty, torch.zeros

    def _size(args):
        return tuple(args[0]) if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)) else tuple(args)

    def empty(*size, dtype=None, device=None, **kw):
        return _shared_tensor(_size(size), dtype or torch.float32)      # (mappings are zero-filled: fine for empty and zeros alike)

    def to_shared(self):
        t = _shared_tensor(self.shape, self.dtype)
        t.copy_(self)
        return t
    real_to = torch.Tensor.to

    def to(self, *a, **k):
        tgt = a[0] if a else k.get("device")
        if isinstance(tgt, (str, torch.device)) and "cuda" in str(tgt):
            return to_shared(self)
        if isinstance(tgt, torch.device) and tgt.type == "cpu" and not k and len(a) == 1:
            return to_shared(self)                    # `.to(b.device)` inside the wrappers: stay in shared memory
        return real_to(self, *a, **k)
    torch.empty, torch.zeros = empty, empty
    def full(size, fill_value, dtype=None, device=None, **kw):
        t = _shared_tensor(_size((size,)) if isinstance(size, int) else tuple(size), dtype or torch.float32)
        t.fill_(fill_value)
        return t
    torch.full = full
    torch.ones = lambda *size, dtype=None, device=None, **kw: full(_size(size), 1, dtype=dtype)
    torch.empty_like = torch.zeros_like = lambda t, dtype=None, **kw: _shared_tensor(t.shape, dtype or t.dtype)
    torch.Tensor.cuda = lambda self, *a, **k: to_shared(self)
    torch.Tensor.clone = lambda self, *a, **k: to_shared(self)
    torch.Tensor.to = to
    torch.Tensor.is_cuda = property(lambda self: True)

    class _Stream:
        cuda_stream = 0
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.device = lambda *a, **k: contextlib.nullcontext()
    torch.cuda.set_device = lambda *a, **k: None
