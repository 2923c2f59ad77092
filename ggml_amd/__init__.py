"""ggml_amd — an MI355X-native (gfx950) implementation of ggml's quantized MUL_MAT hot path.

Layers (bottom up):
  csrc/*.hip                 hand-written HIP kernels (int8-dot GEMV, fp16-MFMA GEMM, activation quantizers, ops)
  lib/libcdna4_kernels.so    C-ABI of include/ggml_cdna4.h
  lib/libggml-cdna4.so       ggml backend plug-in (ggml_backend_init) — loaded by unmodified ggml binaries
  ggml_amd.ops               Python host mirror over ctypes (torch only as the device-memory / stream provider)
There is no CPU fallback anywhere: every op raises if the native library or a GPU is missing.
"""
from . import native  # noqa: F401
from .gtypes import GGMLType, row_size, blck_size  # noqa: F401
