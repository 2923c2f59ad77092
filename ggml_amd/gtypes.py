"""ggml type ids and block geometry (include/ggml.h:351-390, src/ggml-common.h:161-328)."""
import enum


class GGMLType(enum.IntEnum):
    F32 = 0
    F16 = 1
    Q4_0 = 2
    Q4_1 = 3
    Q5_0 = 6
    Q5_1 = 7
    Q8_0 = 8
    Q2_K = 10
    Q3_K = 11
    Q4_K = 12
    Q5_K = 13
    Q6_K = 14
    Q8_K = 15
    IQ4_NL = 20
    IQ4_XS = 23
    I32 = 26
    BF16 = 30          # K / V of FLASH_ATTN_EXT only


_TYPE_SIZE = {GGMLType.F32: 4, GGMLType.F16: 2, GGMLType.BF16: 2, GGMLType.Q4_0: 18, GGMLType.Q8_0: 34, GGMLType.Q4_K: 144,
              GGMLType.Q5_K: 176, GGMLType.Q6_K: 210, GGMLType.Q8_K: 292, GGMLType.I32: 4,
              GGMLType.Q5_0: 22, GGMLType.Q2_K: 84, GGMLType.Q3_K: 110, GGMLType.Q4_1: 20, GGMLType.Q5_1: 24, GGMLType.IQ4_NL: 18, GGMLType.IQ4_XS: 136}
_BLCK = {GGMLType.F32: 1, GGMLType.F16: 1, GGMLType.BF16: 1, GGMLType.Q4_0: 32, GGMLType.Q8_0: 32, GGMLType.Q4_K: 256,
         GGMLType.Q5_K: 256, GGMLType.Q6_K: 256, GGMLType.Q8_K: 256, GGMLType.I32: 1,
         GGMLType.Q5_0: 32, GGMLType.Q2_K: 256, GGMLType.Q3_K: 256, GGMLType.Q4_1: 32, GGMLType.Q5_1: 32, GGMLType.IQ4_NL: 32, GGMLType.IQ4_XS: 256}
# Q5_0 / Q2_K / Q3_K / Q4_1 / Q5_1 / IQ4_NL / IQ4_XS: int8-dot GEMV units up to 8 activation rows, above that the Q8_0 / Q6_K MFMA GEMM on exactly re-encoded weights (convert_w.hip)
QUANT_WEIGHT_TYPES = (GGMLType.Q4_0, GGMLType.Q8_0, GGMLType.Q4_K, GGMLType.Q5_K, GGMLType.Q6_K, GGMLType.Q5_0, GGMLType.Q2_K, GGMLType.Q3_K,
                      GGMLType.Q4_1, GGMLType.Q5_1, GGMLType.IQ4_NL, GGMLType.IQ4_XS)


def blck_size(t):
    return _BLCK[GGMLType(t)]


def type_size(t):
    return _TYPE_SIZE[GGMLType(t)]


def row_size(t, k):
    """ggml_row_size (src/ggml.c:1176-1179)"""
    t = GGMLType(t)
    if k % _BLCK[t]:
        raise ValueError("k=%d is not a multiple of the %s block size %d" % (k, t.name, _BLCK[t]))
    return k // _BLCK[t] * _TYPE_SIZE[t]
