// exact_math.h — scalar restatements of the two exponentials the reference CPU backend's SOFT_MAX uses (x86-64 AVX2 build), for GGML_CDNA4_EXACT:
//   exact_v_expf      one lane of ggml_v_expf, /root/reference/src/ggml-cpu/ggml-cpu.c:1912-1949 (same fused / unfused operations in the same order)
//   exact_expf_glibc  glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c + e_exp2f_data.c: N = 32 table, cubic in double), which the CPU runs on the
//                     n % 8 tail of a row (ggml_vec_soft_max_f32, ggml-cpu.c:2086-2090)
// Provenance: exact_expf_glibc restates the ALGORITHM and the table constants of the GNU C Library's expf (glibc 2.35, sysdeps/ieee754/flt-32/e_expf.c and
// e_exp2f_data.c — originally contributed by Szabolcs Nagy / Arm Ltd. as part of the optimized-routines project; glibc is LGPL-2.1-or-later, optimized-routines MIT) because
// bit-identity with the CPU backend's libm call is the point of this mode; exact_v_expf restates ggml_v_expf of the reference (MIT).  No source text is copied.
// Plain C++ (no intrinsics): compiled for the GPU by exact.hip (-ffp-contract=off: only the explicit fma calls fuse) and for the host by
// tests/test_exact_math.py, which checks both bit for bit against this machine's libm / a numpy model of the AVX2 sequence.
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define EXACT_FN __host__ __device__ inline __attribute__((always_inline))
#else
#define EXACT_FN static inline
#endif
EXACT_FN uint32_t ex_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
EXACT_FN float ex_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
EXACT_FN uint64_t ex_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
EXACT_FN double ex_asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

EXACT_FN float exact_v_expf(float x) {
    const float r = 0x1.8p23f;
    const float z = __builtin_fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = __builtin_fmaf(-n, 0x1.7f7d1cp-20f, __builtin_fmaf(-n, 0x1.62e4p-1f, x));
    const uint32_t e = ex_asuint(z) << 23;
    const float k = ex_asfloat(e + ex_asuint(1.0f));
    const float an = __builtin_fabsf(n);
    const float u = b * b;
    const float lin = 0x1.ffffecp-1f * b;
    const float j = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, __builtin_fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)), u, lin);
    if (!(an > 126.f)) return __builtin_fmaf(j, k, k);
    const uint32_t g = (n <= 0.f) ? 0x82000000u : 0u;
    const float s1 = ex_asfloat(g + 0x7f000000u);
    const float s2 = ex_asfloat(e - g);
    if (an > 192.f) return s1 * s1;
    return __builtin_fmaf(s2, j, s2) * s1;
}

EXACT_FN float exact_expf_glibc(float x) {
    const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    const uint32_t ix = ex_asuint(x), abstop = (ix >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                                             // |x| >= 88 or NaN
        if (ix == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_inff();
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    const double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    const uint64_t ki = ex_asuint64(kd);
    kd -= SHIFT;
    const double rr = z - kd;
    uint64_t t = T[ki % 32];
    t += ki << 47;
    const double s = ex_asdouble(t);
    z = C0 * rr + C1;
    const double r2 = rr * rr;
    double y = C2 * rr + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}
