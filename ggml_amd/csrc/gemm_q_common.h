// gemm_q_common.h — device-side building blocks shared by the MFMA GEMM kernels (gemm_q_mfma.hip, gemm_q_t64.hip):
// the nibble -> fp16 unpack helpers, the per-format Raw<TYPE> fragment builders, the loader-side re-layout WDirect<TYPE>,
// the stage geometry WStage<TYPE, SKG> and the kernel parameter block.  Everything here is a template or __forceinline__.
#pragma once
#include "cdna4_common.h"
#include "cdna4_kernels.h"
#include <stdlib.h>
#include <type_traits>

#include "gemm_q_hw.h"            // the GPU-only statements as macros (tools/emul supplies host versions)

// polls (s_sleep between them) a work-group spends waiting for a co-resident partner before it gives up: a few seconds — on a device the caller really owns the partner is
// microseconds away (rounds 2-5 waited 2^26 polls, about a minute, for a word that then never came)
#define CDNA4_SPIN_BOUND (1u << 22)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
__device__ __forceinline__ void glds16(const void *g, void *l_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void_t *)g, (lds_void_t *)l_wave_base, 16, 0, 0);
}
__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }

#define MAGIC2 0x64006400u   // two fp16 1024.0: (MAGIC2 | q) == 1024 + q exactly for q < 1024

// mask / magic held in VGPRs (made opaque to the optimizer once per kernel): `(x & mask) | magic` then selects a single
// v_and_or_b32 — with two literals the compiler must split it into v_and + v_or (VOP3 takes one literal).
struct DqConst {
    uint32_t m4, m8, magic, m4c, magic64;
    __device__ __forceinline__ void init() {
        m4 = 0x000F000Fu; m8 = 0x00FF00FFu; magic = MAGIC2; m4c = 0x00F000F0u; magic64 = 0x54005400u;
        asm volatile("" : "+v"(m4), "+v"(m8), "+v"(magic), "+v"(m4c), "+v"(magic64));
    }
};
__device__ __forceinline__ uint32_t nib(uint32_t x, const DqConst &c) { return (x & c.m4) | c.magic; }
// A nibble can be turned into an exact small fp16 integer WITHOUT shifting it down first when it lies inside the 10-bit
// mantissa, by choosing the exponent that makes its bit position the units place: bits 0-3 under exponent 2^10 give
// 1024+q, bits 4-7 under 2^6 give 64+q (mantissa = q << pos, value = 2^e (1 + mantissa/1024)).  Bits 8-11 and 12-15
// reach into the exponent field and need one shift.  nibq() returns the packed pair and the constant that takes it to
// q - zero exactly.
__device__ __forceinline__ uint32_t nibq(uint32_t x, int pos, const DqConst &c, float zero, half2_t &off) {
    float o; uint32_t v;
    if (pos == 0) { v = (x & c.m4) | c.magic; o = -1024.f - zero; }
    else if (pos == 4) { v = (x & c.m4c) | c.magic64; o = -64.f - zero; }
    else if (pos == 8) { v = ((x >> 8) & c.m4) | c.magic; o = -1024.f - zero; }
    else { v = ((x >> 8) & c.m4c) | c.magic64; o = -64.f - zero; }
    off = half2_t{(half_t)o, (half_t)o};
    return v;
}

// the 8-halves chunk (of the 64-k slice) that MFMA k-step kk / lane-half h consumes
template <int TYPE> __device__ __forceinline__ int chunk_of(int kk, int h) {
    if (QT<TYPE>::KQ) return (kk >> 1) * 4 + 2 * h + (kk & 1);     // low nibbles = k<32, high = k>=32 of the slice
    return 4 * h + kk;                                              // lane-half h owns 32-block h of the slice
}

__device__ __forceinline__ half8_t finish_frag(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, half2_t off, half2_t s, half2_t c) {
    const half2_t r0 = __builtin_elementwise_fma(as_h2(p0) + off, s, c), r1 = __builtin_elementwise_fma(as_h2(p1) + off, s, c);
    const half2_t r2 = __builtin_elementwise_fma(as_h2(p2) + off, s, c), r3 = __builtin_elementwise_fma(as_h2(p3) + off, s, c);
    half8_t f = {r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y};
    return f;
}
__device__ __forceinline__ half2_t splat(float v) { const half_t h = (half_t)v; half2_t r = {h, h}; return r; }

// ------------------------------------------------------------------------------------------------------------
// per-format raw data for one 64-k slice of one weight row, as seen by lane-half h, and the fragment builders.
// `P` is a byte pointer to the row (global) or to the staged superblock (LDS); the address space is inferred.
template <int TYPE> struct Raw;

template <> struct Raw<CDNA4_Q4_K> {
    u32x4 hdr, q;
    template <typename P> __device__ __forceinline__ void load(P blk, int g, int h) { hdr = ld_u32x4(blk); q = ld_u32x4(blk + 16 + 32 * g + 16 * h); }
    // one fragment at a time (software-pipelined schedule): scales once per group, then frag(kk)
    struct Sc { half2_t SL, CL, SH, CH; };
    __device__ __forceinline__ Sc scales(int g) const {
        const uint32_t sc[3] = {hdr.y, hdr.z, hdr.w};
        uint32_t s0, m0, s1, m1;
        k4_scale_min(sc, 2 * g, s0, m0); k4_scale_min(sc, 2 * g + 1, s1, m1);
        const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
        const half_t sl = (half_t)(d * (float)s0), sh = (half_t)(d * (float)s1);
        Sc r; r.SL = half2_t{sl, sl}; r.SH = half2_t{sh, sh};
        r.CL = splat(8.f * (float)sl - dmin * (float)m0); r.CH = splat(8.f * (float)sh - dmin * (float)m1);
        return r;
    }
    __device__ __forceinline__ half8_t frag(int kk, const Sc &z, const DqConst &c) const {
        const half2_t off = {(half_t)-1032.f, (half_t)-1032.f};
        const uint32_t a = (kk & 1) ? q.z : q.x, b = (kk & 1) ? q.w : q.y;
        if (kk < 2) return finish_frag(nib(a, c), nib(a >> 8, c), nib(b, c), nib(b >> 8, c), off, z.SL, z.CL);
        return finish_frag(nib(a >> 4, c), nib(a >> 12, c), nib(b >> 4, c), nib(b >> 12, c), off, z.SH, z.CH);
    }
    // half2 #i (two of the eight k) of fragment kk: the unit the in-wave pipeline drops between two MFMAs
    __device__ __forceinline__ uint32_t pairbits(int kk, int i, const Sc &z, const DqConst &c) const {
        const uint32_t src = (i < 2) ? ((kk & 1) ? q.z : q.x) : ((kk & 1) ? q.w : q.y);
        half2_t off;
        const uint32_t v = nibq(src, (kk >= 2 ? 4 : 0) + ((i & 1) ? 8 : 0), c, 8.f, off);
        const half2_t r = __builtin_elementwise_fma(as_h2(v) + off, kk < 2 ? z.SL : z.SH, kk < 2 ? z.CL : z.CH);
        return __builtin_bit_cast(uint32_t, r);
    }
    __device__ __forceinline__ void frags(int g, int h, half8_t (&f)[4], const DqConst &c) const {
        const uint32_t sc[3] = {hdr.y, hdr.z, hdr.w};
        uint32_t s0, m0, s1, m1;
        k4_scale_min(sc, 2 * g, s0, m0); k4_scale_min(sc, 2 * g + 1, s1, m1);
        const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
        const half_t sl = (half_t)(d * (float)s0), sh = (half_t)(d * (float)s1);
        const half2_t SL = {sl, sl}, SH = {sh, sh};
        const half2_t CL = splat(8.f * (float)sl - dmin * (float)m0), CH = splat(8.f * (float)sh - dmin * (float)m1);
        const half2_t off = {(half_t)-1032.f, (half_t)-1032.f};
        f[0] = finish_frag(nib(q.x, c), nib(q.x >> 8, c), nib(q.y, c), nib(q.y >> 8, c), off, SL, CL);
        f[1] = finish_frag(nib(q.z, c), nib(q.z >> 8, c), nib(q.w, c), nib(q.w >> 8, c), off, SL, CL);
        f[2] = finish_frag(nib(q.x >> 4, c), nib(q.x >> 12, c), nib(q.y >> 4, c), nib(q.y >> 12, c), off, SH, CH);
        f[3] = finish_frag(nib(q.z >> 4, c), nib(q.z >> 12, c), nib(q.w >> 4, c), nib(q.w >> 12, c), off, SH, CH);
    }
    __device__ __forceinline__ void frags(int g, int h, half8_t (&f)[4]) const {
        const uint32_t sc[3] = {hdr.y, hdr.z, hdr.w};
        uint32_t s0, m0, s1, m1;
        k4_scale_min(sc, 2 * g, s0, m0); k4_scale_min(sc, 2 * g + 1, s1, m1);
        const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
        const half_t sl = (half_t)(d * (float)s0), sh = (half_t)(d * (float)s1);
        const half2_t SL = {sl, sl}, SH = {sh, sh};
        const half2_t CL = splat(8.f * (float)sl - dmin * (float)m0), CH = splat(8.f * (float)sh - dmin * (float)m1);
        const half2_t off = {(half_t)-1032.f, (half_t)-1032.f};
        const uint32_t M = 0x000F000Fu;
        f[0] = finish_frag((q.x & M) | MAGIC2, ((q.x >> 8) & M) | MAGIC2, (q.y & M) | MAGIC2, ((q.y >> 8) & M) | MAGIC2, off, SL, CL);
        f[1] = finish_frag((q.z & M) | MAGIC2, ((q.z >> 8) & M) | MAGIC2, (q.w & M) | MAGIC2, ((q.w >> 8) & M) | MAGIC2, off, SL, CL);
        f[2] = finish_frag(((q.x >> 4) & M) | MAGIC2, ((q.x >> 12) & M) | MAGIC2, ((q.y >> 4) & M) | MAGIC2, ((q.y >> 12) & M) | MAGIC2, off, SH, CH);
        f[3] = finish_frag(((q.z >> 4) & M) | MAGIC2, ((q.z >> 12) & M) | MAGIC2, ((q.w >> 4) & M) | MAGIC2, ((q.w >> 12) & M) | MAGIC2, off, SH, CH);
    }
};

template <> struct Raw<CDNA4_Q5_K> {
    u32x4 hdr, qh, q;
    template <typename P> __device__ __forceinline__ void load(P blk, int g, int h) {
        hdr = ld_u32x4(blk); qh = ld_u32x4(blk + 16 + 16 * h); q = ld_u32x4(blk + 48 + 32 * g + 16 * h);
    }
    __device__ __forceinline__ void frags(int g, int h, half8_t (&f)[4], const DqConst &) const { frags(g, h, f); }
    struct Sc { half2_t SL, CL, SH, CH; int bl, bh; };
    __device__ __forceinline__ Sc scales(int g) const {
        const uint32_t sc[3] = {hdr.y, hdr.z, hdr.w};
        uint32_t s0, m0, s1, m1;
        k4_scale_min(sc, 2 * g, s0, m0); k4_scale_min(sc, 2 * g + 1, s1, m1);
        const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
        const half_t sl = (half_t)(d * (float)s0), sh = (half_t)(d * (float)s1);
        Sc r; r.SL = half2_t{sl, sl}; r.SH = half2_t{sh, sh};
        r.CL = splat(16.f * (float)sl - dmin * (float)m0); r.CH = splat(16.f * (float)sh - dmin * (float)m1);
        r.bl = 2 * g; r.bh = 2 * g + 1;
        return r;
    }
    __device__ __forceinline__ half8_t frag(int kk, const Sc &z, const DqConst &) const {
        const half2_t off = {(half_t)-1040.f, (half_t)-1040.f};
        const uint32_t a = (kk & 1) ? q.z : q.x, b = (kk & 1) ? q.w : q.y, ha = (kk & 1) ? qh.z : qh.x, hb = (kk & 1) ? qh.w : qh.y;
        if (kk < 2) return finish_frag(pair(a, ha, 0, z.bl, 0), pair(a, ha, 0, z.bl, 8), pair(b, hb, 0, z.bl, 0), pair(b, hb, 0, z.bl, 8), off, z.SL, z.CL);
        return finish_frag(pair(a, ha, 4, z.bh, 0), pair(a, ha, 4, z.bh, 8), pair(b, hb, 4, z.bh, 0), pair(b, hb, 4, z.bh, 8), off, z.SH, z.CH);
    }
    __device__ __forceinline__ uint32_t pairbits(int kk, int i, const Sc &z, const DqConst &) const {
        const half2_t off = {(half_t)-1040.f, (half_t)-1040.f};
        const uint32_t src = (i < 2) ? ((kk & 1) ? q.z : q.x) : ((kk & 1) ? q.w : q.y);
        const uint32_t hs = (i < 2) ? ((kk & 1) ? qh.z : qh.x) : ((kk & 1) ? qh.w : qh.y);
        const half2_t r = __builtin_elementwise_fma(as_h2(pair(src, hs, kk >= 2 ? 4 : 0, kk >= 2 ? z.bh : z.bl, (i & 1) ? 8 : 0)) + off,
                                                    kk < 2 ? z.SL : z.SH, kk < 2 ? z.CL : z.CH);
        return __builtin_bit_cast(uint32_t, r);
    }
    // value 1024 + nibble + 16*bit for bytes (0,2) [sh=0] or (1,3) [sh=8] of x; bit taken from hq
    static __device__ __forceinline__ uint32_t pair(uint32_t x, uint32_t hq, int nib_shift, int bit, int sh) {
        return ((x >> (nib_shift + sh)) & 0x000F000Fu) | ((((hq >> (bit + sh)) & 0x00010001u) << 4)) | MAGIC2;
    }
    __device__ __forceinline__ void frags(int g, int h, half8_t (&f)[4]) const {
        const uint32_t sc[3] = {hdr.y, hdr.z, hdr.w};
        uint32_t s0, m0, s1, m1;
        k4_scale_min(sc, 2 * g, s0, m0); k4_scale_min(sc, 2 * g + 1, s1, m1);
        const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
        const half_t sl = (half_t)(d * (float)s0), sh = (half_t)(d * (float)s1);
        const half2_t SL = {sl, sl}, SH = {sh, sh};
        const half2_t CL = splat(16.f * (float)sl - dmin * (float)m0), CH = splat(16.f * (float)sh - dmin * (float)m1);
        const half2_t off = {(half_t)-1040.f, (half_t)-1040.f};
        const int bl = 2 * g, bh = 2 * g + 1;
        f[0] = finish_frag(pair(q.x, qh.x, 0, bl, 0), pair(q.x, qh.x, 0, bl, 8), pair(q.y, qh.y, 0, bl, 0), pair(q.y, qh.y, 0, bl, 8), off, SL, CL);
        f[1] = finish_frag(pair(q.z, qh.z, 0, bl, 0), pair(q.z, qh.z, 0, bl, 8), pair(q.w, qh.w, 0, bl, 0), pair(q.w, qh.w, 0, bl, 8), off, SL, CL);
        f[2] = finish_frag(pair(q.x, qh.x, 4, bh, 0), pair(q.x, qh.x, 4, bh, 8), pair(q.y, qh.y, 4, bh, 0), pair(q.y, qh.y, 4, bh, 8), off, SH, CH);
        f[3] = finish_frag(pair(q.z, qh.z, 4, bh, 0), pair(q.z, qh.z, 4, bh, 8), pair(q.w, qh.w, 4, bh, 0), pair(q.w, qh.w, 4, bh, 8), off, SH, CH);
    }
};

// Q6_K slice s4 of a superblock: half n = s4>>1, p = s4&1 selects quads (2p, 2p+1) = (q1,q2) or (q3,q4)
template <> struct Raw<CDNA4_Q6_K> {
    uint32_t la[4], lv[4], hq[4]; float ds[2];
    template <typename P> __device__ __forceinline__ void load(P blk, int s4, int h) {
        const int n = s4 >> 1, p = s4 & 1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            la[i] = reinterpret_cast<const u32_a2 *>(blk + 64 * n + 16 * h + 4 * i)->v;
            lv[i] = reinterpret_cast<const u32_a2 *>(blk + 64 * n + 32 + 16 * h + 4 * i)->v;
            hq[i] = reinterpret_cast<const u32_a2 *>(blk + 128 + 32 * n + 16 * h + 4 * i)->v;
        }
        const float d = h2f(*reinterpret_cast<const uint16_t *>(blk + 208));
        ds[0] = d * (float)*reinterpret_cast<const int8_t *>(blk + 192 + 8 * n + h + 2 * (2 * p));
        ds[1] = d * (float)*reinterpret_cast<const int8_t *>(blk + 192 + 8 * n + h + 2 * (2 * p + 1));
    }
    static __device__ __forceinline__ uint32_t pair(uint32_t x, uint32_t h, int nib_shift, int bits_shift, int sh) {
        return ((x >> (nib_shift + sh)) & 0x000F000Fu) | ((((h >> (bits_shift + sh)) & 0x00030003u) << 4)) | MAGIC2;
    }
    __device__ __forceinline__ void frags(int s4, int h, half8_t (&f)[4]) const {
        const int p = s4 & 1, ns = 4 * p;                       // nibble shift: q1,q2 low / q3,q4 high
        const half2_t off = {(half_t)-1056.f, (half_t)-1056.f}, zero = {(half_t)0.f, (half_t)0.f};
        const half2_t S0 = splat(ds[0]), S1 = splat(ds[1]);
        const int b0 = 2 * (2 * p), b1 = 2 * (2 * p + 1);      // qh bit offset of the two quads
        f[0] = finish_frag(pair(la[0], hq[0], ns, b0, 0), pair(la[0], hq[0], ns, b0, 8), pair(la[1], hq[1], ns, b0, 0), pair(la[1], hq[1], ns, b0, 8), off, S0, zero);
        f[1] = finish_frag(pair(la[2], hq[2], ns, b0, 0), pair(la[2], hq[2], ns, b0, 8), pair(la[3], hq[3], ns, b0, 0), pair(la[3], hq[3], ns, b0, 8), off, S0, zero);
        f[2] = finish_frag(pair(lv[0], hq[0], ns, b1, 0), pair(lv[0], hq[0], ns, b1, 8), pair(lv[1], hq[1], ns, b1, 0), pair(lv[1], hq[1], ns, b1, 8), off, S1, zero);
        f[3] = finish_frag(pair(lv[2], hq[2], ns, b1, 0), pair(lv[2], hq[2], ns, b1, 8), pair(lv[3], hq[3], ns, b1, 0), pair(lv[3], hq[3], ns, b1, 8), off, S1, zero);
    }
};

// Q4_0 / Q8_0: the 64-k slice is two 32-blocks, lane-half h owns block h.  `blk` already points at it.
template <> struct Raw<CDNA4_Q4_0> {
    uint32_t q[4]; float d;
    template <typename P> __device__ __forceinline__ void load(P blk, int, int) {
        d = h2f(*reinterpret_cast<const uint16_t *>(blk));
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = reinterpret_cast<const u32_a2 *>(blk + 2 + 4 * i)->v;
    }
    __device__ __forceinline__ void frags(int, int, half8_t (&f)[4]) const {
        const half2_t off = {(half_t)-1032.f, (half_t)-1032.f}, zero = {(half_t)0.f, (half_t)0.f}, S = splat(d);
        const uint32_t M = 0x000F000Fu;
        f[0] = finish_frag((q[0] & M) | MAGIC2, ((q[0] >> 8) & M) | MAGIC2, (q[1] & M) | MAGIC2, ((q[1] >> 8) & M) | MAGIC2, off, S, zero);
        f[1] = finish_frag((q[2] & M) | MAGIC2, ((q[2] >> 8) & M) | MAGIC2, (q[3] & M) | MAGIC2, ((q[3] >> 8) & M) | MAGIC2, off, S, zero);
        f[2] = finish_frag(((q[0] >> 4) & M) | MAGIC2, ((q[0] >> 12) & M) | MAGIC2, ((q[1] >> 4) & M) | MAGIC2, ((q[1] >> 12) & M) | MAGIC2, off, S, zero);
        f[3] = finish_frag(((q[2] >> 4) & M) | MAGIC2, ((q[2] >> 12) & M) | MAGIC2, ((q[3] >> 4) & M) | MAGIC2, ((q[3] >> 12) & M) | MAGIC2, off, S, zero);
    }
};
template <> struct Raw<CDNA4_Q8_0> {
    uint32_t q[8]; float d;
    template <typename P> __device__ __forceinline__ void load(P blk, int, int) {
        d = h2f(*reinterpret_cast<const uint16_t *>(blk));
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = reinterpret_cast<const u32_a2 *>(blk + 2 + 4 * i)->v ^ 0x80808080u;   // int8 -> biased u8
    }
    __device__ __forceinline__ void frags(int, int, half8_t (&f)[4]) const {
        const half2_t off = {(half_t)-1152.f, (half_t)-1152.f}, zero = {(half_t)0.f, (half_t)0.f}, S = splat(d);
        const uint32_t M = 0x00FF00FFu;
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            f[kk] = finish_frag((q[2 * kk] & M) | MAGIC2, ((q[2 * kk] >> 8) & M) | MAGIC2, (q[2 * kk + 1] & M) | MAGIC2, ((q[2 * kk + 1] >> 8) & M) | MAGIC2, off, S, zero);
    }
};

// ---- repacked (16-byte-aligned) forms of Q4_0 / Q8_0 / Q6_K, staged through LDS by the 8-wave kernel -------------
// `row` = this row's staged bytes of one 128-k stage in LDS; g_local = which 64-k group of the stage (the wave's khalf);
// scales(g) takes the group's index inside the 256-weight superblock (compile-time after inlining).
template <> struct Raw<CDNA4_Q4_0R> {
    u32x4 hdr, q;                                                    // hdr = fp16 d[8]
    template <typename P> __device__ __forceinline__ void load(P row, int g_local, int h) { hdr = ld_u32x4(row); q = ld_u32x4(row + 16 + 32 * g_local + 16 * h); }
    struct Sc { half2_t SL, CL, SH, CH; };
    __device__ __forceinline__ Sc scales(int g) const {
        const uint32_t dw = g == 0 ? hdr.x : (g == 1 ? hdr.y : (g == 2 ? hdr.z : hdr.w));
        const half2_t d2 = as_h2(dw), zero = {(half_t)0.f, (half_t)0.f};
        Sc r; r.SL = half2_t{d2.x, d2.x}; r.SH = half2_t{d2.y, d2.y}; r.CL = zero; r.CH = zero;
        return r;
    }
    __device__ __forceinline__ uint32_t pairbits(int kk, int i, const Sc &z, const DqConst &c) const {
        const uint32_t src = (i < 2) ? ((kk & 1) ? q.z : q.x) : ((kk & 1) ? q.w : q.y);
        half2_t off;
        const uint32_t v = nibq(src, (kk >= 2 ? 4 : 0) + ((i & 1) ? 8 : 0), c, 8.f, off);
        const half2_t r = (as_h2(v) + off) * (kk < 2 ? z.SL : z.SH);                     // d * (q - 8), one rounding, as the 32-block kernel
        return __builtin_bit_cast(uint32_t, r);
    }
    __device__ __forceinline__ void frags(int g, int, half8_t (&f)[4], const DqConst &c) const {
        const Sc z = scales(g);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { const u32x4 w = {pairbits(kk, 0, z, c), pairbits(kk, 1, z, c), pairbits(kk, 2, z, c), pairbits(kk, 3, z, c)}; f[kk] = __builtin_bit_cast(half8_t, w); }
    }
};

template <> struct Raw<CDNA4_Q8_0R> {
    u32x4 hdr, q0, q1; int hh;                                       // lane-half h owns the 32-block h of the 64-k group
    template <typename P> __device__ __forceinline__ void load(P row, int g_local, int h) {
        hdr = ld_u32x4(row); q0 = ld_u32x4(row + 16 + 64 * g_local + 32 * h); q1 = ld_u32x4(row + 32 + 64 * g_local + 32 * h); hh = h;
        q0.x ^= 0x80808080u; q0.y ^= 0x80808080u; q0.z ^= 0x80808080u; q0.w ^= 0x80808080u;      // int8 -> biased u8
        q1.x ^= 0x80808080u; q1.y ^= 0x80808080u; q1.z ^= 0x80808080u; q1.w ^= 0x80808080u;
    }
    struct Sc { half2_t S; };
    __device__ __forceinline__ Sc scales(int g) const {
        const uint32_t dw = g == 0 ? hdr.x : (g == 1 ? hdr.y : (g == 2 ? hdr.z : hdr.w));
        const half2_t d2 = as_h2(dw);
        const half_t d = hh ? d2.y : d2.x;
        Sc r; r.S = half2_t{d, d};
        return r;
    }
    __device__ __forceinline__ uint32_t pairbits(int kk, int i, const Sc &z, const DqConst &c) const {
        const half2_t off = {(half_t)-1152.f, (half_t)-1152.f};
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const uint32_t src = w[2 * kk + (i >> 1)];
        const half2_t r = (as_h2(((src >> ((i & 1) ? 8 : 0)) & c.m8) | c.magic) + off) * z.S;
        return __builtin_bit_cast(uint32_t, r);
    }
    __device__ __forceinline__ void frags(int g, int, half8_t (&f)[4], const DqConst &c) const {
        const Sc z = scales(g);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { const u32x4 w = {pairbits(kk, 0, z, c), pairbits(kk, 1, z, c), pairbits(kk, 2, z, c), pairbits(kk, 3, z, c)}; f[kk] = __builtin_bit_cast(half8_t, w); }
    }
};

// staged row of half n: [d, pad : 16][scales[16] : 16][ql[64n .. 64n+63] : 64][qh[32n .. 32n+31] : 32][pad : 16]
template <> struct Raw<CDNA4_Q6_KR> {
    u32x4 la, lv, hq, sc; uint32_t dd; int hh;
    template <typename P> __device__ __forceinline__ void load(P row, int, int h) {
        dd = *reinterpret_cast<const uint32_t *>(row); sc = ld_u32x4(row + 16);
        la = ld_u32x4(row + 32 + 16 * h); lv = ld_u32x4(row + 64 + 16 * h); hq = ld_u32x4(row + 96 + 16 * h); hh = h;
    }
    struct Sc { half2_t S0, S1; int ns, b0, b1; };
    __device__ __forceinline__ Sc scales(int g) const {              // g = 2n + p: half n, quads (2p, 2p+1); scale bytes 8n + 4p + h (+2)
        const uint32_t dw = g == 0 ? sc.x : (g == 1 ? sc.y : (g == 2 ? sc.z : sc.w));
        const int s0 = (int)(int8_t)(dw >> (8 * hh)), s1 = (int)(int8_t)(dw >> (8 * hh + 16));
        const float d = h2f(dd & 0xFFFF);
        const int p = g & 1;
        Sc r; r.S0 = splat(d * (float)s0); r.S1 = splat(d * (float)s1); r.ns = 4 * p; r.b0 = 2 * (2 * p); r.b1 = 2 * (2 * p + 1);
        return r;
    }
    __device__ __forceinline__ uint32_t pairbits(int kk, int i, const Sc &z, const DqConst &) const {
        const half2_t off = {(half_t)-1056.f, (half_t)-1056.f};
        const int wi = 2 * (kk & 1) + (i >> 1), sh = (i & 1) ? 8 : 0;
        const u32x4 lsrc = kk < 2 ? la : lv;
        const uint32_t x = wi == 0 ? lsrc.x : (wi == 1 ? lsrc.y : (wi == 2 ? lsrc.z : lsrc.w));
        const uint32_t hb = wi == 0 ? hq.x : (wi == 1 ? hq.y : (wi == 2 ? hq.z : hq.w));
        const int bits = kk < 2 ? z.b0 : z.b1;
        const uint32_t v = ((x >> (z.ns + sh)) & 0x000F000Fu) | ((((hb >> (bits + sh)) & 0x00030003u) << 4)) | MAGIC2;
        const half2_t r = (as_h2(v) + off) * (kk < 2 ? z.S0 : z.S1);
        return __builtin_bit_cast(uint32_t, r);
    }
    __device__ __forceinline__ void frags(int g, int, half8_t (&f)[4], const DqConst &c) const {
        const Sc z = scales(g);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { const u32x4 w = {pairbits(kk, 0, z, c), pairbits(kk, 1, z, c), pairbits(kk, 2, z, c), pairbits(kk, 3, z, c)}; f[kk] = __builtin_bit_cast(half8_t, w); }
    }
};

// ---- "staged" forms: same fragment builders as the repacked ones, but the stage row in LDS only carries THIS stage's scales
template <> struct Raw<CDNA4_Q4_0S> : Raw<CDNA4_Q4_0R> {
    __device__ __forceinline__ Sc scales(int g) const {                // hdr.x / hdr.y = the two fp16 d of 64-k group 0 / 1 of the stage
        const half2_t d2 = as_h2((g & 1) ? hdr.y : hdr.x), zero = {(half_t)0.f, (half_t)0.f};
        Sc r; r.SL = half2_t{d2.x, d2.x}; r.SH = half2_t{d2.y, d2.y}; r.CL = zero; r.CH = zero;
        return r;
    }
    __device__ __forceinline__ void frags(int g, int, half8_t (&f)[4], const DqConst &c) const {
        const Sc z = scales(g);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { const u32x4 w = {pairbits(kk, 0, z, c), pairbits(kk, 1, z, c), pairbits(kk, 2, z, c), pairbits(kk, 3, z, c)}; f[kk] = __builtin_bit_cast(half8_t, w); }
    }
};
template <> struct Raw<CDNA4_Q8_0S> : Raw<CDNA4_Q8_0R> {
    __device__ __forceinline__ Sc scales(int g) const {
        const half2_t d2 = as_h2((g & 1) ? hdr.y : hdr.x);
        const half_t d = hh ? d2.y : d2.x;
        Sc r; r.S = half2_t{d, d};
        return r;
    }
    __device__ __forceinline__ void frags(int g, int, half8_t (&f)[4], const DqConst &c) const {
        const Sc z = scales(g);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { const u32x4 w = {pairbits(kk, 0, z, c), pairbits(kk, 1, z, c), pairbits(kk, 2, z, c), pairbits(kk, 3, z, c)}; f[kk] = __builtin_bit_cast(half8_t, w); }
    }
};
template <> struct Raw<CDNA4_Q6_KS> : Raw<CDNA4_Q6_KR> {
    __device__ __forceinline__ Sc scales(int g) const {                // sc.x / sc.y: scale bytes 4p + h (+2) of this half's eight scales
        const uint32_t dw = (g & 1) ? sc.y : sc.x;
        const int s0 = (int)(int8_t)(dw >> (8 * hh)), s1 = (int)(int8_t)(dw >> (8 * hh + 16));
        const float d = h2f(dd & 0xFFFF);
        const int p = g & 1;
        Sc r; r.S0 = splat(d * (float)s0); r.S1 = splat(d * (float)s1); r.ns = 4 * p; r.b0 = 2 * (2 * p); r.b1 = 2 * (2 * p + 1);
        return r;
    }
    __device__ __forceinline__ void frags(int g, int, half8_t (&f)[4], const DqConst &c) const {
        const Sc z = scales(g);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { const u32x4 w = {pairbits(kk, 0, z, c), pairbits(kk, 1, z, c), pairbits(kk, 2, z, c), pairbits(kk, 3, z, c)}; f[kk] = __builtin_bit_cast(half8_t, w); }
    }
};

// What one loader lane of k_gemm_kq_w12 does for a staged format: lane (row, gl) of a stage = the gl-th 64-k group of one
// weight row.  load(): the group's ORIGINAL bytes into registers (2-byte-aligned dword loads); store(): re-laid into the
// row's slot of the LDS stage.  `src` points at the row's bytes of this 128-k stage.
template <int TYPE> struct WDirect { static constexpr bool value = false; struct Regs {}; };
template <> struct WDirect<CDNA4_Q4_0S> {
    static constexpr bool value = true;
    static constexpr int STAGE_SRC = 4 * 18;
    struct Regs { uint32_t w[9]; };
    __device__ static __forceinline__ Regs load(const uint8_t *src, int gl) {
        Regs r; const uint8_t *s = src + gl * 36;
#pragma unroll
        for (int i = 0; i < 9; i++) r.w[i] = ld_u32_a2(s + 4 * i);
        return r;
    }
    __device__ static __forceinline__ void store(const Regs &r, uint8_t *row, int gl) {
        // bytes: [dA:2][qsA:16][dB:2][qsB:16]; qsA dword i straddles w[i], w[i+1]; qsB dword i = w[5+i]
        uint32_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = (r.w[i] >> 16) | (r.w[i + 1] << 16); b[i] = r.w[5 + i]; }
        const uint32_t dd = (r.w[0] & 0xFFFFu) | (r.w[4] & 0xFFFF0000u);                    // dA | dB << 16
        *reinterpret_cast<uint32_t *>(row + 4 * gl) = dd;
        u32x4 lo, hi;                                                                       // weights 0..15 / 16..31 of both blocks: low nibble = block A (k < 32), high = block B
        lo.x = (a[0] & 0x0F0F0F0Fu) | ((b[0] & 0x0F0F0F0Fu) << 4); lo.y = (a[1] & 0x0F0F0F0Fu) | ((b[1] & 0x0F0F0F0Fu) << 4);
        lo.z = (a[2] & 0x0F0F0F0Fu) | ((b[2] & 0x0F0F0F0Fu) << 4); lo.w = (a[3] & 0x0F0F0F0Fu) | ((b[3] & 0x0F0F0F0Fu) << 4);
        hi.x = ((a[0] >> 4) & 0x0F0F0F0Fu) | (b[0] & 0xF0F0F0F0u); hi.y = ((a[1] >> 4) & 0x0F0F0F0Fu) | (b[1] & 0xF0F0F0F0u);
        hi.z = ((a[2] >> 4) & 0x0F0F0F0Fu) | (b[2] & 0xF0F0F0F0u); hi.w = ((a[3] >> 4) & 0x0F0F0F0Fu) | (b[3] & 0xF0F0F0F0u);
        *reinterpret_cast<u32x4 *>(row + 16 + 32 * gl) = lo; *reinterpret_cast<u32x4 *>(row + 32 + 32 * gl) = hi;
    }
};
template <> struct WDirect<CDNA4_Q8_0S> {
    static constexpr bool value = true;
    static constexpr int STAGE_SRC = 4 * 34;
    struct Regs { uint32_t w[17]; };
    __device__ static __forceinline__ Regs load(const uint8_t *src, int gl) {
        Regs r; const uint8_t *s = src + gl * 68;
#pragma unroll
        for (int i = 0; i < 17; i++) r.w[i] = ld_u32_a2(s + 4 * i);
        return r;
    }
    __device__ static __forceinline__ void store(const Regs &r, uint8_t *row, int gl) {
        // bytes: [dA:2][qA:32][dB:2][qB:32]; qA dword i straddles w[i], w[i+1]; qB dword i = w[9+i]
        uint32_t a[8];
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = (r.w[i] >> 16) | (r.w[i + 1] << 16);
        *reinterpret_cast<uint32_t *>(row + 4 * gl) = (r.w[0] & 0xFFFFu) | (r.w[8] & 0xFFFF0000u);
        u32x4 *dst = reinterpret_cast<u32x4 *>(row + 16 + 64 * gl);
        dst[0] = u32x4{a[0], a[1], a[2], a[3]}; dst[1] = u32x4{a[4], a[5], a[6], a[7]};
        dst[2] = u32x4{r.w[9], r.w[10], r.w[11], r.w[12]}; dst[3] = u32x4{r.w[13], r.w[14], r.w[15], r.w[16]};
    }
};
template <> struct WDirect<CDNA4_Q6_KS> {
    static constexpr bool value = true;
    static constexpr int STAGE_SRC = 0;                                // a stage is half n of a 210-byte superblock: offsets below
    struct Regs { uint32_t ql[8], qh[4], x[2]; };
    // src = the superblock; part = which 128-weight half; lane gl takes bytes [32 gl, 32 gl + 32) of ql, [16 gl, +16) of qh,
    // and gl = 0: d, gl = 1: the half's eight scales
    __device__ static __forceinline__ Regs load(const uint8_t *sbk, int part, int gl) {
        Regs r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.ql[i] = ld_u32_a2(sbk + 64 * part + 32 * gl + 4 * i);
#pragma unroll
        for (int i = 0; i < 4; i++) r.qh[i] = ld_u32_a2(sbk + 128 + 32 * part + 16 * gl + 4 * i);
        if (gl == 0) { r.x[0] = ld_u16(sbk + 208); r.x[1] = 0; }
        else { r.x[0] = ld_u32_a2(sbk + 192 + 8 * part); r.x[1] = ld_u32_a2(sbk + 196 + 8 * part); }
        return r;
    }
    __device__ static __forceinline__ void store(const Regs &r, uint8_t *row, int gl) {
        *reinterpret_cast<u32x2 *>(row + 16 * gl) = u32x2{r.x[0], r.x[1]};                    // [d, pad] at 0, [scales[8]] at 16
        u32x4 *ql = reinterpret_cast<u32x4 *>(row + 32 + 32 * gl);
        ql[0] = u32x4{r.ql[0], r.ql[1], r.ql[2], r.ql[3]}; ql[1] = u32x4{r.ql[4], r.ql[5], r.ql[6], r.ql[7]};
        *reinterpret_cast<u32x4 *>(row + 96 + 16 * gl) = u32x4{r.qh[0], r.qh[1], r.qh[2], r.qh[3]};
    }
};


struct gemm_params {
    const uint8_t *W; int64_t w_row_bytes;
    const half_t *xh; int64_t xh_row;   // xh: k-panel-major fp16 image (see quantize_act.hip); xh_row unused
    float *Y; int64_t y_row;
    int M, K, B, splitk, tiles_m, tiles_b;
    int sb_split;                                       // hand-off: superblocks [0, sb_split) -> ks=0, the rest -> ks=1
    int xchg_l2;                                        // split-K exchange through the XCD's L2 (partners co-located) instead of write-through
    int tune;                                           // experiment bits from CDNA4_TUNE (bit0: static s_setprio 1 for the khalf-1 waves)
    float *partial; unsigned *flags;   // split-K = 2 exchange: exported half tiles [tile][ks][64][128]; one flag per (tile, ks): 0 = idle, 16 | XCC id = published (cleared by its READER)
    unsigned long long *trace;   // profiling builds only (k_gemm_kq_w8<TYPE, true>): per-phase s_memtime stamps of block 0
    // grouped MUL_MAT_ID (k_gemm_kq_t64<.., IDS = true>; appended last: no other kernel's argument offsets move): the activation image
    // holds the (token, slot) rows SORTED BY EXPERT, every expert's run starting on a 128-row boundary; activation tile t belongs to
    // expert tile_expert[t] (< 0: unused tile, the work-group exits), its weights start at W + expert * w_expert_bytes, and image row r
    // is output row row_dst[r] (< 0: padding row, not stored)
    const int32_t *tile_expert; const int32_t *row_dst; int64_t w_expert_bytes;
    // k_gemm_kq_t64 only (appended last): the MUL_MAT's tail, applied to every element in the store that produces it (epilogue.h)
    cdna4_epilogue epi;
    // k_gemm_kq_t64<.., FQ> only (appended last): the ONE-LAUNCH step — the fp32 activation rows (xf, rows xf_row elements apart) are quantized to the Q8_K-rounded
    // fp16 image `xh` by the work-groups themselves in front of a grid barrier (gbar: its words, see cdna4_grid_barrier)
    const float *xf; int64_t xf_row; unsigned *gbar;
    // k_gemm_kq_t64<.., 128, IDS> only (appended last; either may be null): tile_order[j] = the image tile the j-th group of tiles_m work-groups takes (fullest first),
    // tile_nfrag[t] = how many of tile t's four 32-row fragments hold rows (k_moe_plan, quantize_act.hip)
    const int32_t *tile_order; const int32_t *tile_nfrag;
    // every kernel with an exchange that WAITS for a co-resident work-group (appended last): a word of pinned host memory (or null) that a wait which ran into its bound
    // sets — the host turns it into an error status and demotes the library to the non-waiting routes (cdna4_gemm_fault_word, ggml_cdna4_device_fault)
    unsigned *fault;
};
// a spin on a partner work-group gave up: the tile is poisoned (NaN) AND the host is told (system-scope store to the pinned fault word) — never NaN with status 0
__device__ __forceinline__ void cdna4_report_fault(unsigned *fault, unsigned code) {
    if (fault) __hip_atomic_store(fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Grid-wide barrier of a launch whose work-groups are ALL resident (the launcher's condition; one thread per work-group calls it, behind a __syncthreads()
// that follows every wave's `s_waitcnt vmcnt(0)` on its write-through stores).  Two levels so that no word takes more than 32 + 8 arrivals: work-group b arrives
// at the counter of group b & 7 (dispatch puts block b on XCD b % 8 — a speed assumption only), the last of a group at the top counter, the last of those bumps every
// group's generation word, which is what the others poll (relaxed agent-scope loads, s_sleep between polls).  Counters are back at zero when the launch ends and the
// generations only ever grow: no per-launch state on the host, so the launch replays from a HIP graph.  Words (each on its own 128-byte line): gb[32 g] arrivals of
// group g, gb[256] top, gb[288 + 32 g] generation of group g.  Returns false when the partners never arrived (a grid that was not resident after all): the
// caller poisons its output.  Data published before the barrier must have been stored write-through (sc1) and is read behind it with sc1 loads
// (MI355X_MICROARCH.md "inter-workgroup visibility": {sc1 stores + vmcnt(0) + flag | sc1 loads} needs no fence on either side).
__device__ __forceinline__ bool cdna4_grid_barrier(unsigned *gb, unsigned nblk) {
    const unsigned g = blockIdx.x & 7u, ngrp = nblk < 8u ? nblk : 8u, members = (nblk - g + 7u) / 8u;
    unsigned *gen = gb + 288 + 32 * g;
    unsigned gen0 = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(__HIPCC__)
    asm volatile("" : "+v"(gen0) :: "memory");                        // the generation is in a register BEFORE this work-group arrives (two relaxed operations on different words may pass each other)
#endif
    if (__hip_atomic_fetch_add(gb + 32 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
        __hip_atomic_store(gb + 32 * g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(gb + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1u) {
            __hip_atomic_store(gb + 256, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned k = 0; k < ngrp; k++) __hip_atomic_fetch_add(gb + 288 + 32 * k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    unsigned spins = 0;
    while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen0 && ++spins < CDNA4_SPIN_BOUND) __builtin_amdgcn_s_sleep(1);
    return spins < CDNA4_SPIN_BOUND;
}
// (A one-word form — every work-group adds a weight, the weights sum to 2^32, everybody polls the word until it is zero again — was built and measured in round 5: 256
// arrivals and 255 pollers on ONE address cost +4 us at the headline shape and +10 us at 4096 x 11008 x 512 against this two-level form; profiles/r05/onelaunch_ab.txt.)

template <int N> __device__ __forceinline__ void wait_vmcnt() { CDNA4_WAIT_VM(N); }

template <int TYPE, int SKG> struct WStage {        // which 16-B pieces of a superblock a stage needs
    static constexpr int NP = QT<TYPE>::BYTES / 16;
    static constexpr int HDR = TYPE == CDNA4_Q5_K ? 3 : 1;             // header (+ qh) pieces, needed by every stage
    static constexpr int NPH = SKG == 4 ? NP : HDR + 2 * SKG;          // pieces per row per stage (2 per 64-k group)
    __device__ static __forceinline__ int src_piece(int p, int part) { return (SKG == 4 || p < HDR) ? p : HDR + 2 * SKG * part + (p - HDR); }
};
// repacked forms (8-wave kernel only, SKG = 2).  Row strides of 9 pieces (144 B = 36 dwords) keep the per-lane b128 reads
// of 32 consecutive rows conflict-free; Q6_KR pads its 8 useful pieces with a dummy 9th for that reason.
template <> struct WStage<CDNA4_Q4_0R, 2> : WStage<CDNA4_Q4_K, 2> {};
template <> struct WStage<CDNA4_Q8_0R, 2> {                            // [d[8]] + 4 pieces per 64-k group
    static constexpr int NPH = 9;
    __device__ static __forceinline__ int src_piece(int p, int part) { return p < 1 ? p : 1 + 8 * part + (p - 1); }
};
template <> struct WStage<CDNA4_Q6_KR, 2> {                            // [d][scales] + ql(half n) 4 pieces + qh(half n) 2 pieces + dummy
    static constexpr int NPH = 9;
    __device__ static __forceinline__ int src_piece(int p, int part) { return p < 2 ? p : (p < 6 ? 2 + 4 * part + (p - 2) : (p < 8 ? 10 + 2 * part + (p - 6) : 0)); }
};
// staged forms: the LDS stage row has the repacked forms' size; the loader lanes write it, nothing is DMA'd for the weights
template <> struct WStage<CDNA4_Q4_0S, 2> { static constexpr int NPH = 5; __device__ static __forceinline__ int src_piece(int, int) { return 0; } };
template <> struct WStage<CDNA4_Q8_0S, 2> { static constexpr int NPH = 9; __device__ static __forceinline__ int src_piece(int, int) { return 0; } };
template <> struct WStage<CDNA4_Q6_KS, 2> { static constexpr int NPH = 9; __device__ static __forceinline__ int src_piece(int, int) { return 0; } };


// host-side helpers shared by the launchers (defined in gemm_q_mfma.hip)
void *cdna4_gemm_scratch(size_t bytes, int kind);      // per-device scratch, zero-filled when (re)allocated; kind 0: split-K exchange, 1: repacked weights
int cdna4_gemm_cu_count();
int cdna4_launch_gemm_t64(const cdna4_gemm_args &a, int tm, int splitk, hipStream_t st);     // gemm_q_t64.hip: 64(m) x 128(b) wave tiles; tm 0 / 128 / 256, splitk 0 / 1 / 2
int cdna4_gemm_shared_device();                         // 1: the AUTO routes must not choose an exchange that spins on a co-resident partner (gemm_q_mfma.hip)
int cdna4_gemm_set_shared_device(int shared);
int cdna4_gemm_coresident_cus();                        // CUs that may be assumed to hold a grid all at once: the device's, or 0 in shared mode
unsigned *cdna4_gemm_fault_word();                      // the device-visible fault word of the current device (pinned host memory), or nullptr
int cdna4_gemm_take_fault(bool clear);                  // the fault code kernels reported since the last clear (0: none); a non-zero answer also switches the library to shared-device mode
bool cdna4_gemm_lds_supported(const cdna4_gemm_args &a);                                     // gemm_q_lds.hip: weights dequantized into fp16 LDS tiles, 256-wide activation tile
int cdna4_launch_gemm_lds(const cdna4_gemm_args &a, int tm, int splitk, hipStream_t st, int form = 0);     //   tm 0 / 128 / 256, splitk 0 = choose; form 1 = k_gemm_w4 (one wave per SIMD); form 2 = k_gemm_r8 (32 x 256 wave tiles)
bool cdna4_gemm_r8_preferred(const cdna4_gemm_args &a);                                      //   AUTO takes k_gemm_r8 for this call (large grids of 256 x 256 tiles)
