// gemm_q_x4l.hip — EXPERIMENTAL (not selected by default, not yet run on a GPU): Q4_K prefill GEMM with a 256(m) x 128(b)
// work-group tile, FOUR compute waves (one per SIMD) and FOUR loader waves.
//
// Why this shape (ablations of k_gemm_kq_w12): in the 128x128 kernels the activation LDS-DMA is the largest
// removable term of a stage, the barriers and the unpack are free, and `mfma_valu` shows that ONE wave per SIMD sustains
// ~33.5 pipe-cycles per MFMA with one pairbits (and_or, pk_add, pk_fma) per MFMA beside it.  So:
//   * compute wave mg owns rows [64 mg, 64 mg + 64) as two 32-row blocks for all 128 b and ALL 128 k of a stage: every
//     activation fragment read from LDS feeds two MFMAs (half the activation DMA bytes and ds_reads per MFMA of the 128x128
//     tile), there is no intra-work-group K split and therefore no LDS reduction in the epilogue, and a stage is 64 MFMAs
//     per wave between two barriers (2048 pipe-cycles instead of 1024);
//   * exactly one pairbits per MFMA: k-step t's 8 MFMAs (2 row blocks x 4 b blocks) build the 8 half2 of k-step t+1's two
//     weight fragments, across group and stage boundaries (the stage barrier sits between k-steps 6 and 7);
//   * the loader waves issue every LDS-DMA piece and turn the 6-bit scales / mins into the fp16 (s, c) table (as in
//     k_gemm_kq_w12); the superblock header is never staged — the stage row is 64 B of nibbles (16-B chunks XOR-swizzled by
//     (row >> 2) & 3 on the source address: conflict-free ds_read_b128 at a 64-byte row stride) plus 16 B of table;
//   * ring of 3 stages x 52 KB; split-K S in {1, 2, 4} with the symmetric exchange of k_gemm_kq_x2.
// Same arithmetic per weight as every other Q4_K kernel here (Raw<Q4_K>::pairbits with the loaders' table), same k order.
#include "gemm_q_common.h"
#include "gemm_q_x4l_hw.h"

// NMB = 32-row blocks per compute wave: 2 = the 256 x 128 tile described above; 1 = a 128 x 128 tile with the same structure (one
// activation fragment per MFMA again, but still no K split inside the work-group, one wave per SIMD and loader waves) for
// the grids that are too small for 256-row tiles (the headline 4096 x 4096 x 512: 128 tiles x split-K 2).
// NCW = compute waves: 4 (one per SIMD) or, with NMB = 1, 8 (two per SIMD, 139 registers each: 12 waves fit the 168-register
// budget) on a 256 x 128 tile — the partner wave hides a wave's LDS latency as in the shipped kernels, at the price of one
// activation ds_read per MFMA again; the activation DMA per MFMA stays halved.
// TYPE = Q4_K, or Q5_K (128 x 128 form only: its stage row also carries the superblock's 32 bytes of fifth bits, and 3 x 60 KB
// does not fit): the fifth bits sit in their own area, two 16-byte chunks per row, chunk XOR-swizzled by (row >> 3) & 1.
template <int S, int NMB, int NCW, int TYPE = CDNA4_Q4_K>
__global__ __launch_bounds__((NCW + 4) * 64) void k_gemm_q4k_x4l(const gemm_params p) {
    constexpr bool Q5 = TYPE == CDNA4_Q5_K;
    static_assert(TYPE == CDNA4_Q4_K || (Q5 && NMB * NCW == 4), "Q5_K: 128-row tile only");
    constexpr int BNF = 4, TB = 128, TM = 32 * NMB * NCW, NST = 3, NTR = TM / 128;   // NTR: table rows per loader lane
    constexpr int RS = 256, XS = TB * RS;                 // activations: 128 rows x 256 B
    constexpr int BLK = QT<TYPE>::BYTES;                  // 144 / 176
    constexpr int QOFF = Q5 ? 48 : 16;                    // byte offset of the nibbles in a superblock (Q5_K: header, 32 B of fifth bits, nibbles)
    constexpr int WQS = TM * 64, QHS = Q5 ? TM * 32 : 0, TS = TM * 16;   // nibbles: TM rows x 64 B; fifth bits: TM x 32 B; table: TM x 2 groups x 8 B
    constexpr int ST = XS + WQS + QHS + TS;               // 53,248 (Q4_K 256 rows) / 43,008 (Q4_K 128 rows) / 47,104 (Q5_K 128 rows)
    constexpr int NHL = Q5 ? TM / 128 : 0;                // fifth-bit wave-pieces per loader wave and stage
    constexpr int NXL = 8, NWL = TM / 64, NLD = NXL + NWL + NHL; // DMA wave-pieces per loader wave and stage
    constexpr int SMEM = NST * ST > 128 * 1024 ? NST * ST : 128 * 1024;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM];
    __shared__ int xchg_failed;

    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= NCW;
    const int mg = is_loader ? wave - NCW : wave;         // compute: its row group; loader: its index (0..3)
    const int nblk = gridDim.x;
    int L = blockIdx.x;
    if ((nblk & 7) == 0) L = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int tile_b = L % p.tiles_b; L /= p.tiles_b;
    const int ks = L % S, tile_m = L / S;
    const int m0 = tile_m * TM, b0 = tile_b * TB;
    const int nsb_all = p.K / 256, nsb_base = nsb_all / S, nsb_rem = nsb_all % S;
    const int nsb = nsb_base + (ks < nsb_rem ? 1 : 0), sb0 = ks * nsb_base + (ks < nsb_rem ? ks : nsb_rem);
    const int nstage = nsb * 2;                           // >= 4 (the launcher guarantees two superblocks per work-group)

    const uint32_t lds0 = X4L_LDS_BASE(smem);
    const char *const xbase = (const char *)p.xh + ((int64_t)sb0 * 2 * p.B + b0) * 256;
    const char *const wbase = (const char *)p.W + (int64_t)m0 * p.w_row_bytes + (int64_t)sb0 * BLK;

    floatx16 acc[NMB][BNF];
#pragma unroll
    for (int mb = 0; mb < NMB; mb++)
#pragma unroll
        for (int i = 0; i < BNF; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mb][i][r] = 0.f;

    if (tid == 0) xchg_failed = 0;

    if (is_loader) {
        // ================================================================ loader waves
        uint32_t xvoff[NXL], wvoff[NWL], qhvoff[NHL > 0 ? NHL : 1], hoff[NTR];
#pragma unroll
        for (int i = 0; i < NXL; i++) {                    // activation wave-piece q = mg + 4 i: LDS bytes [1024 q, 1024 q + 1024) of the slot
            const int pc = (mg + 4 * i) * 64 + lane, row = pc >> 4, c = (pc & 15) ^ (row & 15);
            xvoff[i] = (uint32_t)(min(b0 + row, p.B - 1) - b0) * 256u + c * 16;
        }
#pragma unroll
        for (int i = 0; i < NWL; i++) {                    // nibble wave-piece q = mg + 4 i: rows 16 q .. 16 q + 15, four 16-B chunks each
            const int pc = (mg + 4 * i) * 64 + lane, row = pc >> 2, c = (pc & 3) ^ ((row >> 2) & 3);
            wvoff[i] = (uint32_t)(min(m0 + row, p.M - 1) - m0) * (uint32_t)p.w_row_bytes + (uint32_t)QOFF + c * 16;
        }
#pragma unroll
        for (int i = 0; i < NHL; i++) {                    // fifth-bit wave-piece q = mg + 4 i: rows 32 q .. 32 q + 31, two 16-B chunks each
            const int pc = (mg + 4 * i) * 64 + lane, row = pc >> 1, c = (pc & 1) ^ ((row >> 3) & 1);
            qhvoff[i] = (uint32_t)(min(m0 + row, p.M - 1) - m0) * (uint32_t)p.w_row_bytes + 16u + c * 16;
        }
        const int lidx = (mg << 6) | lane, lrow = lidx >> 1, lgl = lidx & 1;      // table: rows lrow (and lrow + 128), group lgl of the stage
#pragma unroll
        for (int r = 0; r < NTR; r++) hoff[r] = (uint32_t)(min(m0 + lrow + 128 * r, p.M - 1) - m0) * (uint32_t)p.w_row_bytes;

        auto dma16 = [&](const char *sbase, uint32_t voff, uint32_t lds_addr) __attribute__((always_inline)) {
            X4L_DMA16(voff, sbase, lds_addr);
        };
        auto issue = [&](int st, int slot) __attribute__((always_inline)) {          // stage st = (superblock st >> 1, half st & 1)
            const uint32_t l = lds0 + slot * ST;
            const char *xs = xbase + (int64_t)st * p.B * 256, *wsb = wbase + (int64_t)(st >> 1) * BLK, *ws = wsb + (st & 1) * 64;
#pragma unroll
            for (int i = 0; i < NXL; i++) dma16(xs, xvoff[i], l + (mg + 4 * i) * 1024);
#pragma unroll
            for (int i = 0; i < NWL; i++) dma16(ws, wvoff[i], l + XS + (mg + 4 * i) * 1024);
#pragma unroll
            for (int i = 0; i < NHL; i++) dma16(wsb, qhvoff[i], l + XS + WQS + (mg + 4 * i) * 1024);      // (the same 32 bytes for both stages of a superblock)
        };
        struct Hdr { u32x4 r[NTR]; };
        // superblock headers of this lane's table rows, SYNCHRONOUS (see gemm_q_x4l_hw.h).  It is called before a block's DMA
        // pieces are issued, when at most the previous block's pieces — issued a whole stage earlier — can still be in flight, so
        // the vmcnt(0) inside costs the header's own latency, in a loader wave, once per superblock.
        auto hload = [&](Hdr &v, int sbr) __attribute__((always_inline)) {
            const char *sb = wbase + (int64_t)sbr * BLK;
            if constexpr (NTR == 2) X4L_GLOAD16x2_SYNC(v.r[0], v.r[1], hoff[0], hoff[1], sb);
            else X4L_GLOAD16_SYNC(v.r[0], hoff[0], sb);
        };
        auto tab_store = [&](const Hdr &hd, int part, int slot) __attribute__((always_inline)) {    // the arithmetic of Raw<Q4_K>::scales()
            const int g = part * 2 + lgl;
#pragma unroll
            for (int r = 0; r < NTR; r++) {
                const u32x4 hdr = hd.r[r];
                int s0, mn0, s1, mn1;
                k4_scale_min_rt(hdr.y, hdr.z, hdr.w, 2 * g, s0, mn0); k4_scale_min_rt(hdr.y, hdr.z, hdr.w, 2 * g + 1, s1, mn1);
                const float d = h2f(hdr.x & 0xFFFF), dmin = h2f(hdr.x >> 16);
                const half_t sl = (half_t)(d * (float)s0), sh = (half_t)(d * (float)s1);
                constexpr float ZERO = Q5 ? 16.f : 8.f;             // Raw<TYPE>::scales(): the constant that folds the -zero offset
                const half_t cl = (half_t)(ZERO * (float)sl - dmin * (float)mn0), ch = (half_t)(ZERO * (float)sh - dmin * (float)mn1);
                u32x2 e; e.x = __builtin_bit_cast(uint32_t, half2_t{sl, cl}); e.y = __builtin_bit_cast(uint32_t, half2_t{sh, ch});
                *reinterpret_cast<u32x2 *>(smem + slot * ST + XS + WQS + QHS + ((lrow + 128 * r) * 2 + lgl) * 8) = e;
            }
        };
        // prologue: stages 0, 1, 2 -> slots 0, 1, 2
        Hdr h0, hcur;
        hload(h0, 0); hload(hcur, 1);
        issue(0, 0); issue(1, 1); issue(2, 2);
        tab_store(h0, 0, 0); tab_store(h0, 1, 1); tab_store(hcur, 0, 2);
        X4L_WAIT_VM(2 * NLD);                                                   // stage 0 has landed
        X4L_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // barrier B_s (between k-steps 6 and 7 of stage s, s < nstage - 1): stage s + 1 landed, slot of stage s free -> stage s + 3
        int slot = 0;
        Hdr hnext = hcur;
        for (int s = 0; s + 1 < nstage; s++) {
            if (s + 2 < nstage) X4L_WAIT_VM(NLD); else X4L_WAIT_VM(0);           // only DMA pieces are ever outstanding here
            X4L_WAIT_LGKM0();                                                    // the table written last block is in LDS
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int st = s + 3;
            if (st < nstage) {
                if ((st & 1) == 0) hcur = hnext;                                 // stage st opens superblock st / 2: its header arrived one block ago
                const bool need = ((st + 1) & 1) == 0 && st + 1 < nstage;
                tab_store(hcur, st & 1, slot);
                if (need) hload(hnext, (st + 1) >> 1);                           // the next superblock's header (used from the next block on)
                issue(st, slot);
            }
            slot = slot == 2 ? 0 : slot + 1;
        }
    } else {
        // ================================================================ compute waves
        DqConst dq; dq.init();
        const int xrow_off = j * RS, xswz = j & 15;
        Raw<TYPE> rq[NMB][2];                             // [row block mb][64-k group g of the stage]: only .q is used
        typename Raw<TYPE>::Sc z[NMB][2];
        half8_t xa[2][BNF];                               // activation fragments of one k-step, double-buffered
        uint32_t cur[NMB][4], nxt[NMB][4];
        auto read_xa = [&](int slot_, int t, int buf) __attribute__((always_inline)) {          // k-step t = (g, kk) of the stage
            const uint8_t *xs = smem + slot_ * ST + xrow_off;
            const int coff = (((t >> 2) * 8 + chunk_of<TYPE>(t & 3, h)) ^ xswz) << 4;
#pragma unroll
            for (int bf = 0; bf < BNF; bf++) xa[buf][bf] = *reinterpret_cast<const half8_t *>(xs + bf * 32 * RS + coff);
        };
        auto read_w = [&](int slot_, int g, int part) __attribute__((always_inline)) {          // nibbles (+ fifth bits) + table entry of group g of the stage (half `part` of its superblock)
#pragma unroll
            for (int mb = 0; mb < NMB; mb++) {
                const int row = mg * 32 * NMB + mb * 32 + j;
                rq[mb][g].q = *reinterpret_cast<const u32x4 *>(smem + slot_ * ST + XS + row * 64 + (((2 * g + h) ^ ((row >> 2) & 3)) << 4));
                if constexpr (Q5) {
                    rq[mb][g].qh = *reinterpret_cast<const u32x4 *>(smem + slot_ * ST + XS + WQS + row * 32 + ((h ^ ((row >> 3) & 1)) << 4));
                    z[mb][g].bl = 2 * (part * 2 + g); z[mb][g].bh = z[mb][g].bl + 1;   // bit planes of the superblock-global 64-k group
                }
                const u32x2 te = *reinterpret_cast<const u32x2 *>(smem + slot_ * ST + XS + WQS + QHS + (row * 2 + g) * 8);
                const half2_t lo = as_h2(te.x), hi = as_h2(te.y);
                z[mb][g].SL = half2_t{lo.x, lo.x}; z[mb][g].CL = half2_t{lo.y, lo.y}; z[mb][g].SH = half2_t{hi.x, hi.x}; z[mb][g].CH = half2_t{hi.y, hi.y};
            }
        };
        // the 4 NMB MFMAs of k-step t with `between(n)` after the n-th
        auto mfma8 = [&](int buf, auto &&between) __attribute__((always_inline)) {
            half8_t wfk[NMB];
#pragma unroll
            for (int mb = 0; mb < NMB; mb++) { const u32x4 cw = {cur[mb][0], cur[mb][1], cur[mb][2], cur[mb][3]}; wfk[mb] = __builtin_bit_cast(half8_t, cw); }
#pragma unroll
            for (int mb = 0; mb < NMB; mb++)
#pragma unroll
                for (int bf = 0; bf < BNF; bf++) {
                    __builtin_amdgcn_sched_barrier(0);
                    acc[mb][bf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[buf][bf], wfk[mb], acc[mb][bf], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    between(mb * 4 + bf);
                }
#pragma unroll
            for (int mb = 0; mb < NMB; mb++)
#pragma unroll
                for (int i = 0; i < 4; i++) cur[mb][i] = nxt[mb][i];
        };
        // prologue: stage 0 landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_w(0, 0, 0); read_xa(0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < NMB; mb++)
#pragma unroll
            for (int i = 0; i < 4; i++) { cur[mb][i] = rq[mb][0].pairbits(0, i, z[mb][0], dq); nxt[mb][i] = 0; }
        int slot = 0;
        for (int s = 0; s < nstage; s++) {
            const int slot1 = slot == 2 ? 0 : slot + 1;
            const bool has_next = s + 1 < nstage;
#pragma unroll
            for (int t = 0; t < 7; t++) {                  // k-steps 0..6: one pairbits of k-step t + 1 after every MFMA
                read_xa(slot, t + 1, (t + 1) & 1);
                if (t == 1) read_w(slot, 1, s & 1);        // the second group's nibbles / table: needed from k-step 3 on
                mfma8(t & 1, [&](int n) __attribute__((always_inline)) {
                    const int t1 = t + 1, g1 = t1 >> 2, mb = n >> 2, i = n & 3;
                    nxt[mb][i] = rq[mb][g1].pairbits(t1 & 3, i, z[mb][g1], dq);
                });
            }
            if (has_next) {
                X4L_WAIT_LGKM0();                                                // this wave's last reads of `slot` have returned
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                read_w(slot1, 0, (s + 1) & 1); read_xa(slot1, 0, 0);             // stage s + 1, published by the barrier
                __builtin_amdgcn_sched_barrier(0);
            }
            // k-step 7: the 4 NMB half2 of the next stage's first fragments go behind the LAST 2 NMB MFMAs, two each (their LDS
            // reads were issued just above and need a few MFMAs to return)
            // (on the last stage the same arithmetic runs on stale registers and its result is never used: no branch in the stream)
            mfma8(1, [&](int n) __attribute__((always_inline)) {
                if (n >= 2 * NMB) {
#pragma unroll
                    for (int e = 0; e < 2; e++) { const int q = (n - 2 * NMB) * 2 + e, mb = q >> 2, i = q & 3; nxt[mb][i] = rq[mb][0].pairbits(0, i, z[mb][0], dq); }
                }
            });
            slot = slot1;
        }
    }

    // ---- epilogue: (no K-half sum: a compute wave holds complete partial sums) symmetric S-way exchange, [b][m] tile through
    //      LDS, 1-KB output rows stored by all 512 threads
    __syncthreads();
    const int tile_id = tile_m * p.tiles_b + tile_b;
    constexpr int NBF = BNF / S;                                        // accumulator b-blocks kept per work-group
    if constexpr (S > 1) {
        constexpr int PF4 = NCW * NMB * NBF * 4 * 64;                      // float4 per (tile, dst, src) partial: [mg][mb][bfl][q4][lane]
        float4 *pbase = reinterpret_cast<float4 *>(p.partial) + (size_t)tile_id * S * S * PF4;
        auto exchange = [&](auto KS) __attribute__((always_inline)) {
            constexpr int me = decltype(KS)::value;
            if (!is_loader) {
#pragma unroll
                for (int d = 0; d < S; d++) {
                    if (d == me) continue;
                    float4 *dst = pbase + (size_t)(d * S + me) * PF4;
                    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, PF4 * 16, 0x00020000);
#pragma unroll
                    for (int mb = 0; mb < NMB; mb++)
#pragma unroll
                        for (int bl = 0; bl < NBF; bl++)
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++) {
                                const int bf = d * NBF + bl;
                                const float4 v = make_float4(acc[mb][bf][4 * q4], acc[mb][bf][4 * q4 + 1], acc[mb][bf][4 * q4 + 2], acc[mb][bf][4 * q4 + 3]);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (((((mg * NMB + mb) * NBF + bl) * 4 + q4) * 64) + lane) * 16, 0, 16);
                            }
                }
                X4L_WAIT_VM(0);
            }
            __syncthreads();
            if (tid == 0) {
                // same word format as the other kernels' flags (launch tag << 4 | XCC id): they share the flag scratch
                __hip_atomic_store(p.flags + tile_id * S + me, p.epoch << 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int o = 0; o < S; o++) {
                    if (o == me) continue;
                    unsigned spins = 0;
                    while (((__hip_atomic_load(p.flags + tile_id * S + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ (p.epoch << 4)) >> 4) != 0 && ++spins < (1u << 26)) __builtin_amdgcn_s_sleep(1);
                    if (spins >= (1u << 26)) xchg_failed = 1;             // a partner never showed up (not co-resident): fail LOUDLY
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (!is_loader) {
#pragma unroll
                for (int o = 0; o < S; o++) {                            // fixed order: deterministic
                    if (o == me) continue;
                    const float4 *src = pbase + (size_t)(me * S + o) * PF4;
                    float4 t[NMB * NBF * 4];
#pragma unroll
                    for (int i = 0; i < NMB * NBF * 4; i++) t[i] = src[((mg * NMB * NBF * 4) + i) * 64 + lane];
                    if (xchg_failed) {                                      // NaN tile instead of a silently wrong sum
#pragma unroll
                        for (int i = 0; i < NMB * NBF * 4; i++) t[i].x = __builtin_nanf("");
                    }
#pragma unroll
                    for (int mb = 0; mb < NMB; mb++)
#pragma unroll
                        for (int bl = 0; bl < NBF; bl++)
#pragma unroll
                            for (int q4 = 0; q4 < 4; q4++) {
                                const float4 v = t[(mb * NBF + bl) * 4 + q4];
                                const int bf = me * NBF + bl;
                                acc[mb][bf][4 * q4] += v.x; acc[mb][bf][4 * q4 + 1] += v.y; acc[mb][bf][4 * q4 + 2] += v.z; acc[mb][bf][4 * q4 + 3] += v.w;
                            }
                }
            }
        };
        if (ks == 0) exchange(std::integral_constant<int, 0>{});
        else if (ks == 1) exchange(std::integral_constant<int, 1>{});
        else if (S > 2 && ks == 2) exchange(std::integral_constant<int, (S > 2 ? 2 : 0)>{});
        else if (S > 2) exchange(std::integral_constant<int, (S > 2 ? 3 : 0)>{});
    }
    const int row_lo = (S > 1) ? ks * (128 / S) : 0;
    constexpr int NROWS = 128 / S, CLD = TM;
    float *ctile = reinterpret_cast<float *>(smem);                       // the ring is dead: [NROWS][256] fp32 <= 128 KB
    __syncthreads();
    if (!is_loader) {
#pragma unroll
        for (int mb = 0; mb < NMB; mb++)
#pragma unroll
            for (int bf = 0; bf < BNF; bf++) {
                if (bf * 32 < row_lo || bf * 32 >= row_lo + NROWS) continue;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int bl = bf * 32 + (r & 3) + 8 * (r >> 2) + 4 * h - row_lo;           // C/D layout: row = (r&3) + 8*(r>>2) + 4*(lane>>5)
                    ctile[bl * CLD + mg * 32 * NMB + mb * 32 + j] = acc[mb][bf][r];
                }
            }
    }
    __syncthreads();
    {
        constexpr int F4 = TM / 4, RPP = 512 / F4;                          // float4 per row, rows per pass (the first 512 threads store)
        const int c4 = tid % F4, r0 = tid / F4;
#pragma unroll
        for (int pass = 0; pass < (tid < 512 ? NROWS / RPP : 0); pass++) {
            const int bl = pass * RPP + r0, b = b0 + row_lo + bl, m = m0 + c4 * 4;
            const float4 v = *reinterpret_cast<const float4 *>(ctile + bl * CLD + c4 * 4);
            if (b < p.B && m < p.M) {
                float *dst = p.Y + (int64_t)b * p.y_row + m;
                if (m + 3 < p.M && ((((uintptr_t)dst) & 15) == 0)) *reinterpret_cast<float4 *>(dst) = v;
                else { const float e[4] = {v.x, v.y, v.z, v.w}; for (int t = 0; t < 4 && m + t < p.M; t++) dst[t] = e[t]; }
            }
        }
    }
}

// launcher: Q4_K, K % 256 == 0, 16-byte-aligned rows; splitk 0 = the widest split whose work-groups are all co-resident and
// keep two superblocks each.  Returns -1 (with a message) if the shape does not fit.
template <int NMB, int NCW, int TYPE = CDNA4_Q4_K>
static int launch_x4l(const cdna4_gemm_args &a, int splitk, hipStream_t st) {
    constexpr int TM = 32 * NMB * NCW;
    if (a.type != TYPE || a.K % 256 || ((((uintptr_t)a.W | (uintptr_t)a.w_row_bytes) & 15) != 0))
        return cdna4_set_error_msg("gemm_q: the loader-wave kernel takes 16-byte-aligned Q4_K / Q5_K rows with K % 256 == 0");
    gemm_params p{};
    p.W = a.W; p.w_row_bytes = a.w_row_bytes; p.xh = (const half_t *)a.xh; p.xh_row = a.xh_row_elems;
    p.Y = a.Y; p.y_row = a.y_row_elems; p.M = a.M; p.K = a.K; p.B = a.B;
    p.tiles_m = (a.M + TM - 1) / TM; p.tiles_b = (a.B + 127) / 128;
    const int ntiles = p.tiles_m * p.tiles_b, nsb = a.K / 256, cus = cdna4_gemm_cu_count();
    int S = splitk;
    if (S <= 0) {
        S = 1;
        for (int c = 4; c >= 2; c >>= 1) if (ntiles * c <= cus && nsb / c >= 2) { S = c; break; }
    }
    if (S != 1 && S != 2 && S != 4) return cdna4_set_error_msg("gemm_q: the 256x128 kernels split K 1, 2 or 4 ways");
    if (nsb / S < 2) return cdna4_set_error_msg("gemm_q: the 4+4-wave kernel needs two superblocks of K per work-group");
    if (S > 1 && ntiles * S > cus) return cdna4_set_error_msg("gemm_q: the split-K exchange needs every work-group resident");
    p.splitk = S;
    if (S > 1) {
        const size_t pbytes = (size_t)ntiles * S * (TM * 128 * 4);
        char *sc = (char *)cdna4_gemm_scratch(pbytes + (size_t)ntiles * S * 4 + 256, 0);
        if (!sc) return cdna4_set_error_msg("gemm_q: cannot allocate split-K scratch");
        p.partial = (float *)sc; p.flags = (unsigned *)(sc + pbytes);
        p.epoch = cdna4_gemm_next_epoch();
    }
    const dim3 grid(ntiles * S);
    if (S == 1) hipLaunchKernelGGL((k_gemm_q4k_x4l<1, NMB, NCW, TYPE>), grid, dim3((NCW + 4) * 64), 0, st, p);
    else if (S == 2) hipLaunchKernelGGL((k_gemm_q4k_x4l<2, NMB, NCW, TYPE>), grid, dim3((NCW + 4) * 64), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_q4k_x4l<4, NMB, NCW, TYPE>), grid, dim3((NCW + 4) * 64), 0, st, p);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
// form 0: 256 x 128 tile, 4 compute waves x 64 rows; 1: 128 x 128 tile, 4 x 32 rows; 2: 256 x 128 tile, 8 compute waves x 32 rows
int cdna4_launch_gemm_q4k_x4l(const cdna4_gemm_args &a, int splitk, int form, hipStream_t st) {
    if (a.type == CDNA4_Q5_K) return form == 1 ? launch_x4l<1, 4, CDNA4_Q5_K>(a, splitk, st) : cdna4_set_error_msg("gemm_q: Q5_K runs on the 128x128 form of the loader-wave kernel only");
    return form == 1 ? launch_x4l<1, 4>(a, splitk, st) : (form == 2 ? launch_x4l<1, 8>(a, splitk, st) : launch_x4l<2, 4>(a, splitk, st));
}
