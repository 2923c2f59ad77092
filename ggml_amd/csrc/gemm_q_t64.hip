// gemm_q_t64.hip — launcher of k_gemm_kq_t64 (gemm_kq_t64.inc): the prefill GEMM with 64(m) x 128(b) wave tiles.
// Replaces, for Q4_K at B > 8, what ggml_compute_forward_mul_mat does after the activations are quantized
// (/root/reference/src/ggml-cpu/ggml-cpu.c:7510-7605).
#include "gemm_q_common.h"
#include <stdlib.h>
#include "gemm_q_hw.h"
#include "gemm_kq_t64.inc"

// The one-launch step (k_gemm_kq_t64<.., FQ>: quantize -> grid barrier -> multiply): would a launch of `nblk` work-groups take it for these arguments?
//  * the caller handed over the fp32 rows (a.xf; 16-byte aligned rows) next to the image workspace,
//  * every work-group is resident at once — one per CU (132 / 152 KB of LDS) on a device the caller owns (not ggml_cdna4_set_shared_device), and the grid fills more
//    than half the chip (fewer work-groups would quantize longer than the launch they save),
//  * 32-bit byte offsets reach the whole image.    CDNA4_NO_FUSEQ=1 turns it off (A/B, and the two-launch reference of the parity tests).
static bool t64_fuses_quantizer(const cdna4_gemm_args &a, int nblk) {
    static const bool off = getenv("CDNA4_NO_FUSEQ") && atoi(getenv("CDNA4_NO_FUSEQ")) != 0;
    if (off || !a.xf || cdna4_gemm_shared_device() || a.type != CDNA4_Q4_K) return false;      // (Q4_0R meets Q8_0 activations: another quantizer)
    if ((((uintptr_t)a.xf) | (uintptr_t)(a.xf_row_elems * 4)) & 15) return false;
    if ((int64_t)a.B * a.K * 2 + 32768 >= ((int64_t)1 << 31) || (int64_t)a.B * (a.K / 256) >= ((int64_t)1 << 30)) return false;
    const int cus = cdna4_gemm_cu_count();
    if (!(nblk <= cus && nblk * 2 > cus)) return false;
    // ... and the quantizer's share is ONE pass of the work-groups' 32 sixteen-lane groups, at least half full.  Measured on MI355X (profiles/r05/onelaunch_ab.txt, us per call,
    // two launches vs one, same box): 4096x4096x512 (one full pass) 33.0-33.6 vs 32.1-33.0 on three boxes — ahead by 0.4-0.7; 8192x4096x512 (one pass) 45.8-46.9 vs 46.3: level;
    // 4096x11008x512 (2.7 passes: three serial load -> quantize -> write-through rounds) 59.7-61.9 vs 61.4-63.0: BEHIND by 1-1.5; 4096x4096x128 (a quarter pass) 23.5 vs 24.4: behind.
    // Why the gain is this small: in-launch, quantizing costs 3.9 us and the grid barrier 1.2-2.4 us (timing-only ablations) — a chain of memory round trips (fp32 load ->
    // write-through store acknowledged -> arrival -> release -> poll -> first activation DMA) as long as the quantizer's own launch + the kernel boundary it replaces (5.7 us).
    const int64_t units = (int64_t)a.B * (a.K / 256), pass = (int64_t)nblk * 32;
    return units <= pass && units * 2 > pass;
}

// tile rows and split of the launch AUTO (or the given tm / splitk) takes: shared by the launcher and the route probe (`fq`: the quantizer rides inside)
struct t64_plan { int tm, splitk, ntiles; bool ticketed2, fq; };
static int t64_make_plan(const cdna4_gemm_args &a, int tm, int splitk, t64_plan &pl);

bool cdna4_gemm_t64_fuses_quantizer(const cdna4_gemm_args &a, int tm, int splitk) {
    t64_plan pl;
    return t64_make_plan(a, tm, splitk, pl) == 0 && pl.fq;
}

// tile rows (0 = choose) and split-K (0 = choose; 1 or 2) -> launch.  Returns 0, or a negative status with the error text set.
int cdna4_launch_gemm_t64(const cdna4_gemm_args &a, int tm, int splitk, hipStream_t st) {
    t64_plan pl;
    const int prc = t64_make_plan(a, tm, splitk, pl);
    if (prc) return prc;
    tm = pl.tm; splitk = pl.splitk;
    const bool ticketed2 = pl.ticketed2, fq = pl.fq;
    const int cus = cdna4_gemm_cu_count(), nsb = a.K / 256;
    const int tiles_b = (a.B + 127) / 128;
    const int tiles_m = (a.M + tm - 1) / tm, ntiles = tiles_m * tiles_b;
    (void)cus;
    gemm_params p{};
    p.W = a.W; p.w_row_bytes = a.w_row_bytes; p.xh = (const half_t *)a.xh; p.xh_row = a.xh_row_elems;
    p.Y = a.Y; p.y_row = a.y_row_elems; p.M = a.M; p.K = a.K; p.B = a.B; p.splitk = splitk;
    p.tiles_m = tiles_m; p.tiles_b = tiles_b;
    p.epi = a.epi;
    if (splitk >= 2 || fq) {
        // exchange slots: [tile][work-group of the tile][wave] x 16 KB at most; flag words: hand-off flags [tile][ks] in the first 32 KB,
        // deep-split ticket counters [tile] behind them, the grid barrier's words (FQ) at 48 KB; all zero when idle and reset by their last user — no per-launch state on the
        // host: graph-capturable.  Scratch kind 2: the older kernels' areas never mix with these.
        // The words live at a FIXED place (the first 64 KB) so that no shape's exchange slots ever overlay another shape's flags.
        const size_t pbytes = splitk >= 2 ? (size_t)ntiles * splitk * 8 * 16384 : 0, fbytes = 65536;
        if ((size_t)ntiles * 8 > 32768) return cdna4_set_error_msg("gemm_t64: too many tiles for the split-K flag area");
        char *sc = (char *)cdna4_gemm_scratch(fbytes + pbytes, 2);
        if (!sc) return cdna4_set_error_msg("gemm_t64: cannot allocate split-K scratch");
        p.flags = (unsigned *)sc; p.gbar = (unsigned *)(sc + 49152);
        if (fq || (splitk == 2 && !ticketed2)) p.fault = cdna4_gemm_fault_word();       // (the launches that wait: the grid barrier, the hand-off)
        if (splitk >= 2) {
            p.partial = (float *)(sc + fbytes);
            p.sb_split = (nsb + 1) / 2;
            if (ticketed2) p.tune = 2;
            const int nb = ntiles * 2;                                                 // partners share an XCD iff the XCD-aware remap is active and
            p.xchg_l2 = (splitk == 2 && (nb & 7) == 0 && ((nb >> 3) % (tiles_b * 2)) == 0) ? 1 : 0;   // each XCD's slice holds whole (tile_b x ks) groups
        }
    }
    if (fq) {
        p.xf = a.xf; p.xf_row = a.xf_row_elems;
        const int fq_abl = getenv("CDNA4_FQ_ABL") ? atoi(getenv("CDNA4_FQ_ABL")) : 0;      // timing-only, read per call: 4 = no grid barrier, 8 = no quantizer (the image of an earlier call is multiplied)
        p.tune |= fq_abl & 12;
    }
    const dim3 grid(ntiles * splitk);
#ifdef CDNA4_ABLATIONS
    // gemm_bench_abl: variant bits 16+ pick a timing-only instantiation (gemm_kq_t64.inc: ABL)
    p.trace = (unsigned long long *)cdna4_debug_trace;
    const int abl = a.variant >> 16;
#define T64_ABL(A) if (abl == (A)) { if (tm == 128) hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 128, false, (A)>), grid, dim3(512), 0, st, p); else hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 256, false, (A)>), grid, dim3(512), 0, st, p); CDNA4_CHECK_LAUNCH(); return 0; }
    T64_ABL(1) T64_ABL(3) T64_ABL(4) T64_ABL(8) T64_ABL(15) T64_ABL(32) T64_ABL(256)
    // the one-launch step with block 0's milestones (ggml_cdna4_debug_trace set, no variant): the instrumented twin of the FQ form
    if (fq && p.trace && !abl && p.epi.bias == nullptr && p.epi.act == 0 && p.epi.resid == nullptr) { hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 128, false, 256, false, true>), grid, dim3(512), 0, st, p); CDNA4_CHECK_LAUNCH(); return 0; }
    if (abl) return cdna4_set_error_msg("gemm_t64: ablation not instantiated");
#endif
    const bool tail = p.epi.bias != nullptr || p.epi.act != 0 || p.epi.resid != nullptr;
    if (a.type == CDNA4_Q4_0R) {                                         // (a resident Q4_0R image: the same kernels, the {d, 0} sub-block constants)
        if (tail) {
            if (tm == 128) hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_0R, 128, false, 0, true>), grid, dim3(512), 0, st, p);
            else hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_0R, 256, false, 0, true>), grid, dim3(512), 0, st, p);
        } else if (tm == 128) hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_0R, 128>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_0R, 256>), grid, dim3(512), 0, st, p);
        CDNA4_CHECK_LAUNCH();
        return 0;
    }
    if (fq) {                                                            // (128-row tiles only: t64_make_plan)
        if (tail) hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 128, false, 0, true, true>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 128, false, 0, false, true>), grid, dim3(512), 0, st, p);
    } else if (tail) {
        if (tm == 128) hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 128, false, 0, true>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 256, false, 0, true>), grid, dim3(512), 0, st, p);
    } else if (tm == 128) hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 128>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 256>), grid, dim3(512), 0, st, p);
    CDNA4_CHECK_LAUNCH();
    return 0;
}

static int t64_make_plan(const cdna4_gemm_args &a, int tm, int splitk, t64_plan &pl) {
    if (a.type != CDNA4_Q4_K && a.type != CDNA4_Q4_0R) return cdna4_set_error_msg("gemm_t64: Q4_K (or Q4_0 as its resident Q4_0R image) only");
    if (a.K % 256 || a.K < 256) return cdna4_set_error_msg("gemm_t64: K must be a whole number of superblocks");
    if ((((uintptr_t)a.W | (uintptr_t)a.w_row_bytes) & 15) || ((uintptr_t)a.xh & 15)) return cdna4_set_error_msg("gemm_t64: weight rows and the activation image must be 16-byte aligned");
    const int cus = cdna4_gemm_cu_count(), nsb = a.K / 256;
    const int tiles_b = (a.B + 127) / 128;
    // 256-row tiles move half the activation bytes per MFMA and need no K reduction across four waves: they pay as soon as the grid holds ONE of them per CU.
    // Round 4, one box, us per call, 128- vs 256-row tiles (profiles/r04/t64_tiles.txt): 16384x8192x512 137.8 vs 125.4, 16384x4096x512 76.7 vs 68.5,
    // 4096x4096x2048 78.2 vs 69.1, 8192x8192x1024 138.1 vs 125.8, 8192x4096x1024 76.6 vs 68.5, 16384x11008x512 178.9 vs 164.8 (9-12 %); C5 (two per CU) 239 vs
    // 265 since round 2.  Half a tile per CU (8192x8192x512: 71.6 vs 68.9, round 3) is a loss: CUs idle.  (Until round 4 the rule asked for two per CU.)
    // Between whole rounds the better-filled last round wins: 24576x4096x512 (1.5 tiles of 256 rows per CU: 2 rounds, against 3 full rounds of 128-row tiles) 111.1 vs
    // 108.0; 3/4 of a tile per CU (both forms fill 3/4 of the chip once) 12288x4096x512 62.8 vs 54.4, 12288x8192x512 113.3 vs 98.5, 6144x4096x1024 64.5 vs 55.6
    // (profiles/r04/t64_tiles_075.txt).  Rule: 256-row tiles when their round efficiency is within 10 % of the 128-row tiles' (they are ~10 % faster per flop).
    if (tm <= 0) {
        const int t256 = ((a.M + 255) / 256) * tiles_b, t128 = ((a.M + 127) / 128) * tiles_b;
        auto eff = [&](int n) { return (double)n / (double)(((n + cus - 1) / cus) * cus); };
        // (from 3/4 of a tile per CU: what was measured — 0.75 and above are wins, exactly 0.5 is a loss, the band between them was never run: ADVICE r4)
        tm = (t256 * 4 >= cus * 3 && eff(t256) * 1.10 >= eff(t128)) ? 256 : 128;
    }
    if (tm != 128 && tm != 256) return cdna4_set_error_msg("gemm_t64: tile rows are 128 or 256");
    const int tiles_m = (a.M + tm - 1) / tm, ntiles = tiles_m * tiles_b;
    // split-K: 2 = the hand-off between the two co-resident work-groups of a tile; 4 / 8 = deep split for grids far below the chip
    // (a few activation rows over a tall matrix), every work-group >= 2 superblocks, summed in fixed order by the last one to arrive
    bool ticketed2 = false;                                              // split in two with the ticketed sum instead of the (spinning) hand-off
    if (splitk <= 0) {
        splitk = (ntiles * 2 <= cus && nsb >= 2) ? 2 : 1;
        // AUTO splits in two with the TICKETED sum (each work-group parks its finished blocks, the last of a tile's two to arrive adds both in the order ks = 0, 1;
        // nobody waits, no co-residency assumed) — since round 4 the default: same-box, kernel only, hand-off vs ticketed (profiles/r04/handoff_vs_ticketed.txt):
        // 4096x4096x512 26.3-26.7 vs 26.7, 4096x11008x512 52.6-52.7 vs 52.1-52.2, 8192x4096x512 39.0 vs 38.8, 2048x4096x512 20.5-21.1 vs 20.6-20.8 us: level,
        // and bit-identical for even superblock counts (two fp32 terms).  The spinning hand-off stays behind an explicit splitk = 2 (and CDNA4_T64_HANDOFF=1).
        static const bool handoff_env = getenv("CDNA4_T64_HANDOFF") && atoi(getenv("CDNA4_T64_HANDOFF")) != 0;
        if (splitk == 2 && (!handoff_env || cdna4_gemm_shared_device())) { if (tm == 128) ticketed2 = true; else splitk = 1; }
        if (tm == 128 && ntiles * 4 <= cus && nsb >= 8) splitk = 4;
        if (tm == 128 && ntiles * 8 <= cus && nsb >= 16) splitk = 8;
    }
    if (splitk == 2 && tm == 128 && cdna4_gemm_shared_device()) ticketed2 = true;      // shared device: an EXPLICIT split in two takes the ticketed sum as well
    if (splitk != 1 && splitk != 2 && splitk != 4 && splitk != 8) return cdna4_set_error_msg("gemm_t64: split-K is 1, 2, 4 or 8");
    if (splitk == 2 && (ntiles * 2 > cus || nsb < 2)) return cdna4_set_error_msg("gemm_t64: the split-K hand-off needs every work-group resident and two superblocks");
    if (splitk > 2 && (nsb < splitk || tm != 128)) return cdna4_set_error_msg("gemm_t64: the deep K split is built for 128-row tiles and needs a superblock per work-group");
    // the split in two of a launch that carries the quantizer: its grid is resident by construction (t64_fuses_quantizer), so the hand-off — half tiles exchanged,
    // both work-groups store — is legitimate there and halves the parked bytes (the ticketed default parks whole tiles: WRITE_SIZE 25.2 MB for an 8.4 MB result,
    // profiles/r04/pmc_summary.txt).  CDNA4_FQ_TICKETED=1 keeps the ticketed sum (A/B).
    pl.tm = tm; pl.splitk = splitk; pl.ntiles = ntiles; pl.ticketed2 = ticketed2;
    // (the 128-row form only: with the quantizer's prologue in front of it the 256-row form — 32 registers over budget as it is — reloads spilled values INSIDE its loop,
    //  where a scratch access also breaks the counted vmcnt waits; tests/test_build_static.py pins that the 128-row form's loop has none)
    pl.fq = tm == 128 && t64_fuses_quantizer(a, ntiles * splitk);
    static const bool fq_ticketed = getenv("CDNA4_FQ_TICKETED") && atoi(getenv("CDNA4_FQ_TICKETED")) != 0;
    if (pl.fq && splitk == 2 && tm == 128 && !fq_ticketed) pl.ticketed2 = false;
    return 0;
}

// grouped MUL_MAT_ID: one launch over (m tile) x (activation tile of the expert-sorted image); tiles past the last expert's run exit
int cdna4_launch_gemm_t64_ids(const cdna4_gemm_args &a, const int32_t *tile_expert, const int32_t *row_dst, int64_t w_expert_bytes, hipStream_t st) {
    // (Q4_0R: the expert stack's resident re-layout of Q4_0 — capi.hip finds it by the stack's pointer; the same loop with {d_j, 0} constants)
    if ((a.type != CDNA4_Q4_K && a.type != CDNA4_Q4_0R) || a.K % 256 || a.K < 256 || a.B % 128) return cdna4_set_error_msg("gemm_t64_ids: Q4_K (or Q4_0R), whole superblocks, image rows a multiple of 128");
    const bool z40 = a.type == CDNA4_Q4_0R;
    if ((((uintptr_t)a.W | (uintptr_t)a.w_row_bytes | (uintptr_t)w_expert_bytes) & 15) || ((uintptr_t)a.xh & 15)) return cdna4_set_error_msg("gemm_t64_ids: 16-byte alignment of the expert matrices and the image");
    gemm_params p{};
    p.W = a.W; p.w_row_bytes = a.w_row_bytes; p.xh = (const half_t *)a.xh; p.xh_row = a.xh_row_elems;
    p.Y = a.Y; p.y_row = a.y_row_elems; p.M = a.M; p.K = a.K; p.B = a.B; p.splitk = 1;
    // 256-row tiles (no K ways, half the activation bytes per MFMA: 9-12 % at one tile per CU, above) once the grouped grid offers at least 3/4 of a tile per CU;
    // CDNA4_MOE_TM = 128 / 256 forces the form.  8 experts x 2 used x 512 tokens x 4096^2, one box, two alternations: 83.9-84.1 us on 128-row tiles, 74.1-74.2 on 256-row
    // tiles (profiles/r04/moe_tm_ab.txt)
    static const int tm_env = getenv("CDNA4_MOE_TM") ? atoi(getenv("CDNA4_MOE_TM")) : 0;
    const int cus0 = cdna4_gemm_cu_count();
    // Round 5: the plan also writes a tile ORDER (fullest image tiles first) and per-tile fragment counts (k_moe_plan): the 128-row form starts its tiles in that order and
    // issues no MFMAs for the empty 32-row fragments of an expert's last tile.  Measured at 8 x 2 x 512 x 4096^2, one box, us per call (profiles/r05/moe_ab.txt): 128-row tiles
    // 93.5-94.2 without the tables, 92.3-92.7 with them — and 84.0 on 256-row tiles, which stay the choice there: a full-K 128 x 128 tile takes ~40 us (not the 27 us of the
    // headline's half-K tiles), so the eight full tiles x 32 m-tiles are one 40-us round and the light tiles a second of ~30; the 256-row form is ONE round.  The tables
    // therefore serve the grids that take 128-row tiles by the rule below (CDNA4_MOE_PLAN=0: without them).
    static const bool plan_off = getenv("CDNA4_MOE_PLAN") && atoi(getenv("CDNA4_MOE_PLAN")) == 0;
    const int tm = tm_env == 128 || tm_env == 256 ? tm_env : ((((a.M + 255) / 256) * (a.B / 128)) * 4 >= cus0 * 3 ? 256 : 128);
    p.tiles_m = (a.M + tm - 1) / tm; p.tiles_b = a.B / 128;
    p.tile_expert = tile_expert; p.row_dst = row_dst; p.w_expert_bytes = w_expert_bytes;
    if (!plan_off && tm == 128) { p.tile_order = tile_expert + p.tiles_b; p.tile_nfrag = tile_expert + 2 * p.tiles_b; }      // (k_moe_plan wrote them behind tile_expert: capi.hip moe_carve)
    if (tm == 256) {
        if (z40) hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_0R, 256, true>), dim3(p.tiles_m * p.tiles_b), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 256, true>), dim3(p.tiles_m * p.tiles_b), dim3(512), 0, st, p);
        CDNA4_CHECK_LAUNCH();
        return 0;
    }
    // K split in two with the TICKETED sum (the last of a tile's two work-groups to arrive adds both partial tiles, in the order ks = 0, 1, and stores):
    // OPT-IN (CDNA4_MOE_SPLITK=2) — a measured loss: the grouped grid is 384 30-us tiles on 256 CUs (two rounds) and 768 halves would be three rounds of half
    // the length, but every half tile also parks 64 KB write-through and the last arrival reads its partner's: 8 x 2 x 512 x 4096^2 on one MI355X box,
    // 82.0 us unsplit vs 96.4 us split (gpurun_out/s11, profiles/r04/moe_ab.txt).
    static const int sk_env = getenv("CDNA4_MOE_SPLITK") ? atoi(getenv("CDNA4_MOE_SPLITK")) : 1;
    const int ntiles = p.tiles_m * p.tiles_b, cus = cdna4_gemm_cu_count();
    if (sk_env == 2 && a.K / 256 >= 4 && ntiles * 2 > cus && (size_t)ntiles * 8 <= 32768) {
        p.tile_order = nullptr; p.tile_nfrag = nullptr;                  // (the split's (tile, ks) mapping is the table-free one)
        const size_t pbytes = (size_t)ntiles * 2 * 8 * 16384, fbytes = 65536;
        char *sc = (char *)cdna4_gemm_scratch(fbytes + pbytes, 2);
        if (sc) { p.splitk = 2; p.tune = 2; p.flags = (unsigned *)sc; p.partial = (float *)(sc + fbytes); }
    }
    if (z40) hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_0R, 128, true>), dim3(p.tiles_m * p.tiles_b * p.splitk), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_kq_t64<CDNA4_Q4_K, 128, true>), dim3(p.tiles_m * p.tiles_b * p.splitk), dim3(512), 0, st, p);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
