// gemm_q_lds.hip — launcher of k_gemm_r8 (gemm_r8.inc): the prefill GEMM with 32(m) x 256(b) wave tiles and in-register unpack, for grids of at least one 256 x 256
// tile per CU.  Replaces, at B > 8, what ggml_compute_forward_mul_mat does after the activations are quantized (/root/reference/src/ggml-cpu/ggml-cpu.c:7510-7605).
// (Round 4 built two more kernels on this launcher — k_gemm_lds, the dequantize-into-LDS structure with two waves per SIMD, and k_gemm_w4, the same with one wave per SIMD;
//  both measured behind k_gemm_r8 / k_gemm_kq_t64 at every shape (profiles/r04/gemm_bench.txt) and were removed in round 5: git show 5eb5f7c:ggml_amd/csrc/gemm_lds.inc, gemm_w4.inc.)
#include "gemm_q_common.h"
#include "gemm_q_hw.h"
#include "gemm_r8.inc"

// (form 2, k_gemm_r8, also takes Q5_K)
static bool lds_supported_t(const cdna4_gemm_args &a, bool five_ok) {
    if (a.type != CDNA4_Q4_K && !(five_ok && (a.type == CDNA4_Q5_K || a.type == CDNA4_Q4_0R || a.type == CDNA4_Q8_0R || a.type == CDNA4_Q6_K8))) return false;
    if (a.M <= 0 || a.B <= 0 || a.K % 256 || a.K < 256) return false;
    if ((((uintptr_t)a.W | (uintptr_t)a.w_row_bytes) & 15) || ((uintptr_t)a.xh & 15)) return false;
    return true;
}
bool cdna4_gemm_lds_supported(const cdna4_gemm_args &a) { return lds_supported_t(a, false); }

// AUTO routing (gemm_q_mfma.hip: launch_type<Q4_K>, cdna4_gemm_q_fuses_tail): k_gemm_r8 where its 256 x 256 tiles fill the chip UNSPLIT, in whole rounds of
// work-groups (>= 90 % of the last one).  Measured on MI355X, us per call, r8 vs k_gemm_kq_t64's 256-row form, two boxes (profiles/r04/gemm_bench*.txt):
// 32768 x 8192 x 512 216 / 221 vs 238 / 243, 32768 x 4096 x 512 115 / 118 vs 127 / 131 — 9-10 % ahead.  At 128 tiles (16384 x 8192 x 512, 8192 x 8192 x 1024,
// 16384 x 4096 x 512, 4096 x 4096 x 2048) r8 needs split-K = 2 through its reduce-scatter exchange and is level with t64's 256-row tiles on one box (123.4 vs
// 123.7, 123.4 vs 123.5) and 5 % behind on the K = 4096 shapes (71.6 vs 68.0, 72.2 vs 68.6): those grids stay on t64.  Below that (32 tiles: 4096 x 4096 x 512
// 39 vs 25, 4096 x 11008 x 512 59-61 vs 50-52) the 8-way exchange costs far more than the leaner loop saves.
bool cdna4_gemm_r8_preferred(const cdna4_gemm_args &a) {
    static const bool off = getenv("CDNA4_NO_R8") && atoi(getenv("CDNA4_NO_R8")) != 0;
    if (off || !lds_supported_t(a, true)) return false;
    const int cus = cdna4_gemm_cu_count(), ntiles = ((a.M + 255) / 256) * ((a.B + 255) / 256);
    if (ntiles < cus) return false;
    const int rounds = (ntiles + cus - 1) / cus;
    return ntiles * 10 >= rounds * cus * 9;                             // whole rounds of work-groups (>= 90 % of the last one): the tiles are large
}

// tile rows (0 = choose; 128 / 256) and split-K (0 = choose) -> launch.  Returns 0, or a negative status with the error text set.
// form: 2 = k_gemm_r8 (the only one left; 256-row tiles only)
int cdna4_launch_gemm_lds(const cdna4_gemm_args &a, int tm, int splitk, hipStream_t st, int form) {
    if (form != 2) return cdna4_set_error_msg("gemm_lds: k_gemm_lds / k_gemm_w4 were measured behind k_gemm_r8 and removed in round 5 (gemm_variant bit 26 selects k_gemm_r8)");
    if (!lds_supported_t(a, form == 2)) return cdna4_set_error_msg("gemm_r8: Q4_K, Q5_K or the resident re-layouts Q4_0R / Q8_0R / Q6_K8 on 16-byte-aligned rows, whole superblocks");
    const int cus = cdna4_gemm_cu_count(), nsb = a.K / 256;
    const int tiles_b = (a.B + 255) / 256;
    if (form == 2) tm = 256;
    if (tm <= 0) tm = (((a.M + 255) / 256) * tiles_b >= cus) ? 256 : 128;
    if (tm != 128 && tm != 256) return cdna4_set_error_msg("gemm_lds: tile rows are 128 or 256");
    const int tiles_m = (a.M + tm - 1) / tm, ntiles = tiles_m * tiles_b;
    // split-K: S co-resident work-groups per tile reduce-scatter their partial tiles (gemm_lds.inc, epilogue (2)); needs every work-group resident
    // (one per CU) and S to divide the 4 / 8 accumulator fragments of a wave.  Deterministic (fixed summation order).
    const int nfr = 8, nwv = 8;
    if (splitk <= 0) {
        splitk = 1;
        for (int s = 2; s <= 8; s *= 2) if (ntiles * s <= cdna4_gemm_coresident_cus() && nsb >= 2 * s) splitk = s;
    }
    if (splitk > 1 && cdna4_gemm_shared_device()) return cdna4_set_error_msg("gemm_lds: the split-K exchange of these kernels waits for co-resident work-groups; not on a shared device (ggml_cdna4_set_shared_device)");
    if (splitk < 1 || nfr % splitk || (splitk > 1 && ntiles * splitk > cus) || nsb < splitk) return cdna4_set_error_msg("gemm_lds: split-K must divide the wave's fragments, leave a superblock per work-group and keep every work-group resident");
    gemm_params p{};
    p.W = a.W; p.w_row_bytes = a.w_row_bytes; p.xh = (const half_t *)a.xh; p.xh_row = a.xh_row_elems;
    p.Y = a.Y; p.y_row = a.y_row_elems; p.M = a.M; p.K = a.K; p.B = a.B; p.splitk = splitk;
    p.tiles_m = tiles_m; p.tiles_b = tiles_b;
    p.epi = a.epi;
    if (splitk > 1) {
        // counters [tile][arrivals, departures] in a fixed 64-KB area, slots [tile][dst][src][wave][fragment] x 4 KB behind it; scratch kind 9 (this kernel's own)
        const size_t pbytes = (size_t)ntiles * splitk * splitk * nwv * (nfr / splitk) * 4096, fbytes = 65536;
        if ((size_t)ntiles * 2 > 16384) return cdna4_set_error_msg("gemm_lds: too many tiles for the split-K counter area");
        char *sc = (char *)cdna4_gemm_scratch(fbytes + pbytes, 9);
        if (!sc) return cdna4_set_error_msg("gemm_lds: cannot allocate split-K scratch");
        p.flags = (unsigned *)sc; p.partial = (float *)(sc + fbytes);
        p.fault = cdna4_gemm_fault_word();
    }
    const dim3 grid(ntiles * splitk);
#ifdef CDNA4_ABLATIONS
    p.trace = (unsigned long long *)cdna4_debug_trace;
    const int abl = (a.variant >> 16) & 0x1FF;
#define R8_ABL(A) if (form == 2 && a.type == CDNA4_Q4_K && abl == (A)) { hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q4_K, (A)>), grid, dim3(512), 0, st, p); CDNA4_CHECK_LAUNCH(); return 0; }
    R8_ABL(1) R8_ABL(2) R8_ABL(3) R8_ABL(4) R8_ABL(8) R8_ABL(16) R8_ABL(32) R8_ABL(15)
    if (form == 2 && abl) return cdna4_set_error_msg("gemm_r8: ablation not instantiated");
#endif
    if (form == 2) {
        const bool tail = a.epi.bias || a.epi.act || a.epi.resid;
        if (a.type == CDNA4_Q4_0R) { if (tail) hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q4_0R, 0, true>), grid, dim3(512), 0, st, p); else hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q4_0R>), grid, dim3(512), 0, st, p); }
        else if (a.type == CDNA4_Q8_0R) { if (tail) hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q8_0R, 0, true>), grid, dim3(512), 0, st, p); else hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q8_0R>), grid, dim3(512), 0, st, p); }
        else if (a.type == CDNA4_Q6_K8) { if (tail) hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q6_K8, 0, true>), grid, dim3(512), 0, st, p); else hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q6_K8>), grid, dim3(512), 0, st, p); }
        else if (a.type == CDNA4_Q5_K) { if (tail) hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q5_K, 0, true>), grid, dim3(512), 0, st, p); else hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q5_K>), grid, dim3(512), 0, st, p); }
        else if (tail) hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q4_K, 0, true>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((k_gemm_r8<CDNA4_Q4_K>), grid, dim3(512), 0, st, p);
        CDNA4_CHECK_LAUNCH();
        return 0;
    }
    return cdna4_set_error_msg("gemm_lds: unknown form");
}
