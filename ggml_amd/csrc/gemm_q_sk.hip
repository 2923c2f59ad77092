// gemm_q_sk.hip — launcher of k_gemm_kq_sk (gemm_kq_sk.inc): the grouped MUL_MAT_ID prefill as one stream-k launch of persistent work-groups.
// Replaces, for Q4_K experts (and Q4_0 experts with a resident Q4_0R image) at prefill sizes, what ggml_compute_forward_mul_mat_id does after its row lists are built
// (/root/reference/src/ggml-cpu/ggml-cpu.c:7699-7781: one mul_mat per expert over that expert's rows).
#include "gemm_q_common.h"
#include <stdlib.h>
#include "gemm_q_hw.h"
#include "gemm_kq_sk.inc"

// the launch geometry both halves of the step agree on: G spans (= work-groups of the GEMM launch), units per activation tile, tile-record capacity
int cdna4_gemm_sk_spans() {
    static const int env = getenv("CDNA4_SK_SPANS") ? atoi(getenv("CDNA4_SK_SPANS")) : 0;      // (measurement knob)
    const int cus = cdna4_gemm_cu_count();
    return env > 0 ? (env < 1024 ? env : 1024) : (cus < 1024 ? cus : 1024);
}
bool cdna4_gemm_sk_supported(int type, int64_t M, int64_t K, int64_t n_rows, int64_t ntile_cap) {
    if ((type != CDNA4_Q4_K && type != CDNA4_Q4_0R) || K % 256 || K < 256 || M <= 0) return false;
    if (ntile_cap > CDNA4_SK_MAX_TILES || n_rows * 256 >= ((int64_t)1 << 32)) return false;                 // the planner's LDS tables; 32-bit gather offsets within a panel
    const int64_t upt = ((M + 127) / 128) * (K / 256);
    return ntile_cap * upt < ((int64_t)1 << 31);                        // unit indices are int32
}
// a.W / a.w_row_bytes: the expert stack (Q4_K blocks, or the resident Q4_0R image), a.xh: the token-order image of a.B rows, a.Y rows indexed by (token, slot) pair.
// (Round 6 also built the same walk on k_gemm_r8's 256 x 256 tiles — k_gemm_r8_sk, commit "stream-k MUL_MAT_ID: lean planner ..." — parity-green and BEHIND this form: its
//  loop costs 7.2 us per superblock whether four or five of a tile's eight fragments hold rows, exactly four 128 x 128 tiles of this kernel, and an expert's run is padded to
//  256 rows instead of 128: 8 x 2 x 512 x 4096^2 86.9-87.3 us against 86.5 on the same box, profiles/r06/moe_sk_r8_vs_t64_ab.txt.  Removed.)
int cdna4_launch_gemm_sk(const cdna4_gemm_args &a, const int32_t *tile_rec, const int32_t *wg_begin, int G, int64_t w_expert_bytes, hipStream_t st) {
    if ((a.type != CDNA4_Q4_K && a.type != CDNA4_Q4_0R) || a.K % 256 || a.K < 256) return cdna4_set_error_msg("gemm_sk: Q4_K (or Q4_0R), whole superblocks");
    if ((((uintptr_t)a.W | (uintptr_t)a.w_row_bytes | (uintptr_t)w_expert_bytes) & 15) || ((uintptr_t)a.xh & 15)) return cdna4_set_error_msg("gemm_sk: 16-byte alignment of the expert matrices and the image");
    if (G <= 0 || G > 1024) return cdna4_set_error_msg("gemm_sk: 1 .. 1024 spans");
    // parked partial tiles: [span][2][8 waves][8 KB] behind 4 KB of ticket words (zero when idle: reset by their last user — no per-launch state on the host, graph-capturable)
    const size_t tbytes = 4096, pbytes = (size_t)G * 2 * 8 * 8192;
    char *sc = (char *)cdna4_gemm_scratch(tbytes + pbytes, 10);
    if (!sc) return cdna4_set_error_msg("gemm_sk: cannot allocate the exchange scratch");
    sk_params p{};
    p.W = a.W; p.w_row_bytes = a.w_row_bytes; p.w_expert_bytes = w_expert_bytes; p.xh = (const half_t *)a.xh;
    p.Y = a.Y; p.y_row = a.y_row_elems; p.M = a.M; p.K = a.K; p.B = a.B; p.tiles_m = (a.M + 127) / 128;
    p.tile_rec = tile_rec; p.wg_begin = wg_begin; p.tickets = (unsigned *)sc; p.partial = (float *)(sc + tbytes);
#ifdef CDNA4_ABLATIONS
    // measurement build (tools/microbench: libcdna4_kernels_abl.so): with a trace buffer set (ggml_cdna4_debug_trace: G * 512 bytes) the instrumented twin runs
    if (cdna4_debug_trace && a.type == CDNA4_Q4_K) { p.trace = (unsigned long long *)cdna4_debug_trace; hipLaunchKernelGGL((k_gemm_kq_sk<CDNA4_Q4_K, true>), dim3(G), dim3(512), 0, st, p); CDNA4_CHECK_LAUNCH(); return 0; }
#endif
    if (a.type == CDNA4_Q4_0R) hipLaunchKernelGGL((k_gemm_kq_sk<CDNA4_Q4_0R>), dim3(G), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_kq_sk<CDNA4_Q4_K>), dim3(G), dim3(512), 0, st, p);
    CDNA4_CHECK_LAUNCH();
    return 0;
}
