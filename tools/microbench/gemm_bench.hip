// gemm_bench.hip — C++ timing harness for the Q4_K MFMA GEMM through the C-ABI (no Python in the loop): HIP events
// around N back-to-back ggml_cdna4_mul_mat_prepared calls, for a list of (variant, splitk, ablation mask).
//   gemm_bench [M K B]        links ../../ggml_amd/lib/libcdna4_kernels.so
#include "../../include/ggml_cdna4.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <random>

// block 0's milestones (gemm_kq_t64.inc: mile()), us since the kernel's entry stamp
static void t64_milestones(const std::vector<unsigned long long> &tr) {
    const unsigned long long e0 = tr[1];
    auto us = [&](int i) { return tr[i] > e0 ? (double)(tr[i] - e0) / 100.0 : -1.0; };
    printf("  block 0 milestones, us since entry (s_memrealtime):");
    if (tr[4] > e0) printf("  own share quantized %.2f  grid barrier passed %.2f", us(4), us(5));
    printf("  stage 0 in LDS %.2f  loop done %.2f  K ways met in LDS %.2f  halves parked %.2f  partner's flag seen %.2f  tiles exchanged %.2f  stores issued %.2f  stores drained %.2f\n", us(6), us(3), us(266), us(267), us(268), us(7), us(265), us(264));
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);          // a GPU fault aborts the process: keep what was printed so far
    const int64_t M = argc > 1 ? atoll(argv[1]) : 4096, K = argc > 2 ? atoll(argv[2]) : 4096, B = argc > 3 ? atoll(argv[3]) : 512;
    const size_t wbytes = ggml_cdna4_row_size(GGML_CDNA4_TYPE_Q4_K, K) * M;
    std::vector<uint8_t> w(wbytes); std::mt19937 g(1);
    for (auto &b : w) b = (uint8_t)g();
    for (size_t i = 0; i < wbytes; i += 144) { w[i] = 0x00; w[i + 1] = 0x1c; w[i + 2] = 0x00; w[i + 3] = 0x20; }   // fp16 d ~ 0.0039, dmin ~ 0.0078
    std::vector<float> x((size_t)B * K); std::uniform_real_distribution<float> u(-1, 1); for (auto &v : x) v = u(g);
    void *dw, *dx, *dy, *ws; const size_t wsz = ggml_cdna4_mul_mat_workspace_size(GGML_CDNA4_TYPE_Q4_K, K, B);
    hipMalloc(&dw, wbytes); hipMalloc(&dx, x.size() * 4); hipMalloc(&dy, (size_t)B * M * 4); hipMalloc(&ws, wsz);
    hipMemcpy(dw, w.data(), wbytes, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    if (ggml_cdna4_prepare_act(GGML_CDNA4_TYPE_Q4_K, (const float *)dx, K, K, B, ws, wsz, GGML_CDNA4_PATH_GEMM, 0)) { printf("prepare failed: %s\n", ggml_cdna4_last_error()); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct cfg { int variant, splitk, ablate; };
    std::vector<cfg> cfgs;
    // GB_VARIANTS="23,55,151" overrides the variant list (see launch_type() in gemm_q_mfma.hip for the bits)
    std::vector<int> vars = {0, 5, 663};
    if (const char *e = getenv("GB_VARIANTS")) { vars.clear(); for (const char *q = e; *q;) { vars.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q) q++; } }
    std::vector<int> sks = {1, 2};
    if (const char *e = getenv("GB_SPLITKS")) { sks.clear(); for (const char *q = e; *q;) { sks.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q) q++; } }
    for (int v : vars) for (int sk : sks) cfgs.push_back({v, sk, 0});
    std::vector<float> yref, ycur((size_t)B * M);
    printf("M=%lld K=%lld B=%lld  flops=%.3f G\n", (long long)M, (long long)K, (long long)B, 2.0 * M * K * B / 1e9);
    // Measurement protocol: the GPU clocks and caches ramp for the first milliseconds, which used to bias whatever config
    // came first.  So: a long untimed warm-up, then GB_ROUNDS passes over ALL configs (round-robin), 100 launches each, and the
    // MINIMUM per config is reported (with the mean of the rounds beside it).
    auto run_cfg = [&](const cfg &c) { return ggml_cdna4_mul_mat_prepared(GGML_CDNA4_TYPE_Q4_K, dw, ggml_cdna4_row_size(GGML_CDNA4_TYPE_Q4_K, K), (float *)dy, M, M, K, B, ws, wsz, GGML_CDNA4_PATH_GEMM, c.variant, c.splitk, 0); };
    for (auto &c : cfgs) for (int i = 0; i < 3; i++) if (run_cfg(c)) { printf("launch failed (variant %d splitk %d): %s\n", c.variant, c.splitk, ggml_cdna4_last_error()); return 1; }
    { const double per = 2.0 * M * K * B / 500e12; const int nw = (int)(0.05 / per) + 50; for (int i = 0; i < nw; i++) run_cfg(cfgs[i % cfgs.size()]); hipDeviceSynchronize(); }
    const int rounds = getenv("GB_ROUNDS") ? atoi(getenv("GB_ROUNDS")) : 4, n = 100;
    std::vector<double> best(cfgs.size(), 1e30), sum(cfgs.size(), 0.0);
    for (int r = 0; r < rounds; r++)
        for (size_t ci = 0; ci < cfgs.size(); ci++) {
            hipEventRecord(e0, 0); for (int i = 0; i < n; i++) run_cfg(cfgs[ci]); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / n; if (us < best[ci]) best[ci] = us; sum[ci] += us;
        }
    for (size_t ci = 0; ci < cfgs.size(); ci++) {
        const cfg &c = cfgs[ci];
        run_cfg(c); hipDeviceSynchronize();
        hipMemcpy(ycur.data(), dy, ycur.size() * 4, hipMemcpyDeviceToHost);
        if (yref.empty()) yref = ycur;
        double num = 0, den = 0; for (size_t i = 0; i < ycur.size(); i++) { const double d = (double)ycur[i] - yref[i]; num += d * d; den += (double)yref[i] * yref[i]; }
        printf("variant %4d splitk %d : %8.2f us/call (min of %d rounds; mean %6.2f) %8.1f TFLOP/s   rel-L2 vs first config %.2e\n", c.variant, c.splitk, best[ci], rounds, sum[ci] / rounds, 2.0 * M * K * B / (best[ci] * 1e-6) / 1e12, sqrt(num / (den + 1e-30)));
    }
    // per-phase timeline of the 8-wave kernel's first work-group (stages 4..19), from s_memtime stamps; argv[4] = "23,55,.."
    unsigned long long *dtr; hipMalloc(&dtr, 65536);
    std::vector<int> tv; for (const char *q = argc > 4 ? argv[4] : ""; *q;) { tv.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q) q++; }
    for (int v : tv) {
        hipMemset(dtr, 0, 65536);
        ggml_cdna4_debug_trace(dtr);
        // GB_TRACE_REPS > 1: the stamps come from the LAST of that many back-to-back launches, i.e. from the clock the chip
        // settles at under sustained load rather than from one cold launch
        for (int r = 1; r < (getenv("GB_TRACE_REPS") ? atoi(getenv("GB_TRACE_REPS")) : 1); r++)
            ggml_cdna4_mul_mat_prepared(GGML_CDNA4_TYPE_Q4_K, dw, ggml_cdna4_row_size(GGML_CDNA4_TYPE_Q4_K, K), (float *)dy, M, M, K, B, ws, wsz, GGML_CDNA4_PATH_GEMM, v, getenv("GB_TRACE_SPLITK") ? atoi(getenv("GB_TRACE_SPLITK")) : 1, 0);
        ggml_cdna4_mul_mat_prepared(GGML_CDNA4_TYPE_Q4_K, dw, ggml_cdna4_row_size(GGML_CDNA4_TYPE_Q4_K, K), (float *)dy, M, M, K, B, ws, wsz, GGML_CDNA4_PATH_GEMM, v, getenv("GB_TRACE_SPLITK") ? atoi(getenv("GB_TRACE_SPLITK")) : 1, 0);
        hipDeviceSynchronize();
        ggml_cdna4_debug_trace(nullptr);
        std::vector<unsigned long long> tr(8 * 16 * 8 + 1104); hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost);
        if (tr[8 * 16 * 8 + 1103] > tr[8 * 16 * 8 + 1101]) {
            const double cyc = (double)(tr[8 * 16 * 8 + 1102] - tr[8 * 16 * 8 + 1100]), ref = (double)(tr[8 * 16 * 8 + 1103] - tr[8 * 16 * 8 + 1101]);
            printf("  variant %d block 0: %.0f s_memtime ticks in %.2f us of s_memrealtime (100 MHz) -> s_memtime runs at %.0f MHz\n", v, cyc, ref / 100.0, cyc / (ref / 100.0));
        }
        if (v & (1 << 28)) {                         // k_gemm_lds (ablation build, ABL = 256): clock + [wave][period 8..15][phase][L begin, L end, after barrier A, M end] of block 0
            const double cyc = (double)(tr[2] - tr[0]), ref = (double)(tr[3] - tr[1]);
            if (ref > 0) printf("  lds variant %d block 0: %.0f s_memtime ticks in %.2f us (s_memrealtime, 100 MHz) -> %.0f MHz\n", v, cyc, ref / 100.0, cyc / (ref / 100.0));
            printf("  per phase (cycles, mean over periods 8..15): L part | wait at barrier A | M part (8 MFMAs) | wait at barrier B ;  phase length\n");
            const int nkk = (v & (1 << 29)) ? 2 : 4;                 // phases per K tile: the 128-row form has two
            for (int w = 0; w < 8; w++) for (int ph = 0; ph < nkk; ph++) {
                double a[4] = {0, 0, 0, 0}; int n = 0;
                for (int tt = 0; tt < 8; tt++) {
                    const unsigned long long *q = tr.data() + 16 + ((w * 8 + tt) * 4 + ph) * 4;
                    const unsigned long long *qn = ph + 1 < nkk ? q + 4 : (tt < 7 ? tr.data() + 16 + ((w * 8 + tt + 1) * 4) * 4 : nullptr);
                    if (!q[0] || !q[3] || !qn || !qn[0]) continue;
                    a[0] += (double)(q[1] - q[0]); a[1] += (double)(q[2] - q[1]); a[2] += (double)(q[3] - q[2]); a[3] += (double)(qn[0] - q[3]); n++;
                }
                if (n) printf("    wave %d phase %d (%d samples): %6.0f | %6.0f | %6.0f | %6.0f ;  %6.0f\n", w, ph, n, a[0] / n, a[1] / n, a[2] / n, a[3] / n, (a[0] + a[1] + a[2] + a[3]) / n);
            }
            continue;
        }
        if (v & 8192) {                              // k_gemm_kq_t64 (ablation build): clock + [wave][stage 16..31][phase] stamps of block 0
            const double cyc = (double)(tr[2] - tr[0]), ref = (double)(tr[3] - tr[1]);
            if (ref > 0) printf("  t64 variant %d block 0: %.0f s_memtime ticks in %.2f us (s_memrealtime, 100 MHz) -> %.0f MHz\n", v, cyc, ref / 100.0, cyc / (ref / 100.0));
            const uint32_t *st32 = (const uint32_t *)(tr.data() + 8);
            printf("  per stage (cycles): issue of the first k-steps | lgkm+vmcnt wait | barrier wait | last k-step = rest;  stage length\n");
            for (int w = 0; w < 8; w++) {
                double a[4] = {0, 0, 0, 0}, len = 0; int n = 0;
                for (int sgi = 0; sgi + 1 < 16; sgi++) {
                    const uint32_t *q = st32 + (w * 16 + sgi) * 4, *qn = q + 4;
                    if (!q[0] || !qn[0]) continue;
                    a[0] += (uint32_t)(q[1] - q[0]); a[1] += (uint32_t)(q[2] - q[1]); a[2] += (uint32_t)(q[3] - q[2]); a[3] += (uint32_t)(qn[0] - q[3]); len += (uint32_t)(qn[0] - q[0]); n++;
                }
                if (n) printf("    wave %d (%2d stages): %7.0f | %6.0f | %6.0f | %6.0f ;  %7.0f\n", w, n, a[0] / n, a[1] / n, a[2] / n, a[3] / n, len / n);
            }
            for (int w : {0, 4}) { printf("    wave %d stage starts:", w); for (int sgi = 0; sgi < 16; sgi++) printf(" %u", st32[(w * 16 + sgi) * 4] - st32[0]); printf("\n"); }
            t64_milestones(tr);
            continue;
        }
        if (v >= 65536) continue;                    // k_gemm_kq_w12 experiments record the clock only
        {   // which XCD did each work-group land on?  (the tile remap assumes blockIdx % 8)
            int nb = 0, mism = 0; for (int b = 0; b < 1024; b++) { const unsigned long long x = tr[8 * 16 * 8 + 32 + b]; if (b < 256 || x) { nb++; if ((int)x != b % 8) mism++; } }
            printf("  XCC_ID check: %d work-groups recorded, %d with XCC_ID != blockIdx %% 8; first 16:", nb, mism);
            for (int b = 0; b < 16; b++) printf(" %llu", tr[8 * 16 * 8 + 32 + b]); printf("\n");
        }
        {   // kernel-level milestones, relative to the consumer work-group's entry stamp
            const unsigned long long e0 = tr[8 * 16 * 8];
            static const char *nm[8] = {"entry", "loop done", "K halves summed", "tile in LDS", "flag seen", "stores issued", "drained", "flag set"};
            for (int ks = 0; ks < 2; ks++) { printf("  %s:", ks ? "producer (ks=1)" : "consumer (ks=0)"); for (int i = 0; i < 8; i++) if (tr[8 * 16 * 8 + 16 * ks + i]) printf("  %s %lld", nm[i], (long long)(tr[8 * 16 * 8 + 16 * ks + i] - e0)); printf("\n"); }
        }
        printf("trace variant %d (cycles since the stage-4 stamp of wave 0; w8: 0 stage start, 1 after vmcnt wait, 2 after barrier, 4 frag 0 ready, 6 stage end; w8p: 0 T_a begin, 1 T_a done, 2 waits done, 3 barrier passed, 4 next reads issued, 6 T_b done)\n", v);
        const unsigned long long t0 = tr[0];
        for (int w : {0, 4}) for (int st = 0; st < 5; st++) {
            printf("wave %d stage %2d:", w, st + 4);
            for (int ph : {0, 1, 2, 3, 4, 5, 6}) printf(" %7lld", (long long)(tr[(w * 16 + st) * 8 + ph] - t0));
            printf("\n");
        }
    }
    // GB_TRACE_FQ=1 (the -DCDNA4_ABLATIONS library, owned-device mode): the one-launch step (k_gemm_kq_t64<.., FQ>: quantizer share, grid barrier, multiply) with block 0's
    // milestones, and the same launch timed by HIP events WITHOUT the instrumentation's drain (trace buffer off) — the difference between the event time and the last
    // milestone is what the launch itself takes (dispatch to first instruction, last store to completion signal)
    if (getenv("GB_TRACE_FQ")) {
        auto step = [&]() { return ggml_cdna4_mul_mat(GGML_CDNA4_TYPE_Q4_K, dw, ggml_cdna4_row_size(GGML_CDNA4_TYPE_Q4_K, K), (const float *)dx, K, (float *)dy, M, M, K, B, ws, wsz, GGML_CDNA4_PATH_GEMM, 0, 0, 0); };
        for (int i = 0; i < 20; i++) if (step()) { printf("one-launch step failed: %s\n", ggml_cdna4_last_error()); return 1; }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int r = 0; r < 4; r++) { hipEventRecord(e0, 0); for (int i = 0; i < 100; i++) step(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("one-launch step (route %d): %.2f us per launch by HIP events (100 back to back, min of 4)\n", ggml_cdna4_mul_mat_route(GGML_CDNA4_TYPE_Q4_K, M, K, B), best * 10.0f);
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(dtr, 0, 65536);
            for (int i = 0; i < 30; i++) step();                          // (the clock the chip settles at under load)
            ggml_cdna4_debug_trace(dtr);
            step();
            hipDeviceSynchronize();
            ggml_cdna4_debug_trace(nullptr);
            std::vector<unsigned long long> tr(8 * 16 * 8 + 1104); hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost);
            t64_milestones(tr);
        }
    }
    return 0;
}
