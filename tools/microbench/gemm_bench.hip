// gemm_bench.hip — C++ timing harness for the Q4_K MFMA GEMM through the C-ABI (no Python in the loop): HIP events
// around N back-to-back ggml_cdna4_mul_mat_prepared calls, for a list of (variant, splitk, ablation mask).
//   gemm_bench [M K B]        links ../../ggml_amd/lib/libcdna4_kernels.so
#include "../../include/ggml_cdna4.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <random>

int main(int argc, char **argv) {
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
    for (int v : {0, 5, 23}) for (int sk : {1, 2}) cfgs.push_back({v, sk, 0});
    printf("M=%lld K=%lld B=%lld  flops=%.3f G\n", (long long)M, (long long)K, (long long)B, 2.0 * M * K * B / 1e9);
    for (auto c : cfgs) {
        auto run = [&] { return ggml_cdna4_mul_mat_prepared(GGML_CDNA4_TYPE_Q4_K, dw, ggml_cdna4_row_size(GGML_CDNA4_TYPE_Q4_K, K), (float *)dy, M, M, K, B, ws, wsz, GGML_CDNA4_PATH_GEMM, c.variant, c.splitk, 0); };
        for (int i = 0; i < 5; i++) if (run()) { printf("launch failed: %s\n", ggml_cdna4_last_error()); return 1; }
        hipDeviceSynchronize();
        const int n = 100;
        hipEventRecord(e0, 0); for (int i = 0; i < n; i++) run(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("variant %2d splitk %d ablate %2d : %8.2f us/call  %8.1f TFLOP/s\n", c.variant, c.splitk, c.ablate, ms * 1e3 / n, 2.0 * M * K * B / (ms * 1e-3 / n) / 1e12);
    }
    // per-phase timeline of the 8-wave kernel's first work-group (stages 4..19), from s_memtime stamps
    unsigned long long *dtr; hipMalloc(&dtr, 65536); hipMemset(dtr, 0, 65536);
    ggml_cdna4_debug_trace(dtr);
    ggml_cdna4_mul_mat_prepared(GGML_CDNA4_TYPE_Q4_K, dw, ggml_cdna4_row_size(GGML_CDNA4_TYPE_Q4_K, K), (float *)dy, M, M, K, B, ws, wsz, GGML_CDNA4_PATH_GEMM, argc > 4 ? atoi(argv[4]) : 23, 1, 0);
    hipDeviceSynchronize();
    ggml_cdna4_debug_trace(nullptr);
    std::vector<unsigned long long> tr(8 * 16 * 8); hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost);
    printf("trace (cycles since the stage-4 stamp of wave 0; phases: 0 stage start, 1 after vmcnt wait, 2 after barrier, 3 after DMA issue, 4 after follower MFMA, 5 after read+unpack, 6 after leader MFMA)\n");
    const unsigned long long t0 = tr[0];
    for (int w : {0, 4}) for (int st = 0; st < 6; st++) {
        printf("wave %d stage %2d:", w, st + 4);
        for (int ph = 0; ph < 7; ph++) printf(" %7lld", (long long)(tr[(w * 16 + st) * 8 + ph] - t0));
        printf("\n");
    }
    return 0;
}
