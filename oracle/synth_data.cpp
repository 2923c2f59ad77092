// oracle/synth_data.cpp — MEASUREMENT INFRASTRUCTURE ONLY.  The synthetic inputs SURVEY.md §8(d) prescribes for the benchmark:
// W fp32 uniform(-1, 1) from std::mt19937(1234) — the distribution of tests/test-backend-ops.cpp:37,54 — through the
// reference's own ggml_quantize_chunk (src/ggml.c:6410), X fp32 uniform(-1, 1) from std::mt19937(4321).  Writes the row range
// [row_lo, row_hi) of the quantized [M x K] matrix (the generator is run over the whole matrix so that every row shard of a
// multi-GPU run is a slice of the SAME matrix) and the [B x K] activations.
//   synth_data <type> <M> <K> <row_lo> <row_hi> <B> <prefix>      -> <prefix>.w.bin, <prefix>.x.bin
#include "ggml.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

int main(int argc, char ** argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s type M K row_lo row_hi B prefix\n", argv[0]); return 2; }
    ggml_type type = GGML_TYPE_COUNT;
    for (int t = 0; t < GGML_TYPE_COUNT; t++) if (strcmp(ggml_type_name((ggml_type)t), argv[1]) == 0) type = (ggml_type)t;
    if (type == GGML_TYPE_COUNT) { fprintf(stderr, "unknown type\n"); return 2; }
    const int64_t M = atoll(argv[2]), K = atoll(argv[3]), lo = atoll(argv[4]), hi = atoll(argv[5]), B = atoll(argv[6]);
    const std::string prefix = argv[7];
    if (lo < 0 || hi > M || lo >= hi) { fprintf(stderr, "bad row range\n"); return 2; }
    struct ggml_init_params ip = { 1 << 20, nullptr, false };
    ggml_free(ggml_init(ip));                                        // fp16 tables
    std::mt19937 g(1234); std::uniform_real_distribution<float> u(-1.f, 1.f);
    for (int64_t i = 0; i < lo * K; i++) (void)u(g);                 // skip the rows in front of the shard: same stream as the whole matrix
    std::vector<float> wf((size_t)(hi - lo) * K);
    for (auto & v : wf) v = u(g);
    const size_t rb = ggml_row_size(type, K);
    std::vector<uint8_t> wq((size_t)(hi - lo) * rb);
    ggml_quantize_init(type);
    const int nt = (int)std::min<int64_t>(32, std::max<int64_t>(1, (hi - lo) / 64));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
        const int64_t r0 = (hi - lo) * t / nt, r1 = (hi - lo) * (t + 1) / nt;
        if (r1 > r0) ggml_quantize_chunk(type, wf.data() + r0 * K, wq.data() + r0 * rb, 0, r1 - r0, K, nullptr);
    });
    for (auto & t : th) t.join();
    FILE * f = fopen((prefix + ".w.bin").c_str(), "wb"); fwrite(wq.data(), 1, wq.size(), f); fclose(f);
    std::mt19937 gx(4321);
    std::vector<float> x((size_t)B * K);
    for (auto & v : x) v = u(gx);
    f = fopen((prefix + ".x.bin").c_str(), "wb"); fwrite(x.data(), 4, x.size(), f); fclose(f);
    printf("{\"type\":\"%s\",\"M\":%lld,\"K\":%lld,\"rows\":[%lld,%lld],\"B\":%lld}\n", ggml_type_name(type), (long long)M, (long long)K, (long long)lo, (long long)hi, (long long)B);
    return 0;
}
