// oracle/cpu_baseline.cpp — TEST / MEASUREMENT INFRASTRUCTURE ONLY (never part of the product).
// Times the UNMODIFIED reference CPU backend (libggml-cpu.so built by oracle/ref.mk) on one MUL_MAT node and,
// optionally, dumps its inputs/outputs so the GPU path can be compared with the real reference at full size.
// Built into oracle/_ref/cpu_baseline by oracle/ref.mk; uses only the reference's public API
// (include/ggml.h, ggml-backend.h, ggml-cpu.h), the way tests/test-backend-ops.cpp:4474-4482 sets up the CPU backend.
//
//   cpu_baseline <type> <M> <K> <N> <seconds> <threads> [dump_prefix]
// prints one JSON line: {"type":..,"m":..,"k":..,"n":..,"threads":..,"runs":..,"us_per_run":..,"gflops":..}
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static ggml_type parse_type(const char * s) {
    for (int t = 0; t < GGML_TYPE_COUNT; t++) if (strcmp(ggml_type_name((ggml_type)t), s) == 0) return (ggml_type)t;
    fprintf(stderr, "unknown type %s\n", s); exit(2);
}
static void dump(const std::string & path, const void * p, size_t n) { FILE * f = fopen(path.c_str(), "wb"); fwrite(p, 1, n, f); fclose(f); }

int main(int argc, char ** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s type M K N seconds threads [dump_prefix]\n", argv[0]); return 2; }
    const ggml_type type = parse_type(argv[1]);
    const int64_t M = atoll(argv[2]), K = atoll(argv[3]), N = atoll(argv[4]);
    const double seconds = atof(argv[5]); const int threads = atoi(argv[6]);
    const char * prefix = argc > 7 ? argv[7] : nullptr;

    const size_t wbytes = ggml_row_size(type, K) * M;
    ggml_init_params ip = { wbytes + (size_t)(K * N + M * N) * 4 + (16u << 20), nullptr, false };
    ggml_context * ctx = ggml_init(ip);
    ggml_tensor * a = ggml_new_tensor_2d(ctx, type, K, M);
    ggml_tensor * b = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, K, N);
    ggml_tensor * out = ggml_mul_mat(ctx, a, b);
    // fixed seeds: W uniform(-1,1) mt19937(1234) through ggml_quantize_chunk, X mt19937(4321)
    { std::mt19937 g(1234); std::uniform_real_distribution<float> u(-1.f, 1.f);
      std::vector<float> wf((size_t)M * K); for (auto & v : wf) v = u(g);
      if (ggml_is_quantized(type)) { ggml_quantize_init(type); ggml_quantize_chunk(type, wf.data(), a->data, 0, M, K, nullptr); }
      else if (type == GGML_TYPE_F32) memcpy(a->data, wf.data(), wf.size() * 4);
      else ggml_fp32_to_fp16_row(wf.data(), (ggml_fp16_t *)a->data, (int64_t)wf.size()); }
    { std::mt19937 g(4321); std::uniform_real_distribution<float> u(-1.f, 1.f);
      float * x = (float *)b->data; for (int64_t i = 0; i < K * N; i++) x[i] = u(g); }

    ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, out);
    ggml_backend_t cpu = ggml_backend_cpu_init();
    ggml_backend_cpu_set_n_threads(cpu, threads);
    for (int i = 0; i < 2; i++) ggml_backend_graph_compute(cpu, gf);      // warm-up
    int runs = 0; double el = 0;
    auto t0 = std::chrono::steady_clock::now();
    do { ggml_backend_graph_compute(cpu, gf); runs++; el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } while (el < seconds);
    const double us = el / runs * 1e6, gflops = 2.0 * M * N * K / (us * 1e3);
    printf("{\"type\":\"%s\",\"m\":%lld,\"k\":%lld,\"n\":%lld,\"threads\":%d,\"runs\":%d,\"us_per_run\":%.3f,\"gflops\":%.3f}\n",
           ggml_type_name(type), (long long)M, (long long)K, (long long)N, threads, runs, us, gflops);
    if (prefix) {
        dump(std::string(prefix) + ".w.bin", a->data, ggml_nbytes(a));
        dump(std::string(prefix) + ".x.bin", b->data, ggml_nbytes(b));
        dump(std::string(prefix) + ".y.bin", out->data, ggml_nbytes(out));
    }
    ggml_backend_free(cpu);
    ggml_free(ctx);
    return 0;
}
