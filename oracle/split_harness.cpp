// oracle/split_harness.cpp — TEST INFRASTRUCTURE ONLY.  Drives the plug-in's scheduler-facing entry points through ggml's PUBLIC
// API (include/ggml-backend.h), the way an application such as llama.cpp would, and checks every result against the reference
// CPU backend on identical data:
//   split    weights in ggml_backend_split_buffer_type (get_proc_address, include/ggml-backend.h:188), MUL_MAT on the main device
//            (rows sharded per GGML_CDNA4_SPLIT_SELF on a one-GPU box), whole-tensor set / get round trip
//   async    ggml_backend_tensor_set_async / get_async / copy_async between two backends of the device + events
//   host     the pinned host buffer type (ggml_backend_dev_host_buffer_type): allocation, is_host, CPU-visible
//   replay   a transformer MLP block (NORM, MUL, ADD, MUL_MAT, ADD, GELU, MUL_MAT, ADD, ADD) computed again and again with changing inputs:
//            the plug-in replays the unchanged graph from a HIP graph (stderr under GGML_CDNA4_STATS: captures / replays); every result
//            against the CPU backend, and input A eager == input A replayed, bit for bit
//   split_harness <plugin.so> <type: q4_K|q4_0|q8_0|q5_K|q6_K|f16> <M> <K> <B> [resident | hostptr | shared | moe | moeffn]     -> one JSON line
//   shared   MUL_MATs that read the same src1 (wq / wk / wv; w_gate / w_up) take ONE activation quantization: hand-off count, output bytes' hash, time per graph
//   resident weights in the extra buffer type CDNA4_Resident (kernel-native images of the re-encoded formats, built once): types q5_0 q3_K q2_K q4_1 q5_1 iq4_nl iq4_xs, and q4_0 (a 16-byte-aligned re-layout for Q4_K's kernels)
#include "ggml.h"
#include "ggml-alloc.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static double rel_l2(const std::vector<float> & a, const std::vector<float> & b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); i++) { num += ((double)a[i] - b[i]) * ((double)a[i] - b[i]); den += (double)b[i] * b[i]; }
    return std::sqrt(num / (den > 0 ? den : 1));
}
static ggml_type parse_type(const std::string & s) {
    for (int t = 0; t < GGML_TYPE_COUNT; t++) if (s == ggml_type_name((ggml_type)t)) return (ggml_type)t;
    fprintf(stderr, "unknown type %s\n", s.c_str()); exit(2);
}
// Y = mul_mat(W, X) on `backend`, W living in `wbuft`
static std::vector<float> run_mul_mat(ggml_backend_t backend, ggml_backend_buffer_type_t wbuft, ggml_type type, int64_t M, int64_t K, int64_t B,
                                      const std::vector<uint8_t> & wq, const std::vector<float> & x, std::vector<uint8_t> * w_back) {
    ggml_init_params ip = { ggml_tensor_overhead() * 8 + ggml_graph_overhead(), NULL, true };
    ggml_context * wctx = ggml_init(ip), * cctx = ggml_init(ip);
    ggml_tensor * W = ggml_new_tensor_2d(wctx, type, K, M);
    ggml_backend_buffer_t wbuf = ggml_backend_alloc_ctx_tensors_from_buft(wctx, wbuft);
    if (!wbuf) { fprintf(stderr, "weight buffer allocation failed\n"); exit(1); }
    ggml_backend_buffer_set_usage(wbuf, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
    ggml_backend_tensor_set(W, wq.data(), 0, wq.size());
    if (w_back) { w_back->resize(wq.size()); ggml_backend_tensor_get(W, w_back->data(), 0, wq.size()); }
    ggml_tensor * X = ggml_new_tensor_2d(cctx, GGML_TYPE_F32, K, B);
    ggml_tensor * Y = ggml_mul_mat(cctx, W, X);
    ggml_cgraph * gf = ggml_new_graph(cctx);
    ggml_build_forward_expand(gf, Y);
    ggml_gallocr_t ga = ggml_gallocr_new(ggml_backend_get_default_buffer_type(backend));
    if (!ggml_gallocr_alloc_graph(ga, gf)) { fprintf(stderr, "graph allocation failed\n"); exit(1); }
    ggml_backend_tensor_set(X, x.data(), 0, x.size() * sizeof(float));
    if (ggml_backend_graph_compute(backend, gf) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); exit(1); }
    // twice: lanes, events and staging buffers are reused
    if (ggml_backend_graph_compute(backend, gf) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); exit(1); }
    std::vector<float> y((size_t)M * B);
    ggml_backend_tensor_get(Y, y.data(), 0, y.size() * sizeof(float));
    // the reference's own teardown order (examples/gpt-2/main-backend.cpp:936-939): the ggml_context — and with it every ggml_tensor — goes BEFORE the weight buffer; its
    // memory is scribbled over before the buffer is freed, so a buffer that still dereferences its tensors reads garbage (the CDNA4_Resident registry entry would
    // survive, and the next weights at the same address would multiply the freed image — the `rewritten` runs of the resident section see exactly that)
    ggml_gallocr_free(ga);
    const size_t ctx_bytes = ggml_tensor_overhead() * 8 + ggml_graph_overhead();
    ggml_free(wctx); ggml_free(cctx);
    void * scribble[4];
    for (void *& p : scribble) { p = malloc(ctx_bytes); if (p) memset(p, 0xAB, ctx_bytes); }
    ggml_backend_buffer_free(wbuf);
    for (void * p : scribble) free(p);
    return y;
}

// one MLP block on `backend`: y = x + W2 . gelu(W1 . (norm(x) * g + s) + b1) + b2, on ONE graph that is computed again and again
struct mlp_block {
    ggml_backend_t backend; int64_t D, B;
    ggml_context * wctx, * cctx; ggml_backend_buffer_t wbuf; ggml_gallocr_t ga; ggml_cgraph * gf; ggml_tensor * X, * out;
    mlp_block(ggml_backend_t be, ggml_type type, int64_t D_, int64_t H, int64_t B_, const std::vector<uint8_t> & w1, const std::vector<uint8_t> & w2, const std::vector<float> & vec)
        : backend(be), D(D_), B(B_) {
        ggml_init_params ip = { ggml_tensor_overhead() * 32 + ggml_graph_overhead(), NULL, true };
        wctx = ggml_init(ip); cctx = ggml_init(ip);
        ggml_tensor * W1 = ggml_new_tensor_2d(wctx, type, D, H), * W2 = ggml_new_tensor_2d(wctx, type, H, D);
        ggml_tensor * g = ggml_new_tensor_1d(wctx, GGML_TYPE_F32, D), * sft = ggml_new_tensor_1d(wctx, GGML_TYPE_F32, D);
        ggml_tensor * b1 = ggml_new_tensor_1d(wctx, GGML_TYPE_F32, H), * b2 = ggml_new_tensor_1d(wctx, GGML_TYPE_F32, D);
        wbuf = ggml_backend_alloc_ctx_tensors(wctx, backend);
        ggml_backend_tensor_set(W1, w1.data(), 0, w1.size()); ggml_backend_tensor_set(W2, w2.data(), 0, w2.size());
        ggml_backend_tensor_set(g, vec.data(), 0, D * 4); ggml_backend_tensor_set(sft, vec.data() + D, 0, D * 4);
        ggml_backend_tensor_set(b1, vec.data() + 2 * D, 0, H * 4); ggml_backend_tensor_set(b2, vec.data() + 2 * D + H, 0, D * 4);
        X = ggml_new_tensor_2d(cctx, GGML_TYPE_F32, D, B);
        ggml_set_input(X);
        ggml_tensor * cur = ggml_add(cctx, ggml_mul(cctx, ggml_norm(cctx, X, 1e-5f), g), sft);
        cur = ggml_gelu(cctx, ggml_add(cctx, ggml_mul_mat(cctx, W1, cur), b1));
        out = ggml_add(cctx, ggml_add(cctx, ggml_mul_mat(cctx, W2, cur), b2), X);
        ggml_set_output(out);
        gf = ggml_new_graph(cctx);
        ggml_build_forward_expand(gf, out);
        ga = ggml_gallocr_new(ggml_backend_get_default_buffer_type(backend));
        if (!ggml_gallocr_alloc_graph(ga, gf)) { fprintf(stderr, "graph allocation failed\n"); exit(1); }
    }
    std::vector<float> compute(const std::vector<float> & x) {
        ggml_backend_tensor_set(X, x.data(), 0, x.size() * sizeof(float));
        if (ggml_backend_graph_compute(backend, gf) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); exit(1); }
        std::vector<float> y((size_t)D * B);
        ggml_backend_tensor_get(out, y.data(), 0, y.size() * sizeof(float));
        return y;
    }
    ~mlp_block() { ggml_gallocr_free(ga); ggml_backend_buffer_free(wbuf); ggml_free(wctx); ggml_free(cctx); }
};

int main(int argc, char ** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s plugin type M K B\n", argv[0]); return 2; }
    const ggml_type type = parse_type(argv[2]);
    const int64_t M = atoll(argv[3]), K = atoll(argv[4]), B = atoll(argv[5]);
    ggml_backend_reg_t reg = ggml_backend_load(argv[1]);
    if (!reg) { fprintf(stderr, "cannot load %s\n", argv[1]); return 1; }
    ggml_backend_dev_t dev = ggml_backend_reg_dev_get(reg, 0);
    ggml_backend_t gpu = ggml_backend_dev_init(dev, NULL), gpu2 = ggml_backend_dev_init(dev, NULL);
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, NULL);
    if (!gpu || !gpu2 || !cpu) { fprintf(stderr, "backend init failed\n"); return 1; }

    // data: uniform(-1, 1) through ggml_quantize_chunk, as tests/test-backend-ops.cpp does
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> wf((size_t)M * K), x((size_t)B * K);
    for (auto & v : wf) v = u(rng);
    for (auto & v : x) v = u(rng);
    std::vector<uint8_t> wq(ggml_row_size(type, K) * M);
    if (type == GGML_TYPE_F32) memcpy(wq.data(), wf.data(), wq.size());
    else if (type == GGML_TYPE_F16) ggml_fp32_to_fp16_row(wf.data(), (ggml_fp16_t *)wq.data(), (int64_t)M * K);
    else ggml_quantize_chunk(type, wf.data(), wq.data(), 0, M, K, NULL);

    // ---- resident (argv[6] == "resident": only this section): weights in the device's first EXTRA buffer type ("ggml_backend_dev_get_extra_bufts", include/ggml-backend.h:192 —
    //      the plug-in's CDNA4_Resident: the re-encoded formats keep a kernel-native image built once at set_tensor) against the default buffer type and the CPU backend,
    //      at the given B (prefill) and at one row (decode reads the original bytes); set / get round trip; a second write of other weights rebuilds the image
    if (argc > 6 && std::string(argv[6]) == "resident") {
        typedef ggml_backend_buffer_type_t * (*extra_fn)(ggml_backend_dev_t);
        extra_fn get_extra = (extra_fn)ggml_backend_reg_get_proc_address(reg, "ggml_backend_dev_get_extra_bufts");
        if (!get_extra || !get_extra(dev) || !get_extra(dev)[0]) { fprintf(stderr, "no extra buffer types\n"); return 1; }
        ggml_backend_buffer_type_t rbuft = get_extra(dev)[0];
        if (!ggml_backend_dev_supports_buft(dev, rbuft) || ggml_backend_buft_is_host(rbuft)) { fprintf(stderr, "extra buffer type rejected\n"); return 1; }
        std::vector<uint8_t> rw_back;
        const std::vector<float> y_res = run_mul_mat(gpu, rbuft, type, M, K, B, wq, x, &rw_back);
        const std::vector<float> y_def = run_mul_mat(gpu, ggml_backend_dev_buffer_type(dev), type, M, K, B, wq, x, NULL);
        const std::vector<float> y_ref = run_mul_mat(cpu, ggml_backend_get_default_buffer_type(cpu), type, M, K, B, wq, x, NULL);
        std::vector<float> x1(x.begin(), x.begin() + K);
        const std::vector<float> y_res1 = run_mul_mat(gpu, rbuft, type, M, K, 1, wq, x1, NULL);
        const std::vector<float> y_def1 = run_mul_mat(gpu, ggml_backend_dev_buffer_type(dev), type, M, K, 1, wq, x1, NULL);
        // other weights through the same buffer type: the image must follow the bytes
        std::vector<float> wf2((size_t)M * K);
        for (auto & v : wf2) v = u(rng);
        std::vector<uint8_t> wq2(wq.size());
        ggml_quantize_chunk(type, wf2.data(), wq2.data(), 0, M, K, NULL);
        const std::vector<float> y_res2 = run_mul_mat(gpu, rbuft, type, M, K, B, wq2, x, NULL);
        const std::vector<float> y_def2 = run_mul_mat(gpu, ggml_backend_dev_buffer_type(dev), type, M, K, B, wq2, x, NULL);
        printf("{\"type\":\"%s\",\"M\":%lld,\"K\":%lld,\"B\":%lld,\"buft\":\"%s\",\"resident_vs_cpu_rel_l2\":%.3e,\"resident_bit_identical_to_default\":%s,\"decode_bit_identical_to_default\":%s,"
               "\"rewritten_bit_identical_to_default\":%s,\"set_get_roundtrip\":%s,\"resident_vs_default_rel_l2\":%.3e,\"rewritten_vs_default_rel_l2\":%.3e}\n", ggml_type_name(type), (long long)M, (long long)K, (long long)B, ggml_backend_buft_name(rbuft), rel_l2(y_res, y_ref),
               memcmp(y_res.data(), y_def.data(), y_def.size() * 4) == 0 ? "true" : "false", memcmp(y_res1.data(), y_def1.data(), y_def1.size() * 4) == 0 ? "true" : "false",
               memcmp(y_res2.data(), y_def2.data(), y_def2.size() * 4) == 0 ? "true" : "false", rw_back == wq ? "true" : "false", rel_l2(y_res, y_def), rel_l2(y_res2, y_def2));
        ggml_backend_free(gpu); ggml_backend_free(gpu2); ggml_backend_free(cpu);
        return 0;
    }

    // ---- shared (argv[6] == "shared": only this section; M = model width D, K = FFN width H): an attention + FFN front like every llama layer's —
    //      cur = rms_norm(X) * g;  Q = Wq cur, Kk = Wk cur (D / 4 rows: grouped-query), V = Wv cur + bv;  f = rms_norm(Q + X) * g;  out = Wd (silu(Wg f) * (Wu f)) —
    //      three MUL_MATs read `cur` and two read `f`: the plug-in quantizes each of them ONCE (the CPU backend: once per node, ggml-cpu.c:7490-7509).  Reports the hand-offs
    //      the first graph_compute took (proc address ggml_backend_cdna4_act_shared_count), FNV-1a of every output's bytes (the test runs it again under
    //      GGML_CDNA4_NO_ACT_SHARE=1: same bytes), each output against the CPU backend, and the time per graph (HIP-graph replay).
    if (argc > 6 && std::string(argv[6]) == "shared") {
        const int64_t D = M, H = K;
        int grouped = 0;
        auto build = [&](ggml_backend_t be, std::vector<std::vector<float>> & outs, double * us_per_graph, int * shared) {
            ggml_init_params ip = { ggml_tensor_overhead() * 64 + ggml_graph_overhead(), NULL, true };
            ggml_context * wctx = ggml_init(ip), * cctx = ggml_init(ip);
            ggml_tensor * Wq = ggml_new_tensor_2d(wctx, type, D, D), * Wk = ggml_new_tensor_2d(wctx, type, D, D / 4), * Wv = ggml_new_tensor_2d(wctx, type, D, D / 4);
            ggml_tensor * Wg = ggml_new_tensor_2d(wctx, type, D, H), * Wu = ggml_new_tensor_2d(wctx, type, D, H), * Wd = ggml_new_tensor_2d(wctx, type, H, D);
            ggml_tensor * g = ggml_new_tensor_1d(wctx, GGML_TYPE_F32, D), * bv = ggml_new_tensor_1d(wctx, GGML_TYPE_F32, D / 4);
            ggml_backend_buffer_t wbuf = ggml_backend_alloc_ctx_tensors(wctx, be);
            std::mt19937 r2(99);
            std::uniform_real_distribution<float> uu(-1.f, 1.f);
            for (ggml_tensor * W : { Wq, Wk, Wv, Wg, Wu, Wd }) {
                std::vector<float> f((size_t)ggml_nelements(W));
                const float sc = 2.0f / std::sqrt((float)W->ne[0]);
                for (auto & v : f) v = uu(r2) * sc;
                std::vector<uint8_t> q(ggml_nbytes(W));
                ggml_quantize_chunk(type, f.data(), q.data(), 0, W->ne[1], W->ne[0], NULL);
                ggml_backend_tensor_set(W, q.data(), 0, q.size());
            }
            std::vector<float> gv(D), bvv(D / 4);
            for (auto & v : gv) v = 1.f + 0.1f * uu(r2);
            for (auto & v : bvv) v = uu(r2);
            ggml_backend_tensor_set(g, gv.data(), 0, D * 4); ggml_backend_tensor_set(bv, bvv.data(), 0, D);
            ggml_tensor * X = ggml_new_tensor_2d(cctx, GGML_TYPE_F32, D, B);
            ggml_set_input(X);
            ggml_tensor * cur = ggml_mul(cctx, ggml_rms_norm(cctx, X, 1e-5f), g);
            ggml_tensor * Q = ggml_mul_mat(cctx, Wq, cur), * Kk = ggml_mul_mat(cctx, Wk, cur), * V = ggml_add(cctx, ggml_mul_mat(cctx, Wv, cur), bv);
            ggml_tensor * f = ggml_mul(cctx, ggml_rms_norm(cctx, ggml_add(cctx, Q, X), 1e-5f), g);
            ggml_tensor * gate = ggml_mul_mat(cctx, Wg, f), * up = ggml_mul_mat(cctx, Wu, f);
            ggml_tensor * out = ggml_mul_mat(cctx, Wd, ggml_mul(cctx, ggml_silu(cctx, gate), up));
            ggml_set_output(Kk); ggml_set_output(V); ggml_set_output(out);
            ggml_cgraph * gf = ggml_new_graph(cctx);
            ggml_build_forward_expand(gf, Kk); ggml_build_forward_expand(gf, V); ggml_build_forward_expand(gf, out);
            ggml_gallocr_t ga = ggml_gallocr_new(ggml_backend_get_default_buffer_type(be));
            if (!ggml_gallocr_alloc_graph(ga, gf)) { fprintf(stderr, "graph allocation failed\n"); exit(1); }
            std::vector<float> xin(x.begin(), x.begin() + (size_t)D * B);
            typedef int (*count_fn)(void);
            count_fn cnt = be == gpu ? (count_fn)ggml_backend_reg_get_proc_address(reg, "ggml_backend_cdna4_act_shared_count") : nullptr;
            count_fn gcnt = be == gpu ? (count_fn)ggml_backend_reg_get_proc_address(reg, "ggml_backend_cdna4_grouped_count") : nullptr;      // one-row MUL_MATs that rode in another's launch
            const int c0 = cnt ? cnt() : 0, g0 = gcnt ? gcnt() : 0;
            ggml_backend_tensor_set(X, xin.data(), 0, xin.size() * 4);
            if (ggml_backend_graph_compute(be, gf) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); exit(1); }
            ggml_backend_synchronize(be);
            if (shared) *shared = cnt ? cnt() - c0 : -1;
            if (be == gpu) grouped = gcnt ? gcnt() - g0 : -1;
            // (the outputs of the FIRST compute: the graph allocator may hand X's memory to a later node once X's last reader has run, so a second compute of the same
            //  graph without a new tensor_set starts from other activations — the timing loop below does exactly that, on purpose)
            outs.clear();
            for (ggml_tensor * t : { Kk, V, out }) { std::vector<float> y((size_t)ggml_nelements(t)); ggml_backend_tensor_get(t, y.data(), 0, y.size() * 4); outs.push_back(y); }
            if (us_per_graph && !getenv("HARNESS_NO_TIMING")) {                        // (HARNESS_NO_TIMING: CPU-emulated runs of the plug-in skip the 55 timed computes)
                for (int i = 0; i < 5; i++) ggml_backend_graph_compute(be, gf);          // (second appearance: captured; then replays)
                ggml_backend_synchronize(be);
                const int64_t t0 = ggml_time_us();
                const int n = 50;
                for (int i = 0; i < n; i++) ggml_backend_graph_compute(be, gf);
                ggml_backend_synchronize(be);
                *us_per_graph = (double)(ggml_time_us() - t0) / n;
            }
            ggml_gallocr_free(ga); ggml_backend_buffer_free(wbuf); ggml_free(wctx); ggml_free(cctx);
        };
        ggml_time_init();
        std::vector<std::vector<float>> y_gpu, y_cpu;
        double us = 0; int shared = 0;
        build(gpu, y_gpu, &us, &shared);
        if (getenv("HARNESS_NO_CPU")) y_cpu = y_gpu; else build(cpu, y_cpu, nullptr, nullptr);      // (timing runs at full size skip the CPU backend's pass)
        uint64_t h = 1469598103934665603ull;
        for (const auto & y : y_gpu) for (size_t i = 0; i < y.size() * 4; i++) { h ^= ((const uint8_t *)y.data())[i]; h *= 1099511628211ull; }
        long long words_differing = 0;                                       // fp32 words of K, V and out whose BITS differ from the CPU backend's (0 under GGML_CDNA4_EXACT=1)
        for (size_t o = 0; o < y_gpu.size(); o++) for (size_t i = 0; i < y_gpu[o].size(); i++) words_differing += memcmp(&y_gpu[o][i], &y_cpu[o][i], 4) != 0;
        printf("{\"type\":\"%s\",\"D\":%lld,\"H\":%lld,\"B\":%lld,\"act_hand_offs_first_compute\":%d,\"grouped_first_compute\":%d,\"fnv1a\":\"%016llx\",\"us_per_graph\":%.2f,\"k_vs_cpu\":%.3e,\"v_vs_cpu\":%.3e,\"out_vs_cpu\":%.3e,\"words_differing_from_cpu\":%lld}\n",
               ggml_type_name(type), (long long)D, (long long)H, (long long)B, shared, grouped, (unsigned long long)h, us, rel_l2(y_gpu[0], y_cpu[0]), rel_l2(y_gpu[1], y_cpu[1]), rel_l2(y_gpu[2], y_cpu[2]), words_differing);
        ggml_backend_free(gpu); ggml_backend_free(gpu2); ggml_backend_free(cpu);
        return 0;
    }

    // ---- moe (argv[6] == "moe": only this section; B = tokens): MUL_MAT_ID over an expert stack [K, M, 4 experts], two used per token, the stack living in the device's
    //      default buffer type and in its first EXTRA buffer type (CDNA4_Resident: a Q4_0 stack gets ONE resident image, found by the stack's pointer) — each against the
    //      CPU backend, and against each other; prefill size (B tokens) and a single token
    if (argc > 6 && std::string(argv[6]) == "moe") {
        typedef ggml_backend_buffer_type_t * (*extra_fn)(ggml_backend_dev_t);
        extra_fn get_extra = (extra_fn)ggml_backend_reg_get_proc_address(reg, "ggml_backend_dev_get_extra_bufts");
        if (!get_extra || !get_extra(dev) || !get_extra(dev)[0]) { fprintf(stderr, "no extra buffer types\n"); return 1; }
        const int64_t NE = 4, NU = 2;
        std::vector<float> wf3((size_t)M * K * NE);
        for (auto & v : wf3) v = u(rng);
        std::vector<uint8_t> wq3(ggml_row_size(type, K) * M * NE);
        ggml_quantize_chunk(type, wf3.data(), wq3.data(), 0, M * NE, K, NULL);
        auto run = [&](ggml_backend_t be, ggml_backend_buffer_type_t wbuft, int64_t NT) {
            ggml_init_params ip = { ggml_tensor_overhead() * 16 + ggml_graph_overhead(), NULL, true };
            ggml_context * wctx = ggml_init(ip), * cctx = ggml_init(ip);
            ggml_tensor * AS = ggml_new_tensor_3d(wctx, type, K, M, NE);
            ggml_backend_buffer_t wbuf = ggml_backend_alloc_ctx_tensors_from_buft(wctx, wbuft);
            if (!wbuf) { fprintf(stderr, "weight buffer allocation failed\n"); exit(1); }
            ggml_backend_buffer_set_usage(wbuf, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
            ggml_backend_tensor_set(AS, wq3.data(), 0, wq3.size());
            ggml_tensor * Bt = ggml_new_tensor_3d(cctx, GGML_TYPE_F32, K, NU, NT);
            ggml_tensor * IDS = ggml_new_tensor_2d(cctx, GGML_TYPE_I32, NU, NT);
            ggml_set_input(Bt); ggml_set_input(IDS);
            ggml_tensor * Y = ggml_mul_mat_id(cctx, AS, Bt, IDS);
            ggml_set_output(Y);
            ggml_cgraph * gf = ggml_new_graph(cctx);
            ggml_build_forward_expand(gf, Y);
            ggml_gallocr_t ga = ggml_gallocr_new(ggml_backend_get_default_buffer_type(be));
            if (!ggml_gallocr_alloc_graph(ga, gf)) { fprintf(stderr, "graph allocation failed\n"); exit(1); }
            std::vector<float> xb((size_t)K * NU * NT);
            std::mt19937 r3(7);
            for (auto & v : xb) v = u(r3);
            std::vector<int32_t> ids((size_t)NU * NT);
            for (int64_t t = 0; t < NT; t++) { const int e0 = (int)(r3() % NE); ids[t * NU] = e0; ids[t * NU + 1] = (e0 + 1 + (int)(r3() % (NE - 1))) % (int)NE; }
            ggml_backend_tensor_set(Bt, xb.data(), 0, xb.size() * 4); ggml_backend_tensor_set(IDS, ids.data(), 0, ids.size() * 4);
            if (ggml_backend_graph_compute(be, gf) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); exit(1); }
            std::vector<float> y((size_t)M * NU * NT);
            ggml_backend_tensor_get(Y, y.data(), 0, y.size() * 4);
            ggml_gallocr_free(ga); ggml_backend_buffer_free(wbuf); ggml_free(wctx); ggml_free(cctx);
            return y;
        };
        ggml_backend_buffer_type_t rbuft = get_extra(dev)[0], dbuft = ggml_backend_dev_buffer_type(dev), cbuft = ggml_backend_get_default_buffer_type(cpu);
        const std::vector<float> y_res = run(gpu, rbuft, B), y_def = run(gpu, dbuft, B), y_cpu = run(cpu, cbuft, B);
        const std::vector<float> y_res1 = run(gpu, rbuft, 1), y_def1 = run(gpu, dbuft, 1), y_cpu1 = run(cpu, cbuft, 1);
        printf("{\"type\":\"%s\",\"M\":%lld,\"K\":%lld,\"tokens\":%lld,\"n_expert\":4,\"n_used\":2,\"buft\":\"%s\",\"resident_vs_cpu_rel_l2\":%.3e,\"default_vs_cpu_rel_l2\":%.3e,\"resident_vs_default_rel_l2\":%.3e,"
               "\"resident_bit_identical_to_default\":%s,\"one_token_vs_cpu_rel_l2\":%.3e,\"one_token_bit_identical_to_default\":%s}\n", ggml_type_name(type), (long long)M, (long long)K, (long long)B,
               ggml_backend_buft_name(rbuft), rel_l2(y_res, y_cpu), rel_l2(y_def, y_cpu), rel_l2(y_res, y_def), memcmp(y_res.data(), y_def.data(), y_def.size() * 4) == 0 ? "true" : "false",
               rel_l2(y_res1, y_cpu1), memcmp(y_res1.data(), y_def1.data(), y_def1.size() * 4) == 0 ? "true" : "false");
        ggml_backend_free(gpu); ggml_backend_free(gpu2); ggml_backend_free(cpu);
        return 0;
    }

    // ---- moeffn (argv[6] == "moeffn": only this section; M = model width D, K = expert FFN width H, B = tokens): a mixture-of-experts FFN as llama.cpp's build_moe_ffn issues it —
    //      up = mul_mat_id(UP, cur, ids), gate = mul_mat_id(GATE, cur, ids), down = mul_mat_id(DOWN, up * silu(gate), ids); 8 experts, 2 used.  The up and gate stacks read the
    //      same (cur, ids): the second multiplies the first one's front (proc address ggml_backend_cdna4_moe_front_shared_count); GGML_CDNA4_NO_ACT_SHARE=1 for the A/B twin.
    if (argc > 6 && std::string(argv[6]) == "moeffn") {
        const int64_t D = M, H = K, NE = 8, NU = 2, NT = B;
        std::vector<std::vector<uint8_t>> stacks;
        for (int i = 0; i < 3; i++) {
            const int64_t rows = (i < 2 ? H : D) * NE, cols = i < 2 ? D : H;
            std::vector<float> wf3((size_t)rows * cols);
            for (auto & v : wf3) v = u(rng);
            std::vector<uint8_t> q(ggml_row_size(type, cols) * rows);
            ggml_quantize_chunk(type, wf3.data(), q.data(), 0, rows, cols, NULL);
            stacks.push_back(q);
        }
        int shared = -1;
        auto run = [&](ggml_backend_t be, double * us_per_graph) {
            ggml_init_params ip = { ggml_tensor_overhead() * 32 + ggml_graph_overhead(), NULL, true };
            ggml_context * wctx = ggml_init(ip), * cctx = ggml_init(ip);
            ggml_tensor * UP = ggml_new_tensor_3d(wctx, type, D, H, NE), * GATE = ggml_new_tensor_3d(wctx, type, D, H, NE), * DOWN = ggml_new_tensor_3d(wctx, type, H, D, NE);
            ggml_backend_buffer_t wbuf = ggml_backend_alloc_ctx_tensors(wctx, be);
            if (!wbuf) { fprintf(stderr, "weight buffer allocation failed\n"); exit(1); }
            ggml_backend_buffer_set_usage(wbuf, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
            ggml_backend_tensor_set(UP, stacks[0].data(), 0, stacks[0].size()); ggml_backend_tensor_set(GATE, stacks[1].data(), 0, stacks[1].size()); ggml_backend_tensor_set(DOWN, stacks[2].data(), 0, stacks[2].size());
            ggml_tensor * CUR = ggml_new_tensor_3d(cctx, GGML_TYPE_F32, D, 1, NT);
            ggml_tensor * IDS = ggml_new_tensor_2d(cctx, GGML_TYPE_I32, NU, NT);
            ggml_set_input(CUR); ggml_set_input(IDS);
            ggml_tensor * up = ggml_mul_mat_id(cctx, UP, CUR, IDS), * gate = ggml_mul_mat_id(cctx, GATE, CUR, IDS);
            ggml_tensor * out = ggml_mul_mat_id(cctx, DOWN, ggml_mul(cctx, up, ggml_silu(cctx, gate)), IDS);
            ggml_set_output(out);
            ggml_cgraph * gf = ggml_new_graph(cctx);
            ggml_build_forward_expand(gf, out);
            ggml_gallocr_t ga = ggml_gallocr_new(ggml_backend_get_default_buffer_type(be));
            if (!ggml_gallocr_alloc_graph(ga, gf)) { fprintf(stderr, "graph allocation failed\n"); exit(1); }
            std::vector<float> xb((size_t)D * NT);
            std::mt19937 r3(7);
            for (auto & v : xb) v = u(r3);
            std::vector<int32_t> ids((size_t)NU * NT);
            for (int64_t t = 0; t < NT; t++) { const int e0 = (int)(r3() % NE); ids[t * NU] = e0; ids[t * NU + 1] = (e0 + 1 + (int)(r3() % (NE - 1))) % (int)NE; }
            typedef int (*count_fn)(void);
            count_fn cnt = be == gpu ? (count_fn)ggml_backend_reg_get_proc_address(reg, "ggml_backend_cdna4_moe_front_shared_count") : nullptr;
            const int c0 = cnt ? cnt() : 0;
            ggml_backend_tensor_set(CUR, xb.data(), 0, xb.size() * 4); ggml_backend_tensor_set(IDS, ids.data(), 0, ids.size() * 4);
            if (ggml_backend_graph_compute(be, gf) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); exit(1); }
            ggml_backend_synchronize(be);
            if (be == gpu) shared = cnt ? cnt() - c0 : -1;
            std::vector<float> y((size_t)ggml_nelements(out));
            ggml_backend_tensor_get(out, y.data(), 0, y.size() * 4);
            if (us_per_graph && !getenv("HARNESS_NO_TIMING")) {
                for (int i = 0; i < 5; i++) { ggml_backend_tensor_set(CUR, xb.data(), 0, xb.size() * 4); ggml_backend_graph_compute(be, gf); }
                ggml_backend_synchronize(be);
                const int n = 50;
                const int64_t t0 = ggml_time_us();
                for (int i = 0; i < n; i++) ggml_backend_graph_compute(be, gf);
                ggml_backend_synchronize(be);
                *us_per_graph = (double)(ggml_time_us() - t0) / n;
            }
            ggml_gallocr_free(ga); ggml_backend_buffer_free(wbuf); ggml_free(wctx); ggml_free(cctx);
            return y;
        };
        ggml_time_init();
        double us = 0;
        const std::vector<float> y_gpu = run(gpu, &us);
        const std::vector<float> y_cpu = getenv("HARNESS_NO_CPU") ? y_gpu : run(cpu, nullptr);
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < y_gpu.size() * 4; i++) { h ^= ((const uint8_t *)y_gpu.data())[i]; h *= 1099511628211ull; }
        printf("{\"type\":\"%s\",\"D\":%lld,\"H\":%lld,\"tokens\":%lld,\"n_expert\":8,\"n_used\":2,\"moe_fronts_shared_first_compute\":%d,\"fnv1a\":\"%016llx\",\"us_per_graph\":%.2f,\"out_vs_cpu\":%.3e}\n",
               ggml_type_name(type), (long long)D, (long long)H, (long long)NT, shared, (unsigned long long)h, us, rel_l2(y_gpu, y_cpu));
        ggml_backend_free(gpu); ggml_backend_free(gpu2); ggml_backend_free(cpu);
        return 0;
    }

    // ---- hostptr (argv[6] == "hostptr": only this section): ggml_backend_dev_buffer_from_host_ptr (include/ggml-backend.h:170) over page-aligned memory of this process — the
    //      weight tensor is placed in it by HOST address, filled with a plain memcpy, and multiplied in place by the device; against the default buffer type (bit for bit)
    if (argc > 6 && std::string(argv[6]) == "hostptr") {
        ggml_backend_dev_props props; ggml_backend_dev_get_props(dev, &props);
        const size_t bytes = (wq.size() + 4095) / 4096 * 4096 + 4096;
        void * host = nullptr;
        if (posix_memalign(&host, 4096, bytes) != 0) { fprintf(stderr, "posix_memalign failed\n"); return 1; }
        ggml_backend_buffer_t hbuf = props.caps.buffer_from_host_ptr ? ggml_backend_dev_buffer_from_host_ptr(dev, host, bytes, bytes) : NULL;
        if (!hbuf) { printf("{\"type\":\"%s\",\"caps_buffer_from_host_ptr\":%s,\"declined\":true}\n", ggml_type_name(type), props.caps.buffer_from_host_ptr ? "true" : "false"); free(host); return 0; }
        ggml_init_params ip = { ggml_tensor_overhead() * 8 + ggml_graph_overhead(), NULL, true };
        ggml_context * wctx = ggml_init(ip), * cctx = ggml_init(ip);
        ggml_tensor * W = ggml_new_tensor_2d(wctx, type, K, M);
        ggml_tallocr ta = ggml_tallocr_new(hbuf);
        ggml_tallocr_alloc(&ta, W);
        const bool in_place = W->data == ggml_backend_buffer_get_base(hbuf) && (char *)W->data >= (char *)host && (char *)W->data < (char *)host + bytes;
        memcpy(W->data, wq.data(), wq.size());                                    // the host writes its own memory
        std::vector<uint8_t> back(wq.size()); ggml_backend_tensor_get(W, back.data(), 0, back.size());
        ggml_tensor * X = ggml_new_tensor_2d(cctx, GGML_TYPE_F32, K, B);
        ggml_tensor * Y = ggml_mul_mat(cctx, W, X);
        ggml_cgraph * gf = ggml_new_graph(cctx);
        ggml_build_forward_expand(gf, Y);
        ggml_gallocr_t ga = ggml_gallocr_new(ggml_backend_get_default_buffer_type(gpu));
        if (!ggml_gallocr_alloc_graph(ga, gf)) { fprintf(stderr, "graph allocation failed\n"); return 1; }
        ggml_backend_tensor_set(X, x.data(), 0, x.size() * sizeof(float));
        if (ggml_backend_graph_compute(gpu, gf) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); return 1; }
        std::vector<float> y_host((size_t)M * B);
        ggml_backend_tensor_get(Y, y_host.data(), 0, y_host.size() * sizeof(float));
        const std::vector<float> y_def = run_mul_mat(gpu, ggml_backend_dev_buffer_type(dev), type, M, K, B, wq, x, NULL);
        printf("{\"type\":\"%s\",\"M\":%lld,\"K\":%lld,\"B\":%lld,\"caps_buffer_from_host_ptr\":true,\"declined\":false,\"tensor_in_host_range\":%s,\"get_roundtrip\":%s,\"bit_identical_to_default\":%s}\n",
               ggml_type_name(type), (long long)M, (long long)K, (long long)B, in_place ? "true" : "false", back == wq ? "true" : "false", memcmp(y_host.data(), y_def.data(), y_def.size() * 4) == 0 ? "true" : "false");
        ggml_gallocr_free(ga); ggml_backend_buffer_free(hbuf); ggml_free(wctx); ggml_free(cctx); free(host);
        ggml_backend_free(gpu); ggml_backend_free(gpu2); ggml_backend_free(cpu);
        return 0;
    }

    // ---- split
    typedef ggml_backend_buffer_type_t (*split_fn)(int, const float *);
    split_fn get_split = (split_fn)ggml_backend_reg_get_proc_address(reg, "ggml_backend_split_buffer_type");
    if (!get_split) { fprintf(stderr, "ggml_backend_split_buffer_type is not exported\n"); return 1; }
    ggml_backend_buffer_type_t sbuft = get_split(0, NULL);
    if (!sbuft || !ggml_backend_dev_supports_buft(dev, sbuft) || ggml_backend_buft_is_host(sbuft)) { fprintf(stderr, "split buffer type rejected\n"); return 1; }
    std::vector<uint8_t> w_back;
    const std::vector<float> y_split = run_mul_mat(gpu, sbuft, type, M, K, B, wq, x, &w_back);
    const std::vector<float> y_plain = run_mul_mat(gpu, ggml_backend_dev_buffer_type(dev), type, M, K, B, wq, x, NULL);
    const std::vector<float> y_cpu = run_mul_mat(cpu, ggml_backend_get_default_buffer_type(cpu), type, M, K, B, wq, x, NULL);
    const bool roundtrip = w_back == wq;
    const bool same_as_plain = memcmp(y_split.data(), y_plain.data(), y_plain.size() * 4) == 0;
    // ---- K split (the plug-in's own proc address; quantized types with whole superblocks per shard): partial products over K ranges + their sum
    //      (RCCL all-reduce across devices, ggml_cdna4_sum_partials where the shards share a device)
    double ks_vs_cpu = -1, ks_vs_plain = -1; bool ks_roundtrip = false, ks_repeat = false;
    split_fn get_ksplit = (split_fn)ggml_backend_reg_get_proc_address(reg, "ggml_backend_cdna4_ksplit_buffer_type");
    if (get_ksplit && ggml_is_quantized(type) && K % 256 == 0) {
        ggml_backend_buffer_type_t kbuft = get_ksplit(0, NULL);
        if (!kbuft || !ggml_backend_dev_supports_buft(dev, kbuft) || ggml_backend_buft_is_host(kbuft)) { fprintf(stderr, "K-split buffer type rejected\n"); return 1; }
        std::vector<uint8_t> kw_back;
        const std::vector<float> y_k = run_mul_mat(gpu, kbuft, type, M, K, B, wq, x, &kw_back);
        const std::vector<float> y_k2 = run_mul_mat(gpu, kbuft, type, M, K, B, wq, x, NULL);
        ks_roundtrip = kw_back == wq;
        ks_repeat = memcmp(y_k.data(), y_k2.data(), y_k.size() * 4) == 0;
        ks_vs_cpu = rel_l2(y_k, y_cpu); ks_vs_plain = rel_l2(y_k, y_plain);
    }

    // ---- async copies + events between two backends (streams) of the device
    bool async_ok = true;
    {
        ggml_init_params ip = { ggml_tensor_overhead() * 4, NULL, true };
        ggml_context * c = ggml_init(ip);
        ggml_tensor * a = ggml_new_tensor_1d(c, GGML_TYPE_F32, 1 << 20), * b = ggml_new_tensor_1d(c, GGML_TYPE_F32, 1 << 20);
        ggml_backend_buffer_t buf = ggml_backend_alloc_ctx_tensors_from_buft(c, ggml_backend_dev_buffer_type(dev));
        std::vector<float> src(1 << 20), dst(1 << 20, 0.f);
        for (size_t i = 0; i < src.size(); i++) src[i] = (float)i * 0.5f;
        ggml_backend_tensor_set_async(gpu, a, src.data(), 0, src.size() * 4);
        ggml_backend_tensor_copy_async(gpu, gpu2, a, b);                      // gpu2's stream waits for the copy issued on gpu's
        ggml_backend_event_t ev = ggml_backend_event_new(dev);
        if (!ev) async_ok = false;
        else {
            ggml_backend_event_record(ev, gpu2);
            ggml_backend_event_wait(gpu, ev);
            ggml_backend_tensor_get_async(gpu, b, dst.data(), 0, dst.size() * 4);
            ggml_backend_synchronize(gpu);
            ggml_backend_event_synchronize(ev);
            ggml_backend_event_free(ev);
            async_ok = dst == src;
        }
        ggml_backend_buffer_free(buf); ggml_free(c);
    }
    // ---- pinned host buffer type
    bool host_ok = false;
    {
        ggml_backend_buffer_type_t hb = ggml_backend_dev_host_buffer_type(dev);
        if (hb && ggml_backend_buft_is_host(hb)) {
            ggml_backend_buffer_t buf = ggml_backend_buft_alloc_buffer(hb, 1 << 20);
            if (buf) { memset(ggml_backend_buffer_get_base(buf), 0x5a, 1 << 20); host_ok = ((uint8_t *)ggml_backend_buffer_get_base(buf))[12345] == 0x5a; ggml_backend_buffer_free(buf); }
        }
    }
    // ---- replay: decode-sized (1 row: one-launch GEMV with the tail in its store) and prefill-sized (96 rows: MFMA GEMM + split-K exchange)
    bool replay_ok = true; double replay_worst = 0;
    if (ggml_is_quantized(type)) {
        const int64_t D = 1024, H = 4096;
        std::vector<float> f1((size_t)D * H), vec(3 * D + H);
        for (auto & v : f1) v = u(rng) * 0.05f;
        for (auto & v : vec) v = u(rng);
        std::vector<uint8_t> w1(ggml_row_size(type, D) * H), w2(ggml_row_size(type, H) * D);
        ggml_quantize_chunk(type, f1.data(), w1.data(), 0, H, D, NULL);
        ggml_quantize_chunk(type, f1.data(), w2.data(), 0, D, H, NULL);
        for (int64_t Bn : {(int64_t)1, (int64_t)96}) {
            std::vector<std::vector<float>> in(6, std::vector<float>((size_t)D * Bn));
            for (int i = 0; i < 6; i++) for (auto & v : in[i]) v = u(rng);
            in[4] = in[0];                                                     // input A again, after the capture
            mlp_block bg(gpu, type, D, H, Bn, w1, w2, vec), bc(cpu, type, D, H, Bn, w1, w2, vec);
            std::vector<std::vector<float>> yg, yc;
            for (int i = 0; i < 6; i++) { yg.push_back(bg.compute(in[i])); yc.push_back(bc.compute(in[i])); }
            for (int i = 0; i < 6; i++) { const double e = rel_l2(yg[i], yc[i]); if (e > replay_worst) replay_worst = e; if (!(e < 1e-2)) replay_ok = false; }      // (two chained quantized products and an fp16-table GELU: the CPU's own formats differ at 1e-3)
            if (memcmp(yg[0].data(), yg[4].data(), yg[0].size() * 4) != 0) replay_ok = false;
        }
        // two graphs taking turns on one backend (what ggml_backend_sched's splits look like to it): both are captured on their second
        // appearance and replayed on the third, each from its own slot
        {
            std::vector<float> xa((size_t)D * 2), xb((size_t)D * 160);
            for (auto & v : xa) v = u(rng);
            for (auto & v : xb) v = u(rng);
            mlp_block a(gpu, type, D, H, 2, w1, w2, vec), b(gpu, type, D, H, 160, w1, w2, vec);
            std::vector<float> ya[3], yb[3];
            for (int i = 0; i < 3; i++) { ya[i] = a.compute(xa); yb[i] = b.compute(xb); }                     // eager, captured, replayed
            for (int i = 1; i < 3; i++)
                if (memcmp(ya[0].data(), ya[i].data(), ya[0].size() * 4) != 0 || memcmp(yb[0].data(), yb[i].data(), yb[0].size() * 4) != 0) { replay_ok = false; fprintf(stderr, "alternating graphs: replay differs\n"); }
        }
        // a captured graph holds the addresses of the backend's workspace and of the kernel library's scratch: a LARGER graph computed in
        // between moves both, and the small graph must be re-captured, not replayed with the stale addresses
        {
            std::vector<float> xs((size_t)D * 96), xl((size_t)D * 8192);
            for (auto & v : xs) v = u(rng);
            for (auto & v : xl) v = u(rng);
            mlp_block small(gpu2, type, D, H, 96, w1, w2, vec);
            const std::vector<float> y0 = small.compute(xs); small.compute(xs); small.compute(xs);          // eager, capture, replay
            { mlp_block large(gpu2, type, D, H, 8192, w1, w2, vec); large.compute(xl); }   // ONE eager run: grows workspace + split-K scratch, the small graph's exec stays cached
            for (int i = 0; i < 3; i++) {                                                 // stale exec dropped: re-captured at once (nothing left to size), replayed, replayed
                const std::vector<float> y1 = small.compute(xs);
                if (memcmp(y0.data(), y1.data(), y0.size() * 4) != 0) { replay_ok = false; fprintf(stderr, "replay after a workspace move differs\n"); }
            }
        }
    }
    ggml_backend_dev_props props; ggml_backend_dev_get_props(dev, &props);
    printf("{\"ksplit_vs_cpu_rel_l2\":%.3e,\"ksplit_vs_plain_rel_l2\":%.3e,\"ksplit_set_get_roundtrip\":%s,\"ksplit_deterministic\":%s,", ks_vs_cpu, ks_vs_plain, ks_roundtrip ? "true" : "false", ks_repeat ? "true" : "false");
    printf("\"type\":\"%s\",\"M\":%lld,\"K\":%lld,\"B\":%lld,\"split_vs_cpu_rel_l2\":%.3e,\"plain_vs_cpu_rel_l2\":%.3e,\"split_vs_plain_rel_l2\":%.3e,\"split_bit_identical_to_plain\":%s,"
           "\"set_get_roundtrip\":%s,\"graph_replay_ok\":%s,\"graph_replay_worst_rel_l2\":%.3e,\"async_ok\":%s,\"host_buffer_ok\":%s,\"caps_async\":%s,\"caps_host_buffer\":%s,\"caps_events\":%s}\n",
           ggml_type_name(type), (long long)M, (long long)K, (long long)B, rel_l2(y_split, y_cpu), rel_l2(y_plain, y_cpu), rel_l2(y_split, y_plain), same_as_plain ? "true" : "false",
           roundtrip ? "true" : "false", replay_ok ? "true" : "false", replay_worst, async_ok ? "true" : "false", host_ok ? "true" : "false", props.caps.async ? "true" : "false", props.caps.host_buffer ? "true" : "false", props.caps.events ? "true" : "false");
    ggml_backend_free(gpu); ggml_backend_free(gpu2); ggml_backend_free(cpu);
    return 0;
}
