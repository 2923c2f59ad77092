// oracle/gpt2_harness.cpp — TEST INFRASTRUCTURE ONLY (never part of the product).
// BASELINE config #4: runs the reference's OWN examples/gpt-2 graph builder, model loader and eval loop
// (examples/gpt-2/main-backend.cpp, included verbatim below, unmodified) on a chosen ggml backend and dumps the
// logits, so the same Q4_0 model can be evaluated on the reference CPU backend and on our plug-in and compared.
// The only thing this wrapper does that the reference main() does not: it pre-sets model.backend — the loader
// only falls back to the CPU backend `if (!model.backend)` (main-backend.cpp:223) — and it feeds a FIXED token
// sequence instead of sampling, so both runs see identical inputs.
//
//   gpt2_harness <model.bin> <backend: CPU | device name e.g. CDNA40> <plugin.so | -> <out.bin> <n_prompt> <n_decode> <threads>
#define main gpt2_reference_main
#include "examples/gpt-2/main-backend.cpp"
#undef main

// per-node comparison of the gpt-2 graph between the chosen backend and the CPU backend (what test-backend-ops does
// for single ops, ggml_backend_compare_graph_backend, src/ggml-backend.cpp:1814-1851) — prints rel-L2 per node
static bool cmp_cb(int index, struct ggml_tensor * t1, struct ggml_tensor * t2, void * ud) {
    const bool resync = ud != NULL;     // RESYNC: after comparing, overwrite our result with the CPU's, so every op is
                                        // judged on IDENTICAL inputs (per-op parity, no error accumulation)
    if (t1->type != GGML_TYPE_F32) return true;
    const size_t n = ggml_nelements(t1);
    if (!ggml_is_contiguous(t1) || !ggml_is_contiguous(t2)) { printf("node %4d %-14s %-24s (non-contiguous, skipped)\n", index, ggml_op_desc(t1), t1->name); return true; }
    std::vector<float> a(n), b(n);
    ggml_backend_tensor_get(t1, a.data(), 0, n * 4);
    ggml_backend_tensor_get(t2, b.data(), 0, n * 4);
    double num = 0, den = 0; size_t nbad = 0;
    for (size_t i = 0; i < n; i++) {
        if (std::isinf(b[i]) && a[i] == b[i]) continue;
        if (!std::isfinite(a[i]) || !std::isfinite(b[i])) { nbad++; continue; }
        num += ((double)a[i] - b[i]) * ((double)a[i] - b[i]); den += (double)b[i] * b[i];
    }
    printf("node %4d %-14s %-24s [%5lld,%5lld,%3lld] rel_l2=%.3e%s\n", index, ggml_op_desc(t1), t1->name, (long long)t1->ne[0], (long long)t1->ne[1], (long long)t1->ne[2],
           den > 0 ? sqrt(num / den) : sqrt(num), nbad ? " NONFINITE-MISMATCH" : "");
    if (resync) ggml_backend_tensor_set(t1, b.data(), 0, n * 4);
    return true;
}

int main(int argc, char ** argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s model backend plugin out n_prompt n_decode threads\n", argv[0]); return 2; }
    const std::string fname = argv[1], backend = argv[2], plugin = argv[3], out = argv[4];
    const int n_prompt = atoi(argv[5]), n_decode = atoi(argv[6]), n_threads = atoi(argv[7]);
    ggml_time_init();

    gpt_vocab vocab;
    gpt2_model model;
    if (backend != "CPU") {
        if (plugin != "-" && !ggml_backend_load(plugin.c_str())) { fprintf(stderr, "failed to load %s\n", plugin.c_str()); return 1; }
        model.backend = ggml_backend_init_by_name(backend.c_str(), NULL);
        if (!model.backend) { fprintf(stderr, "backend %s not found\n", backend.c_str()); return 1; }
    }
    if (!gpt2_model_load(fname, model, vocab, 1024, 0)) { fprintf(stderr, "failed to load model\n"); return 1; }
    fprintf(stderr, "harness: backend = %s\n", ggml_backend_name(model.backend));

    ggml_gallocr_t allocr = ggml_gallocr_new(ggml_backend_get_default_buffer_type(model.backend));
    {   // worst-case graph reservation exactly as main-backend.cpp:826-841
        int n_tokens = std::min(model.hparams.n_ctx, std::max(n_prompt, 8));
        int n_past = model.hparams.n_ctx - n_tokens;
        ggml_gallocr_reserve(allocr, gpt2_graph(model, n_past, n_tokens));
    }
    // fixed pseudo-random token ids (LCG), identical for every backend
    uint32_t st = 12345;
    auto next_tok = [&]() { st = st * 1664525u + 1013904223u; return (gpt_vocab::id)((st >> 8) % (uint32_t)model.hparams.n_vocab); };
    std::vector<gpt_vocab::id> prompt(n_prompt);
    for (auto & t : prompt) t = next_tok();

    if (out == "PERTURB") {     // conditioning of the REFERENCE itself: CPU backend vs CPU backend with the first LayerNorm gain scaled by (1 + 1e-6)
        std::vector<float> la, lb;
        if (!gpt2_eval(model, allocr, n_threads, 0, prompt, la)) return 1;
        struct ggml_tensor * t = model.layers[0].ln_1_g;
        std::vector<float> w(ggml_nelements(t));
        ggml_backend_tensor_get(t, w.data(), 0, ggml_nbytes(t));
        for (auto & v : w) v *= 1.000001f;
        ggml_backend_tensor_set(t, w.data(), 0, ggml_nbytes(t));
        if (!gpt2_eval(model, allocr, n_threads, 0, prompt, lb)) return 1;
        double num = 0, den = 0; for (size_t i = 0; i < la.size(); i++) { num += ((double)la[i] - lb[i]) * ((double)la[i] - lb[i]); den += (double)la[i] * la[i]; }
        printf("{\"mode\":\"perturb\",\"backend\":\"%s\",\"relative_perturbation\":1e-6,\"logits_rel_l2\":%.6e}\n", ggml_backend_name(model.backend), sqrt(num / den));
        return 0;
    }
    if (out == "COMPARE" || out == "RESYNC") {     // node-by-node against the CPU backend, first on the prompt batch then on one decode step
        ggml_backend_t cpu = ggml_backend_cpu_init();
        ggml_backend_cpu_set_n_threads(cpu, n_threads);
        int n_past = 0;
        for (int pass = 0; pass < 2; pass++) {
            std::vector<gpt_vocab::id> toks = pass == 0 ? prompt : std::vector<gpt_vocab::id>{ next_tok() };
            const int N = (int)toks.size();
            struct ggml_cgraph * gf = gpt2_graph(model, n_past, N);
            ggml_gallocr_alloc_graph(allocr, gf);
            ggml_backend_tensor_set(ggml_graph_get_tensor(gf, "embd"), toks.data(), 0, N * sizeof(int32_t));
            for (int i = 0; i < N; ++i) { int32_t v = n_past + i; ggml_backend_tensor_set(ggml_graph_get_tensor(gf, "position"), &v, i * sizeof(int32_t), sizeof(v)); }
            printf("=== pass %d: n_past=%d N=%d ===\n", pass, n_past, N);
            ggml_backend_compare_graph_backend(model.backend, cpu, gf, cmp_cb, out == "RESYNC" ? (void *)1 : NULL);
            n_past += N;
        }
        return 0;
    }
    FILE * f = fopen(out.c_str(), "wb");
    std::vector<float> logits;
    int n_past = 0;
    const int64_t t0 = ggml_time_us();
    if (!gpt2_eval(model, allocr, n_threads, n_past, prompt, logits)) return 1;
    const int64_t t1 = ggml_time_us();
    fwrite(logits.data(), sizeof(float), logits.size(), f);
    n_past += n_prompt;
    for (int i = 0; i < n_decode; i++) {
        std::vector<gpt_vocab::id> one = { next_tok() };
        if (!gpt2_eval(model, allocr, n_threads, n_past, one, logits)) return 1;
        fwrite(logits.data(), sizeof(float), logits.size(), f);
        n_past += 1;
    }
    const int64_t t2 = ggml_time_us();
    fclose(f);
    printf("{\"backend\":\"%s\",\"n_prompt\":%d,\"n_decode\":%d,\"prompt_ms\":%.3f,\"decode_ms_per_token\":%.3f}\n",
           ggml_backend_name(model.backend), n_prompt, n_decode, (t1 - t0) / 1e3, n_decode ? (t2 - t1) / 1e3 / n_decode : 0.0);
    ggml_gallocr_free(allocr);
    return 0;
}
