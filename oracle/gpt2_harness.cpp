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
