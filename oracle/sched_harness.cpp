// oracle/sched_harness.cpp — TEST INFRASTRUCTURE ONLY (never part of the product).
// Runs the reference's OWN examples/gpt-2/main-sched.cpp — model loader with per-layer backend assignment, graph builder,
// ggml_backend_sched evaluation (included verbatim below, unmodified) — with the MI355X plug-in in the GPU slot.
// main-sched.cpp picks its GPU backend at compile time (`#ifdef GGML_USE_CUDA ... ggml_backend_cuda_init(0)`,
// main-sched.cpp:115-123); this file is compiled with -DGGML_USE_CUDA and DEFINES ggml_backend_cuda_init to hand out the
// plug-in's backend (loaded through ggml_backend_load, i.e. the registry path of src/ggml-backend-reg.cpp:393), so not a
// line of the reference source changes.  A fixed token sequence replaces sampling so that every run sees identical inputs.
//
//   sched_harness <model.bin> <n_gpu_layers> <plugin.so> <out.bin> <n_prompt> <n_decode> <threads> [parallel]
//   n_gpu_layers = 0: reference CPU backend only; parallel = 1: ggml_backend_sched_new(..., parallel = true) (4 input copies +
//   events: exercises event_record / event_wait / event_synchronize of the plug-in)
#define main gpt2_sched_reference_main
#include "examples/gpt-2/main-sched.cpp"
#undef main

static std::string g_plugin;
extern "C" ggml_backend_t ggml_backend_cuda_init(int device) {
    static bool loaded = false;
    if (!loaded) { if (!ggml_backend_load(g_plugin.c_str())) { fprintf(stderr, "failed to load %s\n", g_plugin.c_str()); return NULL; } loaded = true; }
    const std::string name = "CDNA4" + std::to_string(device);
    return ggml_backend_init_by_name(name.c_str(), NULL);
}

int main(int argc, char ** argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s model n_gpu_layers plugin out n_prompt n_decode threads [parallel]\n", argv[0]); return 2; }
    gpt_params params;
    params.model = argv[1]; params.n_gpu_layers = atoi(argv[2]); g_plugin = argv[3];
    const std::string out = argv[4];
    const int n_prompt = atoi(argv[5]), n_decode = atoi(argv[6]);
    params.n_threads = atoi(argv[7]);
    const bool parallel = argc > 8 && atoi(argv[8]) != 0;
    params.n_batch = std::max(n_prompt, 8);
    ggml_time_init();

    gpt_vocab vocab;
    gpt2_model model;
    if (!gpt2_model_load(params.model, model, vocab, params)) { fprintf(stderr, "failed to load model\n"); return 1; }
    // the scheduler, exactly as main-sched.cpp:945-957 (parallel is the one knob added)
    ggml_backend_sched_t sched = ggml_backend_sched_new(model.backends.data(), NULL, model.backends.size(), GPT2_MAX_NODES, parallel);
    {
        int n_tokens = std::min(model.hparams.n_ctx, params.n_batch);
        int n_past = model.hparams.n_ctx - n_tokens;
        ggml_backend_sched_reserve(sched, gpt2_graph(model, n_past, std::vector<gpt_vocab::id>(n_tokens, 0)));
    }
    uint32_t st = 12345;                                            // the same LCG as gpt2_harness.cpp
    auto next_tok = [&]() { st = st * 1664525u + 1013904223u; return (gpt_vocab::id)((st >> 8) % (uint32_t)model.hparams.n_vocab); };
    std::vector<gpt_vocab::id> prompt(n_prompt);
    for (auto & t : prompt) t = next_tok();

    FILE * f = fopen(out.c_str(), "wb");
    std::vector<float> logits;
    int n_past = 0;
    const int64_t t0 = ggml_time_us();
    if (!gpt2_eval(model, sched, n_past, prompt, logits)) return 1;
    const int64_t t1 = ggml_time_us();
    fwrite(logits.data(), sizeof(float), logits.size(), f);
    n_past += n_prompt;
    for (int i = 0; i < n_decode; i++) {
        std::vector<gpt_vocab::id> one = { next_tok() };
        if (!gpt2_eval(model, sched, n_past, one, logits)) return 1;
        fwrite(logits.data(), sizeof(float), logits.size(), f);
        n_past += 1;
    }
    const int64_t t2 = ggml_time_us();
    fclose(f);
    std::string names;
    for (auto b : model.backends) { if (!names.empty()) names += "+"; names += ggml_backend_name(b); }
    printf("{\"backends\":\"%s\",\"n_gpu_layers\":%d,\"parallel\":%d,\"n_splits\":%d,\"n_prompt\":%d,\"n_decode\":%d,\"prompt_ms\":%.3f,\"decode_ms_per_token\":%.3f}\n",
           names.c_str(), params.n_gpu_layers, (int)parallel, ggml_backend_sched_get_n_splits(sched), n_prompt, n_decode, (t1 - t0) / 1e3, n_decode ? (t2 - t1) / 1e3 / n_decode : 0.0);
    ggml_backend_sched_free(sched);
    return 0;
}
