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

// ---- GGUF (BASELINE configs[3] "gpt-2 117M GGUF Q4_0"): the same model as a GGUF file.
//   WRITE: the reference's own writer (gguf_init_empty / gguf_set_val_* / gguf_add_tensor / gguf_write_to_file, src/gguf.cpp:1270-...) on the
//          tensors the reference's .bin loader produced — `gpt2_harness model.bin CPU - TOGGUF:<out.gguf> 0 0 1`.
//   READ:  a model path ending in .gguf is loaded through the PRODUCT's reader (include/ggml_cdna4_gguf.h in libcdna4_kernels.so, dlopen'ed: header,
//          key/value pairs and tensor table parsed from the mapping) and its upload path (ggml_cdna4_gguf_upload: mapping -> pinned staging -> HBM)
//          straight into the plug-in's weight buffer — replacing gpt2_model_load's stdio read + ggml_backend_tensor_set (main-backend.cpp:96-420).
//          The tensors, their names, shapes and the KV cache are created exactly as gpt2_model_load creates them (:236-344), the graph builder and
//          the eval loop are the reference's, unmodified.
#include <dlfcn.h>
#include "gguf.h"
#include <chrono>
struct cdna4_gguf_api {
    void * h = nullptr;
    void * (*open)(const char *, int); void (*close)(void *);
    int64_t (*find_key)(const void *, const char *); int (*val)(const void *, int64_t, int, void *);
    int64_t (*n_tensors)(const void *); const char * (*tensor_name)(const void *, int64_t); int (*tensor_type)(const void *, int64_t);
    int (*tensor_ne)(const void *, int64_t, int64_t *); size_t (*tensor_size)(const void *, int64_t); const void * (*tensor_data)(const void *, int64_t);
    int (*upload)(const void *, int64_t, void *, size_t, void *); const char * (*last_error)(void);
    bool load(const char * path) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return false; }
#define SYM(f, n) f = (decltype(f))dlsym(h, n); if (!f) { fprintf(stderr, "missing %s\n", n); return false; }
        SYM(open, "ggml_cdna4_gguf_open") SYM(close, "ggml_cdna4_gguf_close") SYM(find_key, "ggml_cdna4_gguf_find_key") SYM(val, "ggml_cdna4_gguf_val")
        SYM(n_tensors, "ggml_cdna4_gguf_n_tensors") SYM(tensor_name, "ggml_cdna4_gguf_tensor_name") SYM(tensor_type, "ggml_cdna4_gguf_tensor_type")
        SYM(tensor_ne, "ggml_cdna4_gguf_tensor_ne") SYM(tensor_size, "ggml_cdna4_gguf_tensor_size") SYM(tensor_data, "ggml_cdna4_gguf_tensor_data")
        SYM(upload, "ggml_cdna4_gguf_upload") SYM(last_error, "ggml_cdna4_last_error")
#undef SYM
        return true;
    }
};
static const char * kHparamKeys[6] = {"gpt2.n_vocab", "gpt2.n_ctx", "gpt2.n_embd", "gpt2.n_head", "gpt2.n_layer", "gpt2.ftype"};

static bool gpt2_write_gguf(const gpt2_model & model, const std::string & path) {
    struct gguf_context * g = gguf_init_empty();
    gguf_set_val_str(g, "general.architecture", "gpt2");
    const int32_t hp[6] = {model.hparams.n_vocab, 1024, model.hparams.n_embd, model.hparams.n_head, model.hparams.n_layer, model.hparams.ftype};
    for (int i = 0; i < 6; i++) gguf_set_val_i32(g, kHparamKeys[i], hp[i]);
    for (const auto & kv : model.tensors) {
        if (kv.first == "model/lm_head") continue;                         // tied to model/wte in every gpt-2 file (main-backend.cpp:415-418)
        ggml_set_name(kv.second, kv.first.c_str());
        gguf_add_tensor(g, kv.second);                                     // host data: the model was loaded on the CPU backend
    }
    const bool ok = gguf_write_to_file(g, path.c_str(), false);
    gguf_free(g);
    return ok;
}

static bool gpt2_model_load_gguf(const std::string & fname, gpt2_model & model, int n_ctx, const char * kernels_so) {
    static cdna4_gguf_api A;
    if (!A.h && !A.load(kernels_so)) return false;
    void * g = A.open(fname.c_str(), 1);
    if (!g) { fprintf(stderr, "gguf open failed: %s\n", A.last_error()); return false; }
    int32_t hp[6];
    for (int i = 0; i < 6; i++) { const int64_t k = A.find_key(g, kHparamKeys[i]); if (k < 0 || A.val(g, k, 5 /* GGUF INT32 */, &hp[i])) { fprintf(stderr, "gguf: missing %s\n", kHparamKeys[i]); return false; } }
    auto & h = model.hparams;
    h.n_vocab = hp[0]; h.n_ctx = hp[1]; h.n_embd = hp[2]; h.n_head = hp[3]; h.n_layer = hp[4]; h.ftype = hp[5];
    const int64_t nt = A.n_tensors(g);
    struct ggml_init_params ip = { ggml_tensor_overhead() * (size_t)(nt + 4), NULL, true };
    struct ggml_context * ctx = model.ctx_w = ggml_init(ip);
    if (!model.backend) model.backend = ggml_backend_cpu_init();
    std::vector<int64_t> ids;
    for (int64_t i = 0; i < nt; i++) {
        int64_t ne[4]; A.tensor_ne(g, i, ne);
        struct ggml_tensor * t = ne[1] == 1 ? ggml_new_tensor_1d(ctx, (enum ggml_type)A.tensor_type(g, i), ne[0]) : ggml_new_tensor_2d(ctx, (enum ggml_type)A.tensor_type(g, i), ne[0], ne[1]);
        ggml_set_name(t, A.tensor_name(g, i));
        model.tensors[A.tensor_name(g, i)] = t; ids.push_back(i);
    }
    auto T = [&](const std::string & n) { auto it = model.tensors.find(n); if (it == model.tensors.end()) { fprintf(stderr, "gguf: tensor %s missing\n", n.c_str()); exit(1); } return it->second; };
    model.ln_f_g = T("model/ln_f/g"); model.ln_f_b = T("model/ln_f/b"); model.wte = T("model/wte"); model.wpe = T("model/wpe"); model.lm_head = model.wte;
    model.layers.resize(h.n_layer);
    for (int i = 0; i < h.n_layer; i++) {
        auto & l = model.layers[i]; const std::string p = "model/h" + std::to_string(i) + "/";
        l.ln_1_g = T(p + "ln_1/g"); l.ln_1_b = T(p + "ln_1/b"); l.ln_2_g = T(p + "ln_2/g"); l.ln_2_b = T(p + "ln_2/b");
        l.c_attn_attn_w = T(p + "attn/c_attn/w"); l.c_attn_attn_b = T(p + "attn/c_attn/b"); l.c_attn_proj_w = T(p + "attn/c_proj/w"); l.c_attn_proj_b = T(p + "attn/c_proj/b");
        l.c_mlp_fc_w = T(p + "mlp/c_fc/w"); l.c_mlp_fc_b = T(p + "mlp/c_fc/b"); l.c_mlp_proj_w = T(p + "mlp/c_proj/w"); l.c_mlp_proj_b = T(p + "mlp/c_proj/b");
    }
    model.buffer_w = ggml_backend_alloc_ctx_tensors(ctx, model.backend);
    model.hparams.n_ctx = n_ctx;
    {   // key + value memory: main-backend.cpp:312-344
        struct ggml_init_params kp = { ggml_tensor_overhead() * 2, NULL, true };
        struct ggml_context * kctx = model.ctx_kv = ggml_init(kp);
        const int n_elements = h.n_embd * h.n_layer * n_ctx;
        model.memory_k = ggml_new_tensor_1d(kctx, GGML_TYPE_F32, n_elements);
        model.memory_v = ggml_new_tensor_1d(kctx, GGML_TYPE_F32, n_elements);
        model.buffer_kv = ggml_backend_alloc_ctx_tensors(kctx, model.backend);
    }
    const bool host = ggml_backend_buffer_is_host(model.buffer_w);
    size_t total = 0; const auto t0 = std::chrono::steady_clock::now();
    for (size_t k = 0; k < ids.size(); k++) {
        struct ggml_tensor * t = model.tensors[A.tensor_name(g, ids[k])];
        const size_t n = A.tensor_size(g, ids[k]);
        if (n != ggml_nbytes(t)) { fprintf(stderr, "gguf: %s has %zu bytes, tensor wants %zu\n", t->name, n, ggml_nbytes(t)); return false; }
        if (host) memcpy(t->data, A.tensor_data(g, ids[k]), n);
        else if (A.upload(g, ids[k], t->data, n, NULL)) { fprintf(stderr, "gguf upload of %s failed: %s\n", t->name, A.last_error()); return false; }   // t->data of the plug-in's buffers IS the device address
        total += n;
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "{\"gguf_load\":\"%s\",\"tensors\":%zu,\"bytes\":%zu,\"seconds\":%.4f,\"GBps\":%.3f,\"path\":\"%s\"}\n", fname.c_str(), ids.size(), total, sec, total / sec / 1e9,
            host ? "mapping -> host buffer (memcpy)" : "mapping -> pinned staging -> HBM (ggml_cdna4_gguf_upload)");
    A.close(g);
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
    const bool is_gguf = fname.size() > 5 && fname.substr(fname.size() - 5) == ".gguf";
    if (is_gguf) {
        const char * ks = getenv("CDNA4_KERNELS_SO");
        if (!ks) { fprintf(stderr, "a .gguf model needs CDNA4_KERNELS_SO=<path to libcdna4_kernels.so> (the product's GGUF reader)\n"); return 2; }
        if (!gpt2_model_load_gguf(fname, model, 1024, ks)) { fprintf(stderr, "failed to load model\n"); return 1; }
    } else if (!gpt2_model_load(fname, model, vocab, 1024, 0)) { fprintf(stderr, "failed to load model\n"); return 1; }
    if (out.rfind("TOGGUF:", 0) == 0) return gpt2_write_gguf(model, out.substr(7)) ? 0 : 1;
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
