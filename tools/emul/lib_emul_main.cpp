// lib_emul_main.cpp — driver of the whole-library emulation build (lib_emul.h): calls the library's C-ABI like a host would.  Test infrastructure.
//   lib_emul mul_mat type M K B path w.bin x.bin y.bin          ggml_cdna4_mul_mat (path 0 = auto, 1 = GEMV, 2 = GEMM); w = M contiguous rows, x = [B][K] f32, y = [B][M]
//   lib_emul mul_mat_id type M K n_expert n_used n_b n_tok w.bin x.bin ids.bin y.bin
#include "lib_emul.h"
#include "../../include/ggml_cdna4.h"

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx;
dim3 g_gridDim, g_blockDim;
pthread_barrier_t g_wg_barrier;
WaveState *g_waves;
thread_local std::vector<Pending> t_vmq;
bool g_defer_dma = getenv("EMU_DEFER_DMA") && atoi(getenv("EMU_DEFER_DMA")) != 0;
size_t g_weaken = getenv("EMU_WEAKEN_WAITS") ? (size_t)atoi(getenv("EMU_WEAKEN_WAITS")) : 0;      // self-test of the counted waits: every wait tolerates this many more outstanding operations
}
__attribute__((aligned(16))) uint8_t smem[160 * 1024];             // the dynamic LDS of gemv_q.hip's kernels (a work-group is a process)
void *emu_shared_alloc(size_t n) {                                   // between two inaccessible pages, shared with the work-group processes
    const size_t pg = 4096, body = (n + pg - 1) / pg * pg;
    char *p = (char *)mmap(nullptr, body + 2 * pg, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(2); }
    mprotect(p, pg, PROT_NONE); mprotect(p + pg + body, pg, PROT_NONE);
    return p + pg + ((body - n) & ~(size_t)255);                     // 256-byte aligned like hipMalloc; the rear guard page at most 255 bytes away
}
static void *load(const char *path, size_t *n_out = nullptr) {
    FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    void *p = emu_shared_alloc((size_t)n + 16); if (fread(p, 1, (size_t)n, f) != (size_t)n) exit(2); fclose(f);
    if (n_out) *n_out = (size_t)n;
    return p;
}
static void store(const char *path, const void *p, size_t n) { FILE *f = fopen(path, "wb"); fwrite(p, 1, n, f); fclose(f); }

int main(int argc, char **argv) {
    if (argc >= 10 && !strcmp(argv[1], "mul_mat")) {
        const int type = atoi(argv[2]); const int64_t M = atoll(argv[3]), K = atoll(argv[4]), B = atoll(argv[5]); const int path = atoi(argv[6]);
        size_t wn = 0; void *w = load(argv[7], &wn); float *x = (float *)load(argv[8]);
        float *y = (float *)emu_shared_alloc((size_t)(M * B) * 4);
        for (int64_t i = 0; i < M * B; i++) y[i] = -12345.f;
        const size_t wsn = ggml_cdna4_mul_mat_workspace_size(type, K, B);
        void *ws = emu_shared_alloc(wsn ? wsn : 256);
        // EMU_RESIDENT=1: with a resident kernel-native image of the weights registered first (built twice and compared, like a host does at load) where the type has one
        if (getenv("EMU_RESIDENT") && atoi(getenv("EMU_RESIDENT")) != 0) {
            const size_t in = ggml_cdna4_resident_image_size(type, M, K);
            if (in) {
                void *img = emu_shared_alloc(in);
                if (ggml_cdna4_resident_image_register(type, w, (int64_t)(wn / (size_t)M), M, K, img, 1, nullptr)) { fprintf(stderr, "resident_image_register: %s\n", ggml_cdna4_last_error()); return 1; }
                const int route = ggml_cdna4_mul_mat_route_of(type, w, (int64_t)(wn / (size_t)M), M, K, B);
                fprintf(stderr, "resident image registered: route %d\n", route);
                // EMU_DEFER_DMA=2: defer only on the kernels whose every wait is a counted vmcnt in the source (k_gemm_kq_t64 = 10, k_gemm_r8 = 12).  The staged forms of
                // k_gemm_kq_w12 rely on the compiler's own wait for their loader waves' register loads (everything older — the activation pieces — has landed with them):
                // the emulation completes plain loads at once and has no such wait to retire the deferred copies
                if (getenv("EMU_DEFER_DMA") && atoi(getenv("EMU_DEFER_DMA")) == 2 && route != 10 && route != 12) emu::g_defer_dma = false;
            }
        }
        else if (getenv("EMU_DEFER_DMA") && atoi(getenv("EMU_DEFER_DMA")) == 2) emu::g_defer_dma = false;
        const int rc = ggml_cdna4_mul_mat(type, w, (int64_t)(wn / (size_t)M), x, K, y, M, M, K, B, ws, wsn, path, 0, 0, nullptr);
        if (rc) { fprintf(stderr, "mul_mat: %s\n", ggml_cdna4_last_error()); return 1; }
        store(argv[9], y, (size_t)(M * B) * 4);
        return 0;
    }
    if (argc >= 13 && !strcmp(argv[1], "mul_mat_id")) {
        const int type = atoi(argv[2]); const int64_t M = atoll(argv[3]), K = atoll(argv[4]), NE = atoll(argv[5]), NU = atoll(argv[6]), NB = atoll(argv[7]), NT = atoll(argv[8]);
        size_t wn = 0; void *w = load(argv[9], &wn); float *x = (float *)load(argv[10]); int32_t *ids = (int32_t *)load(argv[11]);
        float *y = (float *)emu_shared_alloc((size_t)(M * NU * NT) * 4);
        for (int64_t i = 0; i < M * NU * NT; i++) y[i] = -12345.f;
        const size_t wsn = ggml_cdna4_mul_mat_id_workspace_size(type, K, NE, NU, NB, NT);
        void *ws = emu_shared_alloc(wsn ? wsn : 256);
        const int64_t rb = (int64_t)(wn / (size_t)(M * NE));
        if (getenv("EMU_RESIDENT") && atoi(getenv("EMU_RESIDENT")) != 0) {      // a resident image of the whole expert stack (M * NE rows)
            const size_t in = ggml_cdna4_resident_image_size(type, M * NE, K);
            if (in) {
                void *img = emu_shared_alloc(in);
                if (ggml_cdna4_resident_image_register(type, w, rb, M * NE, K, img, 1, nullptr)) { fprintf(stderr, "resident_image_register: %s\n", ggml_cdna4_last_error()); return 1; }
            }
        }
        const int rc = ggml_cdna4_mul_mat_id(type, w, rb, rb * M, x, K, NB * K, ids, NU, y, M, NU * M, M, K, NE, NU, NB, NT, ws, wsn, nullptr);
        if (rc) { fprintf(stderr, "mul_mat_id: %s\n", ggml_cdna4_last_error()); return 1; }
        store(argv[12], y, (size_t)(M * NU * NT) * 4);
        return 0;
    }
    fprintf(stderr, "usage: lib_emul mul_mat type M K B path w.bin x.bin y.bin | lib_emul mul_mat_id type M K n_expert n_used n_b n_tok w.bin x.bin ids.bin y.bin\n");
    return 2;
}
