// gemv_emul.cpp — runs the SOURCE of the B = 1 decode kernel k_gemv_q_fused<TYPE, NW, ROWS> (ggml_amd/csrc/gemv_q.hip: activation
// quantizer + int8-dot GEMV in one launch) on the CPU, like the GEMM emulators.  Test infrastructure.
//   gemv_emul type M K w.bin x.bin y.bin      w = M rows of K weights of ggml type `type`, x = K floats, y = M floats (output)
// Shuffles are wave-collective in the emulation: K must be a multiple of 1024 so that the quantizer's lanes fill whole waves.
#include "hip_emul.h"
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <system_error>
#include <thread>
#include <vector>
#undef __shared__
#define __shared__                         // the kernel's only shared array is the dynamic one: `extern uint8_t smem[]` below
__attribute__((aligned(16))) uint8_t smem[160 * 1024];

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx;
dim3 g_gridDim, g_blockDim;
pthread_barrier_t g_wg_barrier;
WaveState *g_waves;
thread_local std::vector<Pending> t_vmq;
bool g_defer_dma = false;
size_t g_weaken = 0;
}
int cdna4_set_error_msg(const char *m) { fprintf(stderr, "error: %s\n", m); return -1; }
int cdna4_set_error(hipError_t, const char *, int) { return -1; }
int cdna4_gemm_cu_count() { const char *e = getenv("EMU_CUS"); return e ? atoi(e) : 256; }      // (gemm_q_mfma.hip in the product: the decode launcher asks it for the 16 x 1 rule)
static void *shared_alloc(size_t n) {
    const size_t pg = 4096, body = (n + pg - 1) / pg * pg;
    char *p = (char *)mmap(nullptr, body + 2 * pg, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(2); }
    mprotect(p, pg, PROT_NONE); mprotect(p + pg + body, pg, PROT_NONE);
    return p + pg + ((body - n) & ~(size_t)15);
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) emu_launch([&](void) { kernel(__VA_ARGS__); }, grid, block)
template <typename F> static void emu_launch(F body, dim3 grid, dim3 block) {
    emu::g_gridDim = grid; emu::g_blockDim = block;
    const int nthreads = (int)block.x;
    std::vector<pid_t> kids;
    bool cannot = false, failed = false;
    auto reap = [&](size_t keep) {
        while (kids.size() > keep) { int st = 0; waitpid(kids.front(), &st, 0); kids.erase(kids.begin());
            if (WIFEXITED(st) && WEXITSTATUS(st) == 77) cannot = true; else if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) failed = true; }
    };
    for (unsigned by = 0; by < grid.y; by++)
        for (unsigned bx = 0; bx < grid.x; bx++) {
            reap(15);                                            // at most 16 work-group processes at a time
            const pid_t pid = fork();
            if (pid < 0) { perror("fork"); exit(77); }
            if (pid > 0) { kids.push_back(pid); continue; }
            prctl(PR_SET_PDEATHSIG, SIGKILL);
            pthread_barrier_init(&emu::g_wg_barrier, nullptr, nthreads);
            std::vector<emu::WaveState> waves(nthreads / 64);
            for (auto &w : waves) pthread_barrier_init(&w.bar, nullptr, 64);
            emu::g_waves = waves.data();
            std::vector<std::thread> th;
            try {
                for (int t = 0; t < nthreads; t++) th.emplace_back([&, t, bx, by] { emu::t_threadIdx = dim3(t); emu::t_blockIdx = dim3(bx, by); body(); });
            } catch (const std::system_error &) { _exit(77); }
            for (auto &t : th) t.join();
            _exit(0);
        }
    reap(0);
    if (failed) { fprintf(stderr, "work-group process failed\n"); exit(3); }
    if (cannot) { fprintf(stderr, "the environment cannot host the emulation\n"); exit(77); }
}

#define CDNA4_HW_OVERRIDE
#define CDNA4_LDS_BASE(smem_) 0u
#define CDNA4_DMA16(voff, sbase, lds_addr) emu::vm_issue(smem + (lds_addr) + 16 * lane, (sbase) + (voff))
#define CDNA4_WAIT_VM(n) emu::vm_wait(n)
#define cdna4_wait_vm_rt(n) emu::vm_wait(n)
#define CDNA4_WAIT_LGKM0() ((void)0)
#define CDNA4_WAIT_LGKM0_VISIBLE() ((void)0)
#define CDNA4_PIN(x) ((void)0)
#include "../../ggml_amd/csrc/gemv_q.hip"

static std::vector<uint8_t> slurp(const char *p) {
    FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> v((size_t)n); if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(2); fclose(f); return v;
}
// gemv_emul type M K w.bin qs.bin y.bin ncol d.bin bsums.bin : the multi-column GEMV (2 <= B <= 8) on pre-quantized activations
static int main_cols(int argc, char **argv) {
    const int type = atoi(argv[1]), M = atoi(argv[2]), K = atoi(argv[3]), ncol = atoi(argv[7]);
    std::vector<uint8_t> w0 = slurp(argv[4]), q0 = slurp(argv[5]), d0 = slurp(argv[8]), b0 = slurp(argv[9]);
    uint8_t *w = (uint8_t *)shared_alloc(w0.size()); int8_t *qs = (int8_t *)shared_alloc(q0.size()); float *d = (float *)shared_alloc(d0.size());
    int16_t *bs = (int16_t *)shared_alloc(b0.size() ? b0.size() : 16); float *y = (float *)shared_alloc((size_t)M * ncol * 4);
    memcpy(w, w0.data(), w0.size()); memcpy(qs, q0.data(), q0.size()); memcpy(d, d0.data(), d0.size()); if (b0.size()) memcpy(bs, b0.data(), b0.size());
    for (int i = 0; i < M * ncol; i++) y[i] = -12345.f;
    cdna4_gemv_args a{};
    a.type = type; a.W = w; a.w_row_bytes = (int64_t)(w0.size() / (size_t)M); a.qs = qs; a.d = d; a.bsums = bs; a.Y = y; a.y_col_stride = M; a.M = M; a.K = K; a.ncol = ncol;
    // EMU_STAGED=1: the form that copies the pre-quantized rows into LDS once per work-group (k_gemv_q_fused<.., NB, PREQ>)
    if ((getenv("EMU_STAGED") ? cdna4_launch_gemv_q_staged(a, nullptr) : cdna4_launch_gemv_q(a, nullptr)) != 0) return 1;
    FILE *f = fopen(argv[6], "wb"); fwrite(y, 4, (size_t)M * ncol, f); fclose(f);
    return 0;
}
int main(int argc, char **argv) {
    if (argc >= 10) return main_cols(argc, argv);
    if (argc < 7) { fprintf(stderr, "usage: gemv_emul type M K w.bin x.bin y.bin [ncol]   |   gemv_emul type M K w.bin qs.bin y.bin ncol d.bin bsums.bin\n"); return 2; }
    const int type = atoi(argv[1]), M = atoi(argv[2]), K = atoi(argv[3]), ncol = argc > 7 ? atoi(argv[7]) : 1;      // ncol rows of x: the one-launch small-batch form
    std::vector<uint8_t> w0 = slurp(argv[4]), x0 = slurp(argv[5]);
    uint8_t *w = (uint8_t *)shared_alloc(w0.size()); float *x = (float *)shared_alloc(x0.size()), *y = (float *)shared_alloc((size_t)M * ncol * 4);
    memcpy(w, w0.data(), w0.size()); memcpy(x, x0.data(), x0.size());
    for (int i = 0; i < M * ncol; i++) y[i] = -12345.f;
    cdna4_gemv_args a{};
    a.type = type; a.W = w; a.w_row_bytes = (int64_t)(w0.size() / (size_t)M); a.Y = y; a.y_col_stride = M; a.M = M; a.K = K; a.ncol = ncol;
    if ((ncol == 1 ? cdna4_launch_gemv_q_fused(a, x, nullptr) : cdna4_launch_gemv_q_fused_n(a, x, K, nullptr)) != 0) return 1;
    FILE *f = fopen(argv[6], "wb"); fwrite(y, 4, (size_t)M * ncol, f); fclose(f);
    return 0;
}
