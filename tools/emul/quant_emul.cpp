// quant_emul.cpp — runs the SOURCE of the activation quantizers (ggml_amd/csrc/quantize_act.hip: k_quantize_q8_K, k_quantize_q8_0) on
// the CPU.  Test infrastructure.
//   quant_emul kind K B x.bin qs.bin d.bin bsums.bin xh.bin      kind: 0 = Q8_K, 1 = Q8_0 (AVX2 rounding), 2 = Q8_0 (_ref rounding), 3 = Q8_1
// Outputs: int8 qs[B][K], float d[B][K/QK], int16 bsums[B][K/16] (Q8_K; Q8_1: float s[B][K/32] in the same bytes), and the fp16 activation
// image of the MFMA GEMM.
#include "hip_emul.h"
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <system_error>
#include <thread>
#include <vector>

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
    // many small work-groups: a process runs a contiguous RANGE of them one after the other (no work-group state survives)
    const unsigned nproc = 16, per = (grid.x + nproc - 1) / nproc;
    std::vector<pid_t> kids;
    for (unsigned pi = 0; pi < nproc && pi * per < grid.x; pi++) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); exit(77); }
        if (pid > 0) { kids.push_back(pid); continue; }
        prctl(PR_SET_PDEATHSIG, SIGKILL);
        for (unsigned b = pi * per; b < std::min(grid.x, (pi + 1) * per); b++) {
            pthread_barrier_init(&emu::g_wg_barrier, nullptr, nthreads);
            std::vector<emu::WaveState> waves(nthreads / 64);
            for (auto &w : waves) pthread_barrier_init(&w.bar, nullptr, 64);
            emu::g_waves = waves.data();
            std::vector<std::thread> th;
            try {
                for (int t = 0; t < nthreads; t++) th.emplace_back([&, t, b] { emu::t_threadIdx = dim3(t); emu::t_blockIdx = dim3(b); body(); });
            } catch (const std::system_error &) { _exit(77); }
            for (auto &t : th) t.join();
        }
        _exit(0);
    }
    bool cannot = false, failed = false;
    for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (WIFEXITED(st) && WEXITSTATUS(st) == 77) cannot = true; else if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { failed = true; if (WIFSIGNALED(st)) fprintf(stderr, "work-group process killed by signal %d\n", WTERMSIG(st)); } }
    if (failed) { fprintf(stderr, "work-group process failed\n"); exit(3); }
    if (cannot) { fprintf(stderr, "the environment cannot host the emulation\n"); exit(77); }
}

#include "../../ggml_amd/csrc/quantize_act.hip"

static std::vector<uint8_t> slurp(const char *p) {
    FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> v((size_t)n); if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(2); fclose(f); return v;
}
static void dump(const char *p, const void *d, size_t n) { FILE *f = fopen(p, "wb"); fwrite(d, 1, n, f); fclose(f); }
int main(int argc, char **argv) {
    if (argc < 9) { fprintf(stderr, "usage: quant_emul kind K B x.bin qs.bin d.bin bsums.bin xh.bin\n"); return 2; }
    const int kind = atoi(argv[1]); const int64_t K = atoll(argv[2]), B = atoll(argv[3]);
    std::vector<uint8_t> x0 = slurp(argv[4]);
    float *x = (float *)shared_alloc(x0.size()); memcpy(x, x0.data(), x0.size());
    const int64_t qk = kind == 0 ? 256 : 32;
    int8_t *qs = (int8_t *)shared_alloc((size_t)(B * K)); float *d = (float *)shared_alloc((size_t)(B * (K / qk)) * 4);
    int16_t *bs = (int16_t *)shared_alloc((size_t)(B * (K / 16)) * 2); uint8_t *xh = (uint8_t *)shared_alloc((size_t)(B * K) * 2);
    int rc = kind == 0 ? cdna4_launch_quantize_q8_K(x, K, K, B, qs, d, bs, xh, nullptr) :
             kind == 3 ? cdna4_launch_quantize_q8_1(x, K, K, B, qs, d, reinterpret_cast<float *>(bs), xh, false, nullptr) : cdna4_launch_quantize_q8_0(x, K, K, B, qs, d, xh, kind == 2, nullptr);
    if (rc != 0) return 1;
    dump(argv[5], qs, (size_t)(B * K)); dump(argv[6], d, (size_t)(B * (K / qk)) * 4); dump(argv[7], bs, (size_t)(B * (K / 16)) * 2); dump(argv[8], xh, (size_t)(B * K) * 2);
    return 0;
}
