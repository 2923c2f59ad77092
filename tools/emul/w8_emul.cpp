// w8_emul.cpp — runs the SOURCE of k_gemm_kq_w8 / k_gemm_kq_w8p (ggml_amd/csrc/gemm_kq_w8.inc + gemm_w8_epilogue.inc: the 8-wave
// kernels — Q5_K's default, the shallow-K fallback, the A/B baselines and their TRACE builds) on the CPU like w12_emul.cpp.
// Test infrastructure.
//   w8_emul M K B w.bin xh.bin y.bin splitk kernel xchg_l2 [type]     kernel: 20 / 12 / 0 = k_gemm_kq_w8 OPT, 64 = w8p, +1000 = TRACE build
#include "hip_emul.h"
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <system_error>
#include <thread>
#include <vector>

#define CDNA4_HW_OVERRIDE
#define CDNA4_LDS_BASE(smem_) 0u
#define CDNA4_DMA16(voff, sbase, lds_addr) emu::vm_issue(smem + (lds_addr) + 16 * lane, (sbase) + (voff))
#define CDNA4_GLOAD16_PTR(dst, ptr) (memcpy(&(dst), (ptr), 16), emu::vm_issue_done())
#define CDNA4_WAIT_VM_TIED1(n, a) emu::vm_wait(n)
#define CDNA4_WAIT_VM_TIED2(n, a, b) emu::vm_wait(n)
#define CDNA4_WAIT_VM(n) emu::vm_wait(n)
#define cdna4_wait_vm_rt(n) emu::vm_wait(n)
#define CDNA4_WAIT_LGKM0() ((void)0)

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx;
dim3 g_gridDim, g_blockDim;
pthread_barrier_t g_wg_barrier;
WaveState *g_waves;
thread_local std::vector<Pending> t_vmq;
bool g_defer_dma = getenv("EMU_DEFER_DMA") && atoi(getenv("EMU_DEFER_DMA")) != 0;
size_t g_weaken = getenv("EMU_WEAKEN_WAITS") ? (size_t)atoi(getenv("EMU_WEAKEN_WAITS")) : 0;
}
int cdna4_set_error_msg(const char *m) { fprintf(stderr, "error: %s\n", m); return -1; }
// global buffers: shared between the work-group processes, and each ends right in front of an inaccessible page — an access
// past the end of W, the activation image, Y or the exchange scratch kills the work-group process (reported as a failure)
static void *shared_alloc(size_t n) {
    const size_t pg = 4096, body = (n + pg - 1) / pg * pg;
    char *p = (char *)mmap(nullptr, body + 2 * pg, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(2); }
    mprotect(p, pg, PROT_NONE); mprotect(p + pg + body, pg, PROT_NONE);      // one in front as well (tight when n is a page multiple)
    return p + pg + ((body - n) & ~(size_t)15);       // 16-byte aligned, at most 15 bytes of slack before the rear guard page
}

#include "../../ggml_amd/csrc/gemm_q_common.h"
#include "../../ggml_amd/csrc/gemm_q_hw.h"
void *cdna4_debug_trace = nullptr;
#include "../../ggml_amd/csrc/gemm_kq_w8.inc"

template <typename F> static void emu_launch(F body, unsigned nblk, int nthreads) {
    emu::g_gridDim = dim3(nblk); emu::g_blockDim = dim3(nthreads);
    // EMU_BLOCKS=lo:hi runs only work-groups lo..hi-1 (full-size problems: a few work-groups of a big grid)
    unsigned b_lo = 0, b_hi = ~0u;
    if (const char *e = getenv("EMU_BLOCKS")) sscanf(e, "%u:%u", &b_lo, &b_hi);
    std::vector<pid_t> kids;
    for (unsigned b = 0; b < nblk; b++) {
        if (b < b_lo || b >= b_hi) continue;
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); exit(77); }                 // 77: the environment cannot host the emulation (callers skip)
        if (pid > 0) { kids.push_back(pid); continue; }
        prctl(PR_SET_PDEATHSIG, SIGKILL);                      // never outlive the harness
        pthread_barrier_init(&emu::g_wg_barrier, nullptr, nthreads);
        std::vector<emu::WaveState> waves(nthreads / 64);
        for (auto &w : waves) pthread_barrier_init(&w.bar, nullptr, 64);
        emu::g_waves = waves.data();
        std::vector<std::thread> th;
        try {
            for (int t = 0; t < nthreads; t++) th.emplace_back([&, t, b] { emu::t_threadIdx = dim3(t); emu::t_blockIdx = dim3(b); body(); });
        } catch (const std::system_error &) { _exit(77); }        // thread limit of the environment
        for (auto &t : th) t.join();
        _exit(0);
    }
    bool cannot = false, failed = false;
    for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (WIFEXITED(st) && WEXITSTATUS(st) == 77) cannot = true; else if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { failed = true; if (WIFSIGNALED(st)) fprintf(stderr, "work-group process killed by signal %d (11 = an access outside the buffers)\n", WTERMSIG(st)); } }
    if (failed) { fprintf(stderr, "work-group process failed\n"); exit(3); }
    if (cannot) { fprintf(stderr, "the environment cannot host the emulation (process / thread limits)\n"); exit(77); }
}

static std::vector<uint8_t> slurp(const char *p) {
    FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> v((size_t)n); if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(2); fclose(f); return v;
}
int main(int argc, char **argv) {
    if (argc < 10) { fprintf(stderr, "usage: w8_emul M K B w.bin xh.bin y.bin splitk kernel xchg_l2 [type]\n"); return 2; }
    const int M = atoi(argv[1]), K = atoi(argv[2]), B = atoi(argv[3]), splitk = atoi(argv[7]), exp = atoi(argv[8]), l2 = atoi(argv[9]);
    std::vector<uint8_t> w0 = slurp(argv[4]), xh0 = slurp(argv[5]);
    uint8_t *w = (uint8_t *)shared_alloc(w0.size()), *xh = (uint8_t *)shared_alloc(xh0.size());
    memcpy(w, w0.data(), w0.size()); memcpy(xh, xh0.data(), xh0.size());
    float *y = (float *)shared_alloc((size_t)B * M * 4);
    for (size_t i = 0; i < (size_t)B * M; i++) y[i] = -12345.f;
    // the parameter block exactly as launch_w8() (gemm_q_mfma.hip) fills it for this kernel
    gemm_params p{};
    const int type = argc > 10 ? atoi(argv[10]) : CDNA4_Q4_K;
    p.W = w; p.w_row_bytes = (int64_t)(K / 256) * (type == CDNA4_Q5_K ? 176 : 144);
    p.trace = (unsigned long long *)shared_alloc(65536);             // the 64-KiB buffer ggml_cdna4_debug_trace() is given (TRACE builds)
    p.xh = (const half_t *)xh; p.xh_row = K; p.Y = y; p.y_row = M; p.M = M; p.K = K; p.B = B; p.splitk = splitk;
    p.tiles_m = (M + 127) / 128; p.tiles_b = (B + 127) / 128;
    const int ntiles = p.tiles_m * p.tiles_b, total = K / 256;
    if (splitk == 2) {
        const size_t pbytes = (size_t)ntiles * 128 * 128 * 4, fbytes = 65536;
        char *sc = (char *)shared_alloc(fbytes + pbytes);
        memset(sc, 0, fbytes);
        p.flags = (unsigned *)sc; p.partial = (float *)(sc + fbytes);
        int split = (total * 8 + 8) / 16;
        p.sb_split = split < 1 ? 1 : (split > total - 1 ? total - 1 : split);
        p.xchg_l2 = l2;
    } else if (splitk != 1) { fprintf(stderr, "splitk 1 or 2\n"); return 2; }
    const int min_nsb = p.partial ? (p.sb_split < total - p.sb_split ? p.sb_split : total - p.sb_split) : total / splitk;
    const unsigned nblk = (unsigned)(ntiles * splitk);
    const int kern = exp;
    if ((kern % 1000) == 64 && min_nsb < 3) { fprintf(stderr, "k_gemm_kq_w8p needs 3 superblocks of K per work-group\n"); return 2; }
#define RUN(...) emu_launch([&] { __VA_ARGS__(p); }, nblk, 512)
    if (type == CDNA4_Q4_K) switch (kern) {
        case 20: RUN(k_gemm_kq_w8<CDNA4_Q4_K, false, 20>); break;
        case 12: RUN(k_gemm_kq_w8<CDNA4_Q4_K, false, 12>); break;
        case 0: RUN(k_gemm_kq_w8<CDNA4_Q4_K, false, 0>); break;
        case 1000: RUN(k_gemm_kq_w8<CDNA4_Q4_K, true, 0>); break;
        case 1020: RUN(k_gemm_kq_w8<CDNA4_Q4_K, true, 20>); break;
        case 64: RUN(k_gemm_kq_w8p<CDNA4_Q4_K, false>); break;
        case 1064: RUN(k_gemm_kq_w8p<CDNA4_Q4_K, true>); break;
        default: fprintf(stderr, "kernel not built into the emulator\n"); return 2;
    } else if (type == CDNA4_Q5_K) switch (kern) {
        case 20: RUN(k_gemm_kq_w8<CDNA4_Q5_K, false, 20>); break;
        case 64: RUN(k_gemm_kq_w8p<CDNA4_Q5_K, false>); break;
        default: fprintf(stderr, "kernel not built into the emulator\n"); return 2;
    } else { fprintf(stderr, "type not built into the emulator\n"); return 2; }
    FILE *f = fopen(argv[6], "wb"); fwrite(y, 4, (size_t)B * M, f); fclose(f);
    return 0;
}
