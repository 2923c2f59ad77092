// lds_emul.cpp — runs the SOURCE of k_gemm_r8 (ggml_amd/csrc/gemm_r8.inc: 32 x 256 wave tiles, in-register unpack, 256 x 256 work-group tiles; the harness of the
// dequantize-into-LDS kernels of round 4, which were removed in round 5) on the CPU like t64_emul.cpp.  Test infrastructure.
//   lds_emul M K B w.bin xh.bin y.bin splitk tm xchg_l2 [type]
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
#define CDNA4_WAIT_LGKM0_VISIBLE() ((void)0)
#define CDNA4_PIN(x) ((void)0)
#define CDNA4_DMA16_LANES(voff, sbase, lds_addr, nlanes) do { if (lane < (nlanes)) emu::vm_issue(smem + (lds_addr) + 16 * lane, (sbase) + (voff)); else emu::vm_issue_done(); } while (0)
// v_permlane32_swap through the wave's exchange buffer (all 64 lanes execute it)
#define CDNA4_SWAP32(a, b) do { emu::WaveState &w_ = emu::my_wave(); const int l_ = emu::t_threadIdx.x & 63; const uint32_t a_ = (a), b_ = (b); \
    w_.xch[l_] = l_ < 32 ? b_ : a_; pthread_barrier_wait(&w_.bar); const uint32_t o_ = w_.xch[l_ ^ 32]; pthread_barrier_wait(&w_.bar); \
    if (l_ < 32) (b) = o_; else (a) = o_; } while (0)

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
#include "../../ggml_amd/csrc/gemm_r8.inc"

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
    if (argc < 10) { fprintf(stderr, "usage: lds_emul M K B w.bin xh.bin y.bin splitk tm xchg_l2 [type]\n"); return 2; }
    const int M = atoi(argv[1]), K = atoi(argv[2]), B = atoi(argv[3]), splitk = atoi(argv[7]), tm = atoi(argv[8]);
    const int form = atoi(argv[9]) == 2 ? 1 : atoi(argv[9]) == 3 ? 2 : 0;   // (the xchg_l2 argument of the other harnesses: 2 selects k_gemm_w4, 3 k_gemm_r8)
    std::vector<uint8_t> w0 = slurp(argv[4]), xh0 = slurp(argv[5]);
    uint8_t *w = (uint8_t *)shared_alloc(w0.size()), *xh = (uint8_t *)shared_alloc(xh0.size());      // no slack: rows past B are clamped by the kernel
    memcpy(w, w0.data(), w0.size()); memcpy(xh, xh0.data(), xh0.size());
    float *y = (float *)shared_alloc((size_t)B * M * 4);
    for (size_t i = 0; i < (size_t)B * M; i++) y[i] = -12345.f;
    // the parameter block exactly as cdna4_launch_gemm_lds() (gemm_q_lds.hip) fills it
    gemm_params p{};
    const int wtype = argc > 10 ? atoi(argv[10]) : CDNA4_Q4_K;          // 12 Q4_K, 13 Q5_K, 102 Q4_0R, 108 Q8_0R, 115 Q6_K8 (the resident re-layouts: w.bin holds the IMAGE)
    p.W = w; p.w_row_bytes = (int64_t)(K / 256) * (wtype == CDNA4_Q5_K ? 176 : (wtype == CDNA4_Q8_0R ? 272 : (wtype == CDNA4_Q6_K8 ? 288 : 144)));
    p.xh = (const half_t *)xh; p.xh_row = K; p.Y = y; p.y_row = M; p.M = M; p.K = K; p.B = B; p.splitk = splitk;
    if (tm != 128 && tm != 256) { fprintf(stderr, "tm 128 or 256\n"); return 2; }
    p.tiles_m = (M + tm - 1) / tm; p.tiles_b = (B + 255) / 256;
    const int ntiles = p.tiles_m * p.tiles_b, nsb = K / 256;
    const int nfr = form == 2 ? 8 : form ? (tm == 256 ? 16 : 8) : (tm == 256 ? 8 : 4), nwv = form == 1 ? 4 : 8;
    if (splitk < 1 || nfr % splitk || nsb < splitk) { fprintf(stderr, "splitk must divide %d and leave a superblock per work-group\n", nfr); return 2; }
    unsigned *flags = nullptr;
    if (splitk > 1) {                                                   // as cdna4_launch_gemm_lds(): counters, then the exchange slots
        const size_t pbytes = (size_t)ntiles * splitk * splitk * nwv * (nfr / splitk) * 4096, fbytes = 65536;
        char *sc = (char *)shared_alloc(fbytes + pbytes);
        memset(sc, 0, fbytes);
        p.flags = flags = (unsigned *)sc; p.partial = (float *)(sc + fbytes);
    }
    const unsigned nblk = (unsigned)(ntiles * splitk);
    if (form != 2) { fprintf(stderr, "k_gemm_lds / k_gemm_w4 were removed in round 5; pass 3 as the ninth argument (k_gemm_r8)\n"); return 2; }
    if (wtype == CDNA4_Q5_K) emu_launch([&] { k_gemm_r8<CDNA4_Q5_K>(p); }, nblk, 512);
    else if (wtype == CDNA4_Q4_0R) emu_launch([&] { k_gemm_r8<CDNA4_Q4_0R>(p); }, nblk, 512);
    else if (wtype == CDNA4_Q8_0R) emu_launch([&] { k_gemm_r8<CDNA4_Q8_0R>(p); }, nblk, 512);
    else if (wtype == CDNA4_Q6_K8) emu_launch([&] { k_gemm_r8<CDNA4_Q6_K8>(p); }, nblk, 512);
    else emu_launch([&] { k_gemm_r8<CDNA4_Q4_K>(p); }, nblk, 512);
    if (flags) for (int i = 0; i < 16384; i++) if (flags[i] != 0) { fprintf(stderr, "split-K counter word %d was not reset by the last work-group to leave (%u)\n", i, flags[i]); return 4; }
    FILE *f = fopen(argv[6], "wb"); fwrite(y, 4, (size_t)B * M, f); fclose(f);
    return 0;
}
