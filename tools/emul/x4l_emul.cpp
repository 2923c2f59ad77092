// x4l_emul.cpp — runs the SOURCE of k_gemm_q4k_x4l<1> (ggml_amd/csrc/gemm_q_x4l.hip) on the CPU, one OS thread per GPU thread,
// work-group after work-group, and writes Y.  Test infrastructure (tests/test_build_static.py builds and runs it):
//   x4l_emul M K B w.bin xh.bin y.bin [splitk [form [type]]]      w = Q4_K (12) or Q5_K (13) rows, xh = the kernel's fp16 activation image, y = fp32 [B][M] (output)
// A mismatch in barrier counts between loader and compute waves shows up as a hang (the test has a timeout), wrong
// indexing as wrong numbers.  Asynchrony is not modelled (see hip_emul.h).
#include "hip_emul.h"
#include <signal.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>
#include <system_error>
#include <thread>
#include <vector>

#define X4L_HW_OVERRIDE
#define X4L_LDS_BASE(smem_) 0u
#define X4L_DMA16(voff, sbase, lds_addr) emu::vm_issue(smem + (lds_addr) + 16 * lane, (sbase) + (voff))
#define X4L_GLOAD16_SYNC(dst, voff, sbase) (memcpy(&(dst), (sbase) + (voff), 16), emu::vm_wait(0))
#define X4L_GLOAD16x2_SYNC(d0, d1, v0, v1, sbase) (memcpy(&(d0), (sbase) + (v0), 16), memcpy(&(d1), (sbase) + (v1), 16), emu::vm_wait(0))
#define X4L_WAIT_VM(n) emu::vm_wait(n)
#define X4L_WAIT_LGKM0() ((void)0)

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx;
dim3 g_gridDim, g_blockDim;
pthread_barrier_t g_wg_barrier;
WaveState *g_waves;
thread_local std::vector<Pending> t_vmq;
bool g_defer_dma = getenv("EMU_DEFER_DMA") && atoi(getenv("EMU_DEFER_DMA")) != 0;
size_t g_weaken = getenv("EMU_WEAKEN_WAITS") ? (size_t)atoi(getenv("EMU_WEAKEN_WAITS")) : 0;
}
// host-side symbols the launcher in the kernel file refers to
int cdna4_set_error_msg(const char *m) { fprintf(stderr, "error: %s\n", m); return -1; }
int cdna4_set_error(hipError_t, const char *, int) { return -1; }
// memory every work-group must see (the kernel's global buffers): anonymous shared mappings, zero-filled, inherited by the
// forked work-group processes
// global buffers: shared between the work-group processes, and each ends right in front of an inaccessible page — an access
// past the end of W, the activation image, Y or the exchange scratch kills the work-group process (reported as a failure)
static void *shared_alloc(size_t n) {
    const size_t pg = 4096, body = (n + pg - 1) / pg * pg;
    char *p = (char *)mmap(nullptr, body + 2 * pg, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(2); }
    mprotect(p, pg, PROT_NONE); mprotect(p + pg + body, pg, PROT_NONE);      // one in front as well (tight when n is a page multiple)
    return p + pg + ((body - n) & ~(size_t)15);       // 16-byte aligned, at most 15 bytes of slack before the rear guard page
}
void *cdna4_gemm_scratch(size_t n, int) { return shared_alloc(n); }
unsigned cdna4_gemm_next_epoch() { return 1; }
int cdna4_gemm_cu_count() { return 256; }
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) emu_launch([&](void) { kernel(__VA_ARGS__); }, grid, block)
template <typename F> static void emu_launch(F body, dim3 grid, dim3 block) {
    emu::g_gridDim = grid; emu::g_blockDim = block;
    const int nthreads = (int)block.x, nwaves = nthreads / 64;
    // one PROCESS per work-group, all at once: each has its own copy of the kernel's `static` (= __shared__) variables, they
    // share the global buffers (MAP_SHARED), and the work-groups of a split-K tile are co-resident as the exchange requires
    // EMU_BLOCKS=lo:hi runs only work-groups lo..hi-1 (full-size problems: a few work-groups of a big grid)
    unsigned b_lo = 0, b_hi = ~0u;
    if (const char *e = getenv("EMU_BLOCKS")) sscanf(e, "%u:%u", &b_lo, &b_hi);
    std::vector<pid_t> kids;
    for (unsigned b = 0; b < grid.x; b++) {
        if (b < b_lo || b >= b_hi) continue;
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); exit(77); }                 // 77: the environment cannot host the emulation (callers skip)
        if (pid > 0) { kids.push_back(pid); continue; }
        prctl(PR_SET_PDEATHSIG, SIGKILL);                      // never outlive the harness
        pthread_barrier_init(&emu::g_wg_barrier, nullptr, nthreads);
        std::vector<emu::WaveState> waves(nwaves);
        for (auto &w : waves) pthread_barrier_init(&w.bar, nullptr, 64);
        emu::g_waves = waves.data();
        std::vector<std::thread> th;
        try {
            for (int t = 0; t < nthreads; t++)
                th.emplace_back([&, t, b] { emu::t_threadIdx = dim3(t); emu::t_blockIdx = dim3(b); body(); });
        } catch (const std::system_error &) { _exit(77); }        // thread limit of the environment
        for (auto &t : th) t.join();
        _exit(0);
    }
    bool cannot = false, failed = false;
    for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (WIFEXITED(st) && WEXITSTATUS(st) == 77) cannot = true; else if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { failed = true; if (WIFSIGNALED(st)) fprintf(stderr, "work-group process killed by signal %d (11 = an access outside the buffers)\n", WTERMSIG(st)); } }
    if (failed) { fprintf(stderr, "work-group process failed\n"); exit(3); }
    if (cannot) { fprintf(stderr, "the environment cannot host the emulation (process / thread limits)\n"); exit(77); }
}

#include "../../ggml_amd/csrc/gemm_q_x4l.hip"

static std::vector<uint8_t> slurp(const char *p) {
    FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> v((size_t)n); if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(2); fclose(f); return v;
}
int main(int argc, char **argv) {
    if (argc < 7) { fprintf(stderr, "usage: x4l_emul M K B w.bin xh.bin y.bin\n"); return 2; }
    const int M = atoi(argv[1]), K = atoi(argv[2]), B = atoi(argv[3]);
    const int splitk = argc > 7 ? atoi(argv[7]) : 1;
    std::vector<uint8_t> w0 = slurp(argv[4]), xh0 = slurp(argv[5]);
    uint8_t *w = (uint8_t *)shared_alloc(w0.size()), *xh = (uint8_t *)shared_alloc(xh0.size());
    memcpy(w, w0.data(), w0.size()); memcpy(xh, xh0.data(), xh0.size());
    float *y = (float *)shared_alloc((size_t)B * M * 4);
    for (size_t i = 0; i < (size_t)B * M; i++) y[i] = -12345.f;
    cdna4_gemm_args a{};
    const int type = argc > 9 ? atoi(argv[9]) : CDNA4_Q4_K;
    a.type = type; a.W = w; a.w_row_bytes = (int64_t)(K / 256) * (type == CDNA4_Q5_K ? 176 : 144); a.xh = xh; a.xh_row_elems = K;
    a.Y = y; a.y_row_elems = M; a.M = M; a.K = K; a.B = B; a.variant = 8199; a.splitk = splitk;
    if (((uintptr_t)a.W & 15) != 0) { fprintf(stderr, "unaligned W\n"); return 2; }
    if (cdna4_launch_gemm_q4k_x4l(a, splitk, argc > 8 ? atoi(argv[8]) : 0, nullptr) != 0) return 1;
    FILE *f = fopen(argv[6], "wb"); fwrite(y, 4, (size_t)B * M, f); fclose(f);
    return 0;
}
