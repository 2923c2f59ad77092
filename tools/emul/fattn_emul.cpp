// fattn_emul.cpp — runs the SOURCE of the FLASH_ATTN_EXT kernels (ggml_amd/csrc/fattn.hip: k_flash_attn_split / _merge / _wide <64 / 128 / 256>) on the CPU, one OS
// thread per GPU thread, the 32x32x16 fp16 MFMA emulated lane for lane (hip_emul.h).  Test infrastructure.
//   fattn_emul D n_q n_head n_batch n_kv n_head_kv n_batch_kv has_mask mask_rows scale max_bias softcap permuted q.bin k.bin v.bin mask.bin out.bin [kv_type]
// kv_type (default 1 = F16): a ggml block type (2, 3, 6, 7, 8) makes k.bin / v.bin block-quantized rows (the conversion pass in front of the kernels runs too)
// q f32 [n_batch][n_head][n_q][D] (permuted = 1: stored [n_batch][n_q][n_head][D] and described through strides, like the stock test's
// ggml_permute(0, 2, 1, 3) case; likewise k / v), k / v fp16, mask fp16 [mask_rows][n_kv]; out f32 [n_batch][n_q][n_head][D]
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
__attribute__((aligned(16))) uint8_t fa_dyn_lds[160 * 1024];           // the dynamic LDS of k_flash_attn_pipe (a work-group is a process: one array)
int cdna4_set_error_msg(const char *m) { fprintf(stderr, "error: %s\n", m); return -1; }
int cdna4_set_error(hipError_t, const char *, int) { return -1; }
static void *shared_alloc(size_t n) {                                  // between two inaccessible pages: an out-of-bounds access kills the work-group
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
    const unsigned total = grid.x * grid.y * grid.z, nproc = 16, per = (total + nproc - 1) / nproc;
    std::vector<pid_t> kids;
    for (unsigned pi = 0; pi < nproc && pi * per < total; pi++) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); exit(77); }
        if (pid > 0) { kids.push_back(pid); continue; }
        prctl(PR_SET_PDEATHSIG, SIGKILL);
        for (unsigned b = pi * per; b < std::min(total, (pi + 1) * per); b++) {
            const dim3 bi(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
            pthread_barrier_init(&emu::g_wg_barrier, nullptr, nthreads);
            std::vector<emu::WaveState> waves(nthreads / 64);
            for (auto &w : waves) pthread_barrier_init(&w.bar, nullptr, 64);
            emu::g_waves = waves.data();
            std::vector<std::thread> th;
            try {
                for (int t = 0; t < nthreads; t++) th.emplace_back([&, t, bi] { emu::t_threadIdx = dim3(t); emu::t_blockIdx = bi; body(); });
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

// the library's per-device scratch (gemm_q_mfma.hip) and CU count, as far as fattn.hip needs them; EMU_CUS sets the CU count the key split is sized for
void *cdna4_gemm_scratch(size_t bytes, int kind) { static void *p[8] = {}; static size_t n[8] = {}; if (bytes > n[kind & 7]) { p[kind & 7] = shared_alloc(bytes); n[kind & 7] = bytes; } return p[kind & 7]; }    // one area per kind, like the library
int cdna4_gemm_cu_count() { const char *e = getenv("EMU_CUS"); return e ? atoi(e) : 256; }

// a quantized K / V goes through ops.hip's k_q_to_f16_dense first (the same TU here: ops.hip needs these two runtime calls)
#define hipMemcpyDeviceToDevice 0
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
// the GPU-only statements of k_flash_attn_pipe (gemm_q_hw.h): LDS-DMA as a copy (at once, or at the wave's vmcnt wait with EMU_DEFER_DMA=1)
#define CDNA4_HW_OVERRIDE
#define CDNA4_LDS_BASE(smem_) 0u
#define CDNA4_DMA16(voff, sbase, lds_addr) emu::vm_issue(smem + (lds_addr) + 16 * lane, (sbase) + (voff))
#define CDNA4_WAIT_VM(n) emu::vm_wait(n)
#define CDNA4_WAIT_LGKM0() ((void)0)
#include "../../ggml_amd/csrc/ops.hip"
#undef NEED
#include "../../ggml_amd/csrc/fattn.hip"

static void slurp(const char *p, void *dst, size_t n) {
    FILE *f = fopen(p, "rb"); if (!f || fread(dst, 1, n, f) != n) { perror(p); exit(2); } fclose(f);
}
int main(int argc, char **argv) {
    emu::g_defer_dma = getenv("EMU_DEFER_DMA") && atoi(getenv("EMU_DEFER_DMA")) != 0;
    if (argc < 19) { fprintf(stderr, "usage: fattn_emul D n_q n_head n_batch n_kv n_head_kv n_batch_kv has_mask mask_rows scale max_bias softcap permuted q k v mask out\n"); return 2; }
    const int64_t D = atoll(argv[1]), NQ = atoll(argv[2]), H = atoll(argv[3]), B3 = atoll(argv[4]), KV = atoll(argv[5]), HK = atoll(argv[6]), BK = atoll(argv[7]);
    const int has_mask = atoi(argv[8]); const int64_t MR = atoll(argv[9]);
    const float scale = (float)atof(argv[10]), max_bias = (float)atof(argv[11]), softcap = (float)atof(argv[12]); const int permuted = atoi(argv[13]);
    const int kvt = argc > 19 ? atoi(argv[19]) : CDNA4_F16;
    const int64_t kes = kvt == CDNA4_F16 ? 2 : (int64_t)tsize(kvt), kn0 = kvt == CDNA4_F16 ? D : D / bsize(kvt);      // element size / elements per row in the stride arithmetic below
    const size_t nq = (size_t)(B3 * H * NQ * D) * 4, nk = (size_t)(BK * HK * KV * kn0) * kes, nm = (size_t)(MR * KV) * 2, no = (size_t)(B3 * NQ * H * D) * 4;
    void *q = shared_alloc(nq), *k = shared_alloc(nk), *v = shared_alloc(nk), *m = has_mask ? shared_alloc(nm) : nullptr, *o = shared_alloc(no);
    slurp(argv[14], q, nq); slurp(argv[15], k, nk); slurp(argv[16], v, nk); if (has_mask) slurp(argv[17], m, nm);
    memset(o, 0xFF, no);
    ggml_cdna4_tensor tq{}, tk{}, tv{}, tm{}, td{};
    auto fill = [&](ggml_cdna4_tensor &t, void *data, int type, int64_t es, int64_t n0, int64_t n1, int64_t n2, int64_t n3, bool perm) {
        t.data = data; t.type = type; t.ne[0] = n0; t.ne[1] = n1; t.ne[2] = n2; t.ne[3] = n3;
        t.nb[0] = es; t.nb[3] = es * n0 * n1 * n2;
        if (!perm) { t.nb[1] = es * n0; t.nb[2] = es * n0 * n1; } else { t.nb[2] = es * n0; t.nb[1] = es * n0 * n2; }      // memory order [n3][n1][n2][n0]
    };
    fill(tq, q, CDNA4_F32, 4, D, NQ, H, B3, permuted); fill(tk, k, kvt, kes, kn0, KV, HK, BK, permuted); fill(tv, v, kvt, kes, kn0, KV, HK, BK, permuted);
    tk.ne[0] = tv.ne[0] = D;                                            // (fill() computed the strides from blocks per row)
    if (has_mask) fill(tm, m, CDNA4_F16, 2, KV, MR, 1, 1, false);
    fill(td, o, CDNA4_F32, 4, D, H, NQ, B3, false);
    if (ggml_cdna4_op_flash_attn_ext(&tq, &tk, &tv, has_mask ? &tm : nullptr, &td, scale, max_bias, softcap, nullptr)) return 1;
    FILE *f = fopen(argv[18], "wb"); fwrite(o, 1, no, f); fclose(f);
    return 0;
}
