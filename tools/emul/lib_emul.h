// lib_emul.h — the WHOLE kernel library on the CPU: every .hip of ggml_amd/csrc (except fattn.hip, which has its own harness) is compiled as host
// C++ with this header force-included (-include), linked with lib_emul_main.cpp, and driven through the library's own C-ABI — so that the host
// code between the entry points and the kernels (capi.hip: routing, workspace carving, the doubled activation image; gemm_q_mfma.hip: kernel
// selection, the per-call re-encodings and re-layouts, split-K scratch) is executed too, not just each kernel by itself.  Test infrastructure.
// A kernel launch forks one process per work-group (LDS = that process's statics / the `smem` array), one OS thread per GPU thread; device memory
// comes from hipMalloc = MAP_SHARED mappings between guard pages, so the work-groups of a launch see each other's global writes (split-K
// exchanges run between co-resident work-groups as on the GPU).
#pragma once
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
#define CDNA4_DMA16_SC1(voff, sbase, lds_addr) CDNA4_DMA16(voff, sbase, lds_addr)
#define CDNA4_GLOAD16_PTR(dst, ptr) (memcpy(&(dst), (ptr), 16), emu::vm_issue_done())
#define CDNA4_WAIT_VM_TIED1(n, a) emu::vm_wait(n)
#define CDNA4_WAIT_VM_TIED2(n, a, b) emu::vm_wait(n)
#define CDNA4_WAIT_VM(n) emu::vm_wait(n)
#define cdna4_wait_vm_rt(n) emu::vm_wait(n)
#define CDNA4_WAIT_LGKM0() ((void)0)
#define CDNA4_WAIT_LGKM0_VISIBLE() ((void)0)
#define CDNA4_PIN(x) ((void)0)
#define CDNA4_DMA16_LANES(voff, sbase, lds_addr, nlanes) do { if (lane < (nlanes)) emu::vm_issue(smem + (lds_addr) + 16 * lane, (sbase) + (voff)); else emu::vm_issue_done(); } while (0)
#define CDNA4_SWAP32(a, b) do { emu::WaveState &w_ = emu::my_wave(); const int l_ = emu::t_threadIdx.x & 63; const uint32_t a_ = (a), b_ = (b); \
    w_.xch[l_] = l_ < 32 ? b_ : a_; pthread_barrier_wait(&w_.bar); const uint32_t o_ = w_.xch[l_ ^ 32]; pthread_barrier_wait(&w_.bar); \
    if (l_ < 32) (b) = o_; else (a) = o_; } while (0)

#ifdef EMU_DYNAMIC_LDS                 // gemv_q.hip: `extern __shared__ uint8_t smem[]` is the array lib_emul_main.cpp defines
#undef __shared__
#define __shared__
#endif

// ---- the HIP runtime calls the library's host code makes
struct hipDeviceProp_t { int multiProcessorCount; };
#define hipMemcpyDeviceToDevice 0
#define hipMemcpyDeviceToHost 0
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
void *emu_shared_alloc(size_t n);
static inline hipError_t hipMalloc(void **p, size_t n) { *p = emu_shared_alloc(n); return 0; }
static inline hipError_t hipFree(void *) { return 0; }
#define hipHostMallocPortable 1
#define hipHostMallocMapped 2
static inline hipError_t hipHostMalloc(void **p, size_t n, int) { *p = emu_shared_alloc(n); return 0; }      // (the fault word: shared with the work-group processes)
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, int) { *d = h; return 0; }                                  // (guard-paged mappings are left in place)
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memmove(d, s, n); return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { const char *e = getenv("EMU_CUS"); p->multiProcessorCount = e ? atoi(e) : 256; return 0; }

// ---- kernel launch: every work-group a process (all of a launch at once up to 48 work-groups, so that work-groups that wait for each other
// are co-resident; larger grids in batches of 16 — the launchers only pair work-groups on grids that fit the chip, see EMU_CUS)
template <typename F> static void emu_launch(F body, dim3 grid, dim3 block) {
    emu::g_gridDim = grid; emu::g_blockDim = block;
    const int nthreads = (int)(block.x * block.y * block.z);
    const unsigned total = grid.x * grid.y * grid.z, batch = total <= 48 ? total : 16;
    std::vector<pid_t> kids;
    bool cannot = false, failed = false;
    auto reap = [&](size_t keep) {
        while (kids.size() > keep) { int st = 0; waitpid(kids.front(), &st, 0); kids.erase(kids.begin());
            if (WIFEXITED(st) && WEXITSTATUS(st) == 77) cannot = true;
            else if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { failed = true; if (WIFSIGNALED(st)) fprintf(stderr, "work-group process killed by signal %d (11 = an access outside the buffers)\n", WTERMSIG(st)); } }
    };
    for (unsigned b = 0; b < total; b++) {
        if (batch < total) reap(batch - 1);
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); exit(77); }
        if (pid > 0) { kids.push_back(pid); continue; }
        prctl(PR_SET_PDEATHSIG, SIGKILL);
        const dim3 bi(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
        pthread_barrier_init(&emu::g_wg_barrier, nullptr, nthreads);
        std::vector<emu::WaveState> waves((nthreads + 63) / 64);
        for (auto &w : waves) pthread_barrier_init(&w.bar, nullptr, 64);
        emu::g_waves = waves.data();
        std::vector<std::thread> th;
        try {
            for (int t = 0; t < nthreads; t++) th.emplace_back([&, t, bi] { emu::t_threadIdx = dim3(t); emu::t_blockIdx = bi; body(); });
        } catch (const std::system_error &) { _exit(77); }
        for (auto &t : th) t.join();
        _exit(0);
    }
    reap(0);
    if (failed) { fprintf(stderr, "work-group process failed\n"); exit(3); }
    if (cannot) { fprintf(stderr, "the environment cannot host the emulation (process / thread limits)\n"); exit(77); }
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) emu_launch([&](void) { kernel(__VA_ARGS__); }, dim3(grid), dim3(block))
