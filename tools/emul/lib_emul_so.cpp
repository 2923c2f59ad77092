// lib_emul_so.cpp — the whole-library emulation build as a SHARED LIBRARY (lib_emul.h; same objects as lib_emul_main.cpp's program, plus fattn.hip):
// the library's own C-ABI, loadable with ctypes, so that tests written against the GPU library can drive the kernels' sources on the CPU.
// Test infrastructure: nothing under ggml_amd/ knows about it (tests/emul_torch.py swaps it in for tests only).  "Device" memory must come from
// cdna4_emul_alloc (shared mappings: the work-group processes of a launch write into them).
#include "lib_emul.h"

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx;
dim3 g_gridDim, g_blockDim;
pthread_barrier_t g_wg_barrier;
WaveState *g_waves;
thread_local std::vector<Pending> t_vmq;
bool g_defer_dma = false;
size_t g_weaken = 0;
}
__attribute__((aligned(16))) uint8_t smem[160 * 1024];
__attribute__((aligned(16))) uint8_t fa_dyn_lds[160 * 1024];           // fattn.hip: k_flash_attn_pipe
void *emu_shared_alloc(size_t n) {
    const size_t pg = 4096, body = (n + pg - 1) / pg * pg;
    char *p = (char *)mmap(nullptr, body + 2 * pg, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(2); }
    mprotect(p, pg, PROT_NONE); mprotect(p + pg + body, pg, PROT_NONE);
    return p + pg + ((body - n) & ~(size_t)255);
}
extern "C" void *cdna4_emul_alloc(size_t n) { return emu_shared_alloc(n ? n : 1); }
