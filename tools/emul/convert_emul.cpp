// convert_emul.cpp — runs the SOURCE of the exact weight re-encodings (ggml_amd/csrc/convert_w.hip: k_convert_q5_0_q8_0, k_convert_q3_K_q6_K, k_convert_q2_K_q6_K2) on
// the CPU.  Test infrastructure.  The kernels have no barriers and no cross-lane traffic, so the GPU threads simply run one after the other.
//   convert_emul type M K w.bin out.bin          type: 6 = Q5_0 (-> Q8_0), 11 = Q3_K (-> Q6_K), 10 = Q2_K (-> Q6_K scale part | Q6_K minimum part), 20 = IQ4_NL (-> Q8_0),
//                                                3 / 7 = Q4_1 / Q5_1 (-> Q8_0 scale part | Q8_0 minimum part), 23 = IQ4_XS (-> Q6_K h part | Q6_K l part); rows contiguous
#include "hip_emul.h"
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
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    do { const dim3 g_ = (grid), b_ = (block); emu::g_gridDim = g_; emu::g_blockDim = b_; \
         for (unsigned bi = 0; bi < g_.x; bi++) for (unsigned ti = 0; ti < b_.x; ti++) { emu::t_blockIdx = dim3(bi); emu::t_threadIdx = dim3(ti); kernel(__VA_ARGS__); } } while (0)

#include "../../ggml_amd/csrc/convert_w.hip"

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: convert_emul type M K w.bin out.bin\n"); return 2; }
    const int type = atoi(argv[1]); const int64_t M = atoll(argv[2]), K = atoll(argv[3]);
    const size_t in_row = type == CDNA4_Q5_0 ? (size_t)(K / 32) * 22 : type == CDNA4_Q4_1 ? (size_t)(K / 32) * 20 : type == CDNA4_Q5_1 ? (size_t)(K / 32) * 24 : type == CDNA4_IQ4_NL ? (size_t)(K / 32) * 18 :
                          (size_t)(K / 256) * (type == CDNA4_Q2_K ? 84 : type == CDNA4_IQ4_XS ? 136 : 110), out_bytes = cdna4_convert_weights_bytes(type, M, K);
    // exact-size heap blocks: an out-of-bounds access is for the address sanitizer / valgrind to see, the sizes are asserted by the checker
    std::vector<uint8_t> w(in_row * M), out(out_bytes, 0xAA);
    FILE *f = fopen(argv[4], "rb"); if (!f || fread(w.data(), 1, w.size(), f) != w.size()) { perror(argv[4]); return 2; } fclose(f);
    if (cdna4_launch_convert_weights(type, w.data(), (int64_t)in_row, M, K, out.data(), nullptr)) return 1;
    f = fopen(argv[5], "wb"); fwrite(out.data(), 1, out.size(), f); fclose(f);
    return 0;
}
