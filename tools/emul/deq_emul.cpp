// deq_emul.cpp — runs the SOURCE of the to_float kernels (ggml_amd/csrc/ops.hip: k_cpy_q_to_f32 / deq_elem behind ggml_cdna4_dequantize_row) on
// the CPU.  Test infrastructure.  These kernels have no barriers and no cross-lane traffic: the GPU threads run one after the other.
//   deq_emul type K rows.bin out.bin            one quantized row buffer of K weights -> K floats
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
#define hipMemcpyDeviceToDevice 0
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    do { const dim3 g_ = (grid), b_ = (block); emu::g_gridDim = g_; emu::g_blockDim = b_; \
         for (unsigned bi = 0; bi < g_.x; bi++) for (unsigned ti = 0; ti < b_.x; ti++) { emu::t_blockIdx = dim3(bi); emu::t_threadIdx = dim3(ti); kernel(__VA_ARGS__); } } while (0)

#include "../../ggml_amd/csrc/ops.hip"

// deq_emul f16 type K nrows gap rows.bin out.bin : k_q_to_f16_dense (the fp16 copy of a quantized K / V in front of FLASH_ATTN_EXT) on nrows rows of K
// weights that lie `gap` bytes further apart than their size, as ne = (K, nrows / 2, 2, 1) -> nrows * K halves, dense
static int main_f16(int argc, char **argv) {
    const int type = atoi(argv[2]); const int64_t K = atoll(argv[3]), nrows = atoll(argv[4]), gap = atoll(argv[5]);
    FILE *f = fopen(argv[6], "rb"); if (!f) { perror(argv[6]); return 2; }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> w((size_t)n); if (fread(w.data(), 1, w.size(), f) != w.size()) return 2; fclose(f);
    const int64_t rb = (int64_t)(K / bsize(type)) * (int64_t)tsize(type);
    if ((int64_t)w.size() != nrows * (rb + gap) || nrows % 2) { fprintf(stderr, "bad input size\n"); return 2; }
    T4 a{}; a.data = w.data(); a.type = type; a.ne[0] = K; a.ne[1] = nrows / 2; a.ne[2] = 2; a.ne[3] = 1;
    a.nb[0] = (int64_t)tsize(type); a.nb[1] = rb + gap; a.nb[2] = a.nb[1] * a.ne[1]; a.nb[3] = a.nb[2] * 2;
    std::vector<uint16_t> y((size_t)(nrows * K), 0xAAAA);
    if (cdna4_launch_to_f16_dense(&a, y.data(), K, nullptr)) return 1;
    f = fopen(argv[7], "wb"); fwrite(y.data(), 2, y.size(), f); fclose(f);
    return 0;
}
// deq_emul cpyq type K nrows x.bin out.bin : ggml_cdna4_op_cpy F32 [K, nrows] -> block type (the CPY quantizers: from_float of the type)
static int main_cpyq(int argc, char **argv) {
    const int type = atoi(argv[2]); const int64_t K = atoll(argv[3]), nrows = atoll(argv[4]);
    std::vector<float> x((size_t)(K * nrows));
    FILE *f = fopen(argv[5], "rb"); if (!f || fread(x.data(), 4, x.size(), f) != x.size()) { perror(argv[5]); return 2; } fclose(f);
    const int64_t rb = (int64_t)(K / bsize(type)) * (int64_t)tsize(type);
    std::vector<uint8_t> y((size_t)(rb * nrows), 0xAA);
    T4 a{}, d{};
    a.data = x.data(); a.type = CDNA4_F32; a.ne[0] = K; a.ne[1] = nrows; a.ne[2] = a.ne[3] = 1; a.nb[0] = 4; a.nb[1] = 4 * K; a.nb[2] = a.nb[3] = 4 * K * nrows;
    d.data = y.data(); d.type = type; d.ne[0] = K; d.ne[1] = nrows; d.ne[2] = d.ne[3] = 1; d.nb[0] = (int64_t)tsize(type); d.nb[1] = rb; d.nb[2] = d.nb[3] = rb * nrows;
    if (ggml_cdna4_op_cpy(&a, &d, 1, nullptr)) return 1;
    f = fopen(argv[6], "wb"); fwrite(y.data(), 1, y.size(), f); fclose(f);
    return 0;
}
int main(int argc, char **argv) {
    if (argc >= 8 && !strcmp(argv[1], "f16")) return main_f16(argc, argv);
    if (argc >= 7 && !strcmp(argv[1], "cpyq")) return main_cpyq(argc, argv);
    if (argc < 5) { fprintf(stderr, "usage: deq_emul type K rows.bin out.bin\n"); return 2; }
    const int type = atoi(argv[1]); const int64_t K = atoll(argv[2]);
    FILE *f = fopen(argv[3], "rb"); if (!f) { perror(argv[3]); return 2; }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> w((size_t)n); if (fread(w.data(), 1, w.size(), f) != w.size()) return 2; fclose(f);
    std::vector<float> y((size_t)K);
    if (ggml_cdna4_dequantize_row(type, w.data(), y.data(), K, nullptr)) return 1;
    f = fopen(argv[4], "wb"); fwrite(y.data(), 4, y.size(), f); fclose(f);
    return 0;
}
