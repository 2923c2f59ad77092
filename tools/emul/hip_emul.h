// hip_emul.h — just enough of the HIP device environment to run a kernel's SOURCE on the host, one OS thread per GPU thread
// (test infrastructure; see x4l_emul.cpp).  Include it FIRST: it pre-empts <hip/hip_runtime.h> (tools/emul/shim/hip/).
// What is emulated faithfully: threadIdx / blockIdx, work-group barriers, LDS as ordinary memory, the lane layout and
// arithmetic of v_mfma_f32_32x32x16_f16 (fp32 accumulate), plain loads / stores.  What is NOT: asynchrony (LDS-DMA and loads
// complete immediately, so a missing wait cannot be detected), timing, wave scheduling, bank conflicts.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <vector>
using std::max;
using std::min;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static                 // one work-group runs at a time: a static local is shared by its threads
#define __restrict__
typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
template <typename F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }

namespace emu {
struct WaveState { pthread_barrier_t bar; _Float16 A[64][8], B[64][8]; uint32_t xch[64]; };
extern thread_local dim3 t_threadIdx, t_blockIdx;
extern dim3 g_gridDim, g_blockDim;
extern pthread_barrier_t g_wg_barrier;
extern WaveState *g_waves;                // one per wave of the running work-group
inline WaveState &my_wave() { return g_waves[t_threadIdx.x >> 6]; }

// ---- synchronisation with SIMT semantics.  One OS thread per lane means that a lane which branches AROUND a wave-collective (shuffle, MFMA,
// permlane swap) or leaves the kernel never arrives at it, where the GPU simply runs the collective with the lanes that are active.  So a
// wave-collective completes when every lane of the wave has either arrived at it, is waiting at the work-group barrier, or has left the kernel —
// i.e. when nobody else can still come; the work-group barrier likewise counts lanes that have left.  (Lanes that skip a collective reach the
// next barrier or the end of the kernel, which is what tells the others not to wait for them.)  One lock and condition per wave, one for the barrier.
struct WaveSync { pthread_mutex_t m = PTHREAD_MUTEX_INITIALIZER; pthread_cond_t cv = PTHREAD_COND_INITIALIZER; int arrived = 0, at_bar = 0, exited = 0; unsigned gen = 0, at_bar_gen = 0; };
struct Sync {
    WaveSync w[16];
    pthread_mutex_t mb = PTHREAD_MUTEX_INITIALIZER; pthread_cond_t cb = PTHREAD_COND_INITIALIZER;
    int b_arrived = 0, exited = 0; unsigned b_gen = 0;                  // b_gen is read without mb (atomically) by the waves
};
inline Sync g_sync;
inline int n_threads() { return (int)(g_blockDim.x * g_blockDim.y * g_blockDim.z); }
inline int wave_lanes(int w) { const int n = n_threads() - 64 * w; return n < 64 ? n : 64; }
// (with ws.m held) lanes marked "at the barrier" count only while that barrier instance has not been released: the mark carries its generation
inline void try_release_wave(Sync &s, int w) {
    WaveSync &ws = s.w[w];
    const int at_bar = ws.at_bar_gen == __atomic_load_n(&s.b_gen, __ATOMIC_ACQUIRE) ? ws.at_bar : 0;
    if (ws.arrived > 0 && ws.arrived + at_bar + ws.exited == wave_lanes(w)) { ws.arrived = 0; ws.gen++; pthread_cond_broadcast(&ws.cv); }
}
inline void wave_sync() {
    Sync &s = g_sync; const int w = (int)(t_threadIdx.x >> 6); WaveSync &ws = s.w[w];
    pthread_mutex_lock(&ws.m);
    const unsigned g = ws.gen; ws.arrived++; try_release_wave(s, w);
    while (ws.gen == g) pthread_cond_wait(&ws.cv, &ws.m);
    pthread_mutex_unlock(&ws.m);
}
inline void wg_barrier() {
    Sync &s = g_sync; const int w = (int)(t_threadIdx.x >> 6); WaveSync &ws = s.w[w];
    pthread_mutex_lock(&ws.m);
    const unsigned bg = __atomic_load_n(&s.b_gen, __ATOMIC_ACQUIRE);
    if (ws.at_bar_gen != bg) { ws.at_bar = 0; ws.at_bar_gen = bg; }
    ws.at_bar++; try_release_wave(s, w);
    pthread_mutex_unlock(&ws.m);
    pthread_mutex_lock(&s.mb);
    const unsigned g = s.b_gen;
    if (++s.b_arrived + s.exited == n_threads()) { s.b_arrived = 0; __atomic_store_n(&s.b_gen, g + 1, __ATOMIC_RELEASE); pthread_cond_broadcast(&s.cb); }
    else while (s.b_gen == g) pthread_cond_wait(&s.cb, &s.mb);
    pthread_mutex_unlock(&s.mb);
}
struct LaneGuard {                       // armed by the lane's first read of threadIdx; its destructor runs when the lane's OS thread ends
    int wave = -1;
    ~LaneGuard() {
        if (wave < 0) return;
        Sync &s = g_sync; WaveSync &ws = s.w[wave];
        pthread_mutex_lock(&ws.m); ws.exited++; try_release_wave(s, wave); pthread_mutex_unlock(&ws.m);
        pthread_mutex_lock(&s.mb);
        s.exited++;
        if (s.b_arrived > 0 && s.b_arrived + s.exited == n_threads()) { s.b_arrived = 0; __atomic_store_n(&s.b_gen, s.b_gen + 1, __ATOMIC_RELEASE); pthread_cond_broadcast(&s.cb); }
        if (s.exited == n_threads()) {                                  // the last lane out: the next work-group of this process starts clean
            for (int w = 0; w < 16; w++) { s.w[w].arrived = s.w[w].at_bar = s.w[w].exited = 0; }
            s.b_arrived = s.exited = 0;
        }
        pthread_mutex_unlock(&s.mb);
    }
};
inline dim3 &tid_ref() { static thread_local LaneGuard guard; if (guard.wave < 0) guard.wave = (int)(t_threadIdx.x >> 6); return t_threadIdx; }
inline int sync_point(pthread_barrier_t *b) { if (b == &g_wg_barrier) wg_barrier(); else wave_sync(); return 0; }
}  // namespace emu
#define pthread_barrier_wait(b) emu::sync_point(b)      // (the harnesses' pthread barriers stay initialised but idle)
#define threadIdx emu::tid_ref()
#define blockIdx emu::t_blockIdx
#define gridDim emu::g_gridDim
#define blockDim emu::g_blockDim
#define __syncthreads() emu::wg_barrier()
#define __builtin_amdgcn_s_barrier() emu::wg_barrier()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)      // only ever applied to wave-uniform values in these kernels
// spin loops are bounded by iteration count: keep them slow enough for emulated partners, but give up after ~2 minutes of
// waiting (a partner that is not being emulated, e.g. with EMU_BLOCKS) instead of spinning for hours
static inline void emu_sleep() { static thread_local unsigned n = 0; usleep(200); if (++n > 600000u) { fprintf(stderr, "emulated spin loop gave up\n"); _exit(9); } }
#define __builtin_amdgcn_s_sleep(x) emu_sleep()
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
#define __builtin_amdgcn_exp2f(x) exp2f(x)             // v_exp_f32
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_memrealtime() 0ull
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, order)
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, order)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, order)
// cross-lane helpers the shared headers declare but the emulated kernels never call
// __shfl_xor: wave-collective through a per-wave exchange buffer among the lanes that execute it (see Sync above)
template <typename T> static inline T __shfl_xor(T v, int o, int = 64) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    emu::WaveState &w = emu::my_wave();
    const int l = emu::t_threadIdx.x & 63;
    memcpy(&w.xch[l], &v, 4);
    pthread_barrier_wait(&w.bar);
    T r; memcpy(&r, &w.xch[l ^ o], 4);
    pthread_barrier_wait(&w.bar);
    return r;
}
// __shfl_up: lane l receives the value of lane l - d (its own below d); wave-collective like __shfl_xor
template <typename T> static inline T emu_shfl_idx(T v, int src);
template <typename T> static inline T __shfl_up(T v, unsigned d, int = 64) {
    const int l = emu::t_threadIdx.x & 63;
    return emu_shfl_idx(v, l >= (int)d ? l - (int)d : l);
}
// value of lane `src` (any lane of the wave): the fallback of cdna4_common.h's DPP helpers and of v_readlane
template <typename T> static inline T emu_shfl_idx(T v, int src) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    emu::WaveState &w = emu::my_wave();
    const int l = emu::t_threadIdx.x & 63;
    memcpy(&w.xch[l], &v, 4);
    pthread_barrier_wait(&w.bar);
    T r; memcpy(&r, &w.xch[src & 63], 4);
    pthread_barrier_wait(&w.bar);
    return r;
}
#define __builtin_amdgcn_readlane(v, lane_) emu_shfl_idx((int)(v), (lane_))
static inline int emu_sdot4(int a, int b, int c) { for (int i = 0; i < 4; i++) c += (int)(int8_t)(a >> (8 * i)) * (int)(int8_t)(b >> (8 * i)); return c; }
#define __builtin_amdgcn_sdot4(a, b, c, clamp) emu_sdot4(a, b, c)
// LDS-DMA, builtin form: every lane copies `size` bytes from ITS global address to the wave's LDS base + lane * size (performed at once)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) memcpy((char *)(uintptr_t)(l) + (off) + (emu::t_threadIdx.x & 63) * (size), (const void *)(uintptr_t)(g), (size))
// buffer resources (split-K exchange): a descriptor is just the base pointer here
typedef void *__amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_make_buffer_rsrc(ptr, stride, num, flags) ((void *)(ptr))
template <typename V> static inline void emu_buffer_store_b128(V v, void *rsrc, int voff) { memcpy((char *)rsrc + voff, &v, 16); }
#define __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, soff, aux) emu_buffer_store_b128(v, rsrc, voff)

typedef uint32_t emu_u32x4 __attribute__((ext_vector_type(4)));
static inline emu_u32x4 emu_buffer_load_b128(void *rsrc, int voff) { emu_u32x4 v; memcpy(&v, (char *)rsrc + voff, 16); return v; }
#define __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, aux) emu_buffer_load_b128(rsrc, voff)
#define __builtin_amdgcn_s_getreg(x) 0u             // HW_REG_XCC_ID: every emulated work-group sits on "XCD 0"
#define __builtin_amdgcn_s_setprio(x) ((void)0)
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline void unsafeAtomicAdd(float *p, float v) {
    uint32_t *u = reinterpret_cast<uint32_t *>(p), old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    do { float f; memcpy(&f, &old, 4); f += v; memcpy(&nw, &f, 4); } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}

// vector-memory queue of one lane, for the counted waits: with EMU_DEFER_DMA=1 an LDS-DMA copy is performed as LATE as the
// hardware allows — when an s_waitcnt vmcnt(n) of the issuing wave retires it (in order, all but the newest n) — so a wait
// that is missing or too weak leaves stale bytes in LDS and the result goes wrong.  The default (immediate copies) is the
// other extreme, which is the one that exposes a slot being overwritten too EARLY.  Register loads always complete at once
// but still occupy a queue entry, as they do in the hardware's counter.
namespace emu {
struct Pending { void *dst; const void *src; };
extern thread_local std::vector<Pending> t_vmq;
extern bool g_defer_dma;
inline void vm_issue(void *dst, const void *src) { if (g_defer_dma) t_vmq.push_back({dst, src}); else { memcpy(dst, src, 16); t_vmq.push_back({nullptr, nullptr}); } }
inline void vm_issue_done() { t_vmq.push_back({nullptr, nullptr}); }
extern size_t g_weaken;                   // EMU_WEAKEN_WAITS=k: every vmcnt wait tolerates k more outstanding operations (self-test of the check)
inline void vm_wait(size_t n) {
    n += g_weaken;
    while (t_vmq.size() > n) { const Pending p = t_vmq.front(); t_vmq.erase(t_vmq.begin()); if (p.dst) memcpy(p.dst, p.src, 16); }
}
}  // namespace emu

// v_mfma_f32_32x32x16_f16, wave-collective: lane l supplies A[row l % 32][k 8 (l / 32) .. + 7] and B[k ..][col l % 32] and
// receives D[row (r & 3) + 8 (r >> 2) + 4 (l / 32)][col l % 32] for r = 0 .. 15
typedef _Float16 emu_half8 __attribute__((ext_vector_type(8)));
typedef float emu_floatx16 __attribute__((ext_vector_type(16)));
static inline emu_floatx16 emu_mfma_32x32x16_f16(emu_half8 a, emu_half8 b, emu_floatx16 c) {
    emu::WaveState &w = emu::my_wave();
    const int l = emu::t_threadIdx.x & 63;
    for (int e = 0; e < 8; e++) { w.A[l][e] = a[e]; w.B[l][e] = b[e]; }
    pthread_barrier_wait(&w.bar);
    const int n = l & 31, hh = l >> 5;
    for (int r = 0; r < 16; r++) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
        float s = 0.f;
        for (int kh = 0; kh < 2; kh++)
            for (int e = 0; e < 8; e++) s += (float)w.A[i + 32 * kh][e] * (float)w.B[n + 32 * kh][e];
        c[r] += s;
    }
    pthread_barrier_wait(&w.bar);
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_32x32x16_f16(a, b, c)
// v_mfma_i32_16x16x32_i8, wave-collective: lane l supplies A[row l % 16][k 8 (l / 16) .. + 7] and B[k ..][col l % 16] (eight int8 in a 64-bit
// operand each) and receives D[row 4 (l / 16) + v][col l % 16] for v = 0 .. 3
typedef int emu_intx4 __attribute__((ext_vector_type(4)));
static inline emu_intx4 emu_mfma_i32_16x16x32_i8(long a, long b, emu_intx4 c) {
    emu::WaveState &w = emu::my_wave();
    const int l = emu::t_threadIdx.x & 63;
    static_assert(sizeof(w.A[0]) >= 8 && sizeof(w.B[0]) >= 8, "operand staging");
    memcpy(&w.A[l][0], &a, 8); memcpy(&w.B[l][0], &b, 8);
    pthread_barrier_wait(&w.bar);
    const int n = l & 15, g = l >> 4;
    for (int v = 0; v < 4; v++) {
        const int i = 4 * g + v; int s = 0;
        for (int kg = 0; kg < 4; kg++) {
            int8_t ra[8], rb[8]; memcpy(ra, &w.A[i + 16 * kg][0], 8); memcpy(rb, &w.B[n + 16 * kg][0], 8);
            for (int e = 0; e < 8; e++) s += (int)ra[e] * (int)rb[e];
        }
        c[v] += s;
    }
    pthread_barrier_wait(&w.bar);
    return c;
}
#define __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, x, y, z) emu_mfma_i32_16x16x32_i8(a, b, c)
