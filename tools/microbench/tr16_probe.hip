// tr16_probe.hip — pins the semantics of ds_read_b64_tr_b16 that k_flash_attn_pipe (fattn.hip) relies on: within a 16-lane group, lane i supplies the address of
// elements [row i / 4][cols 4 (i % 4) .. + 3] of a 4 x 16 block of 2-byte elements (any row stride) and receives column i: [row 0..3][col i].  Prints PASS / FAIL.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned short u16;
__global__ void k(const u16 *in, u16 *out, int stride) {
    __shared__ __attribute__((aligned(16))) u16 s[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) s[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const u16 *a = s + (g * 4 + (i >> 2)) * stride + 16 * (g & 1) + 4 * (i & 3);
    fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t *)a);
    *(fp16x4_t *)(out + 4 * l) = r;
}
// v_permlane32_swap with both results pinned (fattn.hip: fa_swap32): out[l] = v[l % 32] + 2 v[32 + l % 32]
__global__ void k_swap(const float *in, float *out) {
    const unsigned a = __builtin_bit_cast(unsigned, in[threadIdx.x]);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    unsigned r0 = r[0], r1 = r[1];
    asm volatile("" : "+v"(r0), "+v"(r1));
    out[threadIdx.x] = __builtin_bit_cast(float, r0) + 2.0f * __builtin_bit_cast(float, r1);
}
int main() {
    u16 h[8192], o[256], *di, *dout;
    for (int i = 0; i < 8192; i++) h[i] = (u16)i;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    int bad = 0;
    for (int stride : {16, 64, 72, 128}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, stride);
        hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
            const int i = l & 15, g = l >> 4;
            const u16 want = (u16)((g * 4 + j) * stride + 16 * (g & 1) + i);
            if (o[4 * l + j] != want) { if (bad < 8) printf("stride %d lane %d elem %d: got %u want %u\n", stride, l, j, o[4 * l + j], want); bad++; }
        }
    }
    {
        float hf[64], of[64], *df, *dof;
        for (int i = 0; i < 64; i++) hf[i] = (float)(i + 1);
        hipMalloc(&df, sizeof hf); hipMalloc(&dof, sizeof of);
        hipMemcpy(df, hf, sizeof hf, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, df, dof);
        hipMemcpy(of, dof, sizeof of, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) if (of[l] != hf[l % 32] + 2.0f * hf[32 + l % 32]) { if (bad < 8) printf("swap lane %d: got %g\n", l, of[l]); bad++; }
    }
    printf(bad ? "tr16_probe FAIL (%d)\n" : "tr16_probe PASS\n", bad);
    return bad != 0;
}
