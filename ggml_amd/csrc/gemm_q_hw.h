// gemm_q_hw.h — the statements of the LDS-DMA GEMM kernels (gemm_kq_w12.inc, gemm_kq_t64.inc) that only exist on the GPU — inline assembly for
// the LDS-DMA, asynchronous register loads, the counted waits, the LDS base address — as macros.  tools/emul/ defines
// CDNA4_HW_OVERRIDE and host versions before including the kernel, so that the kernel SOURCE can be executed on the CPU (a
// functional check of indexing, loop structure, barrier counts and the epilogue variants) with no conditional code in the
// kernel itself.  `smem` and `lane` are the kernel's.
#pragma once
#ifndef CDNA4_HW_OVERRIDE
#define CDNA4_LDS_BASE(smem_) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(smem_))
// one LDS-DMA wave-piece: lane L copies 16 bytes from sbase + voff to LDS address lds_addr + 16 L (scalar-base form, M0 = LDS address)
#define CDNA4_DMA16(voff, sbase, lds_addr) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0")
// the same with the sc1 cache policy: served by the L2 / memory side, never by this CU's L1 — for bytes another work-group of the SAME launch stored write-through
// (the activation image of the one-launch step, k_gemm_kq_t64<.., FQ>)
#define CDNA4_DMA16_SC1(voff, sbase, lds_addr) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0")
// 16-byte global load into registers from a per-lane pointer; asynchronous: valid after a vmcnt wait tied to the destination
#define CDNA4_GLOAD16_PTR(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define CDNA4_WAIT_VM_TIED1(n, a) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(n) : "memory")
#define CDNA4_WAIT_VM_TIED2(n, a, b) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(n) : "memory")
#define CDNA4_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define CDNA4_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// counted wait with a wave-uniform RUN-TIME count (s_waitcnt takes an immediate: a scalar branch ladder; counts above 24 wait for 24, which is stricter)
#define CDNA4_WAIT_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void cdna4_wait_vm_rt(int n) {
    switch (n < 0 ? 0 : (n > 24 ? 24 : n)) {
        CDNA4_WAIT_VM_CASE(0) CDNA4_WAIT_VM_CASE(1) CDNA4_WAIT_VM_CASE(2) CDNA4_WAIT_VM_CASE(3) CDNA4_WAIT_VM_CASE(4) CDNA4_WAIT_VM_CASE(5) CDNA4_WAIT_VM_CASE(6)
        CDNA4_WAIT_VM_CASE(7) CDNA4_WAIT_VM_CASE(8) CDNA4_WAIT_VM_CASE(9) CDNA4_WAIT_VM_CASE(10) CDNA4_WAIT_VM_CASE(11) CDNA4_WAIT_VM_CASE(12) CDNA4_WAIT_VM_CASE(13)
        CDNA4_WAIT_VM_CASE(14) CDNA4_WAIT_VM_CASE(15) CDNA4_WAIT_VM_CASE(16) CDNA4_WAIT_VM_CASE(17) CDNA4_WAIT_VM_CASE(18) CDNA4_WAIT_VM_CASE(19) CDNA4_WAIT_VM_CASE(20)
        CDNA4_WAIT_VM_CASE(21) CDNA4_WAIT_VM_CASE(22) CDNA4_WAIT_VM_CASE(23) CDNA4_WAIT_VM_CASE(24)
    }
}
// the same as an instruction the compiler's own wait-count pass SEES (vmcnt 63, expcnt 7, lgkmcnt 0): behind it hipcc knows every ds_read has returned and
// inserts no lgkmcnt wait of its own in front of the MFMAs that use fragments requested a phase earlier (k_gemm_lds; it cannot see an asm wait)
// keeps a 32-bit value materialized HERE: without it hipcc sinks arithmetic whose result is only used behind a later branch or barrier down to that use
// (seen in k_gemm_lds: the dequantizer pieces meant for the MFMA gaps were moved into the next phase's conditional ds_write block)
#define CDNA4_PIN(x) asm volatile("" : "+v"(x))
#define CDNA4_WAIT_LGKM0_VISIBLE() do { __builtin_amdgcn_s_waitcnt(0xC07F); asm volatile("" ::: "memory"); } while (0)
// the same with only lanes [0, nlanes) active (nlanes = 16 / 32, a literal).  EXEC is narrowed INSIDE the statement: an `if (lane < n)`
// around CDNA4_DMA16 makes hipcc merge uniform address arithmetic across the divergent join into VGPRs, which the "s" operands reject
#define CDNA4_DMA16_LANES(voff, sbase, lds_addr, nlanes) do { uint64_t cdna4_exec_;                                                   \
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0" \
                 : "=&s"(cdna4_exec_) : "v"(voff), "s"(sbase), "s"(lds_addr), "i"((nlanes) >= 64 ? -1ll : (long long)((1ull << ((nlanes) & 63)) - 1)) : "memory", "m0"); } while (0)
// v_permlane32_swap: lanes 32-63 of `a` trade places with lanes 0-31 of `b` (both 32-bit); with a == b on entry every lane l ends
// with a = the value of lane l % 32 and b = the value of lane 32 + l % 32
#define CDNA4_SWAP32(a, b) do { auto r_ = __builtin_amdgcn_permlane32_swap((a), (b), false, false); (a) = r_[0]; (b) = r_[1]; } while (0)
#endif
