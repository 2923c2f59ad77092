// gemm_q_x4l_hw.h — the statements of gemm_q_x4l.hip that only exist on the GPU (inline assembly for the LDS-DMA, the counted
// waits, the LDS base address), as macros.  tools/emul/x4l_emul.cpp defines X4L_HW_OVERRIDE and its own versions before
// including the kernel, so that the kernel SOURCE can be executed thread by thread on the CPU (functional check of its
// indexing, loop structure and barrier counts) with no conditional code in the kernel itself.  `smem` / `lane` are the kernel's.
#pragma once
#ifndef X4L_HW_OVERRIDE
#define X4L_LDS_BASE(smem_) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(smem_))
// one LDS-DMA wave-piece: lane L copies 16 bytes from sbase + voff to LDS address lds_addr + 16 L (scalar-base form, M0 = LDS address)
#define X4L_DMA16(voff, sbase, lds_addr) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0")
// 16-byte global loads into registers (scalar base + 32-bit lane offset) that have ARRIVED when the statement ends: the load(s)
// and an s_waitcnt vmcnt(0) are one asm statement.  (Leaving the wait to a later statement does not work: the compiler
// regards an asm output as available at once and is free to copy the registers before the data is there — seen in the ISA.)
#define X4L_GLOAD16_SYNC(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory")
#define X4L_GLOAD16x2_SYNC(d0, d1, v0, v1, sbase) asm volatile("global_load_dwordx4 %0, %2, %4\n\tglobal_load_dwordx4 %1, %3, %4\n\ts_waitcnt vmcnt(0)" : "=&v"(d0), "=&v"(d1) : "v"(v0), "v"(v1), "s"(sbase) : "memory")
#define X4L_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define X4L_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
