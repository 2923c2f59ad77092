// gemm_q_x4l_hw.h — the statements of gemm_q_x4l.hip that only exist on the GPU (inline assembly for the LDS-DMA, the counted
// waits, the LDS base address), as macros.  tools/emul/x4l_emul.cpp defines X4L_HW_OVERRIDE and its own versions before
// including the kernel, so that the kernel SOURCE can be executed thread by thread on the CPU (functional check of its
// indexing, loop structure and barrier counts) with no conditional code in the kernel itself.  `smem` / `lane` are the kernel's.
#pragma once
#ifndef X4L_HW_OVERRIDE
#define X4L_LDS_BASE(smem_) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(smem_))
// one LDS-DMA wave-piece: lane L copies 16 bytes from sbase + voff to LDS address lds_addr + 16 L (scalar-base form, M0 = LDS address)
#define X4L_DMA16(voff, sbase, lds_addr) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0")
// 16-byte global load into registers, scalar base + 32-bit lane offset; asynchronous: the value is valid after a vmcnt wait tied to it
#define X4L_GLOAD16(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")
// s_waitcnt vmcnt(n), tied to four registers that earlier X4L_GLOAD16s produce
#define X4L_WAIT_VM_TIED4(n, a, b, c, d) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(n) : "memory")
#define X4L_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define X4L_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
