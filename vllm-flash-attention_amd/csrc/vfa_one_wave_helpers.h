// Helpers of the kernels that run one wave per SIMD with the whole 512-register file (asm-pinned
// accumulators, asm-issued direct-to-LDS loads).
#pragma once
#include "vfa_kernel_params.h"

namespace vfa {

// Accumulation with the accumulator pinned to the AGPR half of the register file: the
// 16 accumulator tiles of a wave are exactly the 256 AGPRs, everything the vector unit touches
// (scores, fragments, addresses) lives in the 256 arch VGPRs, and the register allocator has no
// choice to make between the two files.  The statement is opaque to the compiler: operands written
// by vector instructions are fenced by operand_fence() below before the first MFMA that reads them.
template <bool BF16>
__device__ __forceinline__ void mma32_agpr(f32x16 &acc, const i32x4 a, const i32x4 b) {
    if constexpr (BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// Two wait states between the vector instructions that produced these MFMA operands and the MFMA
// statements that follow (the compiler pads nothing around an asm statement).
__device__ __forceinline__ void operand_fence(i32x4 &a, i32x4 &b, i32x4 &c, i32x4 &d) {
    asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// Direct-to-LDS load of 16 bytes per lane (1 KiB per wave instruction, written in lane order at
// the wave-uniform LDS address `lds_addr`), issued from an asm statement so that the compiler does
// not count it: its own wait-count pass would otherwise hold every later LDS read behind ALL
// outstanding loads of this kind (it cannot tell the slice being read from the one being
// fetched), which with one wave per SIMD exposes the whole global-memory latency in every step.
// The kernel waits for these loads itself (s_waitcnt vmcnt(0) before the step's barrier).  M0
// carries the LDS address and is restored; the descriptor words are scalar registers.
__device__ __forceinline__ void dma_16b_to_lds(const i32x4 rsrc, const unsigned voff, const unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %2, %1, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(rsrc), "v"(voff), "s"(lds_addr) : "memory");
}
// The same with 4 bytes per lane: lane L writes dword L of the 256 bytes at `lds_addr`, whatever the
// set of active lanes is.
__device__ __forceinline__ void dma_4b_to_lds(const i32x4 rsrc, const unsigned voff, const unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "buffer_load_dword %2, %1, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(rsrc), "v"(voff), "s"(lds_addr) : "memory");
}
// Four 16-byte-per-lane loads (two per descriptor) in one statement, M0 saved and restored once.
__device__ __forceinline__ void dma_16b_x4_to_lds(const i32x4 ra, const unsigned va0, const unsigned va1,
                                                  const unsigned la0, const unsigned la1, const i32x4 rb,
                                                  const unsigned vb0, const unsigned vb1, const unsigned lb0,
                                                  const unsigned lb1) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %1, 0 offen lds\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %1, 0 offen lds\n\t"
                 "s_mov_b32 m0, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %6, 0 offen lds\n\t"
                 "s_mov_b32 m0, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %6, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(ra), "v"(va0), "v"(va1), "s"(la0), "s"(la1), "s"(rb), "v"(vb0), "v"(vb1), "s"(lb0), "s"(lb1)
                 : "memory");
}
// 128-bit buffer descriptor (raw buffer, no stride, `bytes` records of one byte) in scalar registers.
__device__ __forceinline__ i32x4 make_rsrc(const void *base, const unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}


}  // namespace vfa
