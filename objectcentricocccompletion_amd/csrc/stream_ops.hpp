// Inline-asm memory operations of the hand-scheduled gather-GEMM loops (sparse_conv.hip, sparse_conv_sorted.hip).
#pragma once
#include "common.hpp"

namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
  // everything but the N youngest vector-memory operations has landed (global_load_lds writes
  // included), this wave's LDS traffic is done, then the workgroup barrier
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// The loads of the per-offset loop are issued from inline asm.  The compiler's wait-count pass treats
// an LDS-DMA in flight as a possible writer of every LDS address and put `s_waitcnt vmcnt(0)` in front
// of the weight-fragment reads: the MFMAs of every second offset waited for the whole prefetch of the
// next one, and no prefetch could run more than one offset ahead.  With asm loads the compiler knows
// nothing about them, so EVERY consumer needs an explicit wait: stream_wait_vm<N>() followed by
// stream_tie() on the registers about to be read pins the order.  Rules that keep this sound (checked on
// the ISA by tools/check_stream_isa.py): destination registers stay integer vectors until they have
// landed (a cast in flight is real instructions), the loop has ONE exit and no load is left in flight
// into a register the compiler considers dead.
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int OFF>
__device__ __forceinline__ u32x4 stream_buffer_load(i32x4 rsrc, uint32_t voff) {
  u32x4 r;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(r) : "v"(voff), "s"(rsrc), "n"(OFF));
  return r;
}
// 16 bytes per lane from a uniform base (SGPR pair) + 32-bit lane offset + immediate
template <int OFF>
__device__ __forceinline__ u32x4 stream_load_b128_s(uint32_t voff, const void* sbase) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF));
  return r;
}
__device__ __forceinline__ int32_t stream_load_i32(const int32_t* p) {
  int32_t r;
  asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p));
  return r;
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// 64 lanes x 16 bytes, global -> LDS at lds_addr + 16 * lane (LDS base in M0), no register hop
__device__ __forceinline__ void stream_dma_b128(const void* src, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr)
               : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N>
__device__ __forceinline__ void stream_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
}
template <typename T>
__device__ __forceinline__ void stream_tie(T& v) {
  asm volatile("" : "+v"(v));
}
template <typename T>
__device__ __forceinline__ void stream_keep(const T& v) {
  asm volatile("" ::"v"(v));
}

}  // namespace
