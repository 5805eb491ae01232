// Device-wide exclusive prefix sums over uint32 (two launches -- three past 4096 blocks per row --
// no atomics, deterministic).  Used for (a) cell bitmaps -> voxel ranks (popcount scan) and
// (b) per-offset stream compaction of the rulebook.  Batched over `rows`
// independent rows of equal length.  HBM-bound: reads the input twice and
// writes it once.
#pragma once
#include "common.hpp"

namespace ococc_scan {

enum Transform { IDENT = 0, POPC = 1, NONNEG = 2 };

constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kTile = kThreads * kItems;  // 2048 elements per block

template <int T>
__device__ __forceinline__ uint32_t xform(uint32_t v) {
  if (T == POPC) return (uint32_t)__popc(v);
  if (T == NONNEG) return ((int32_t)v >= 0) ? 1u : 0u;
  return v;
}

// inclusive scan of one value per thread across a 256-thread block
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wave_tot[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  if (lane == 63) wave_tot[wave] = v;
  __syncthreads();
  uint32_t add = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) {
    uint32_t t = wave_tot[w];
    if (w < wave) add += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return v + add;
}

template <int T>
__global__ void __launch_bounds__(kThreads)
reduce_kernel(const uint32_t* __restrict__ in, int64_t n, int64_t row_stride,
              uint32_t* __restrict__ block_sums, int blocks_per_row) {
  const int row = blockIdx.y;
  const uint32_t* src = in + (int64_t)row * row_stride;
  const int64_t base = (int64_t)blockIdx.x * kTile;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    int64_t i = base + (int64_t)j * kThreads + threadIdx.x;
    if (i < n) s += xform<T>(src[i]);
  }
  uint32_t tot;
  block_inclusive_scan(s, &tot);
  if (threadIdx.x == 0) block_sums[(int64_t)row * blocks_per_row + blockIdx.x] = tot;
}

// one block per row: exclusive scan of the row's block sums in place; row total -> totals[row]
static __global__ void __launch_bounds__(kThreads)
scan_sums_kernel(uint32_t* __restrict__ block_sums, int blocks_per_row,
                 uint32_t* __restrict__ totals) {
  uint32_t* s = block_sums + (int64_t)blockIdx.x * blocks_per_row;
  uint32_t carry = 0;
  for (int base = 0; base < blocks_per_row; base += kThreads) {
    int i = base + threadIdx.x;
    uint32_t v = i < blocks_per_row ? s[i] : 0u;
    uint32_t tot;
    uint32_t inc = block_inclusive_scan(v, &tot);
    if (i < blocks_per_row) s[i] = carry + inc - v;
    carry += tot;
  }
  if (threadIdx.x == 0 && totals) totals[blockIdx.x] = carry;
}

// out[i] = exclusive prefix of xform(in) within the row.  Thread t owns kItems
// CONSECUTIVE elements so the in-thread running sum is the prefix.
// SELF_BASE: block_sums holds the RAW block totals of reduce_kernel and every block adds up the ones
// in front of it by itself (cheap while a row has few blocks; saves the scan_sums launch).  The row
// total goes to totals[row] from the row's last block.
template <int T, bool SELF_BASE>
__global__ void __launch_bounds__(kThreads)
apply_kernel(const uint32_t* __restrict__ in, int64_t n, int64_t row_stride,
             const uint32_t* __restrict__ block_sums, int blocks_per_row,
             uint32_t* __restrict__ out, int64_t out_row_stride, uint32_t* __restrict__ totals) {
  const int row = blockIdx.y;
  const uint32_t* src = in + (int64_t)row * row_stride;
  uint32_t* dst = out + (int64_t)row * out_row_stride;
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  uint32_t v[kItems];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    int64_t i = base + j;
    v[j] = i < n ? xform<T>(src[i]) : 0u;
    s += v[j];
  }
  uint32_t before;
  if (SELF_BASE) {
    uint32_t part = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += kThreads) part += block_sums[(int64_t)row * blocks_per_row + j];
    block_inclusive_scan(part, &before);
  } else {
    before = block_sums[(int64_t)row * blocks_per_row + blockIdx.x];
  }
  uint32_t tot;
  uint32_t inc = block_inclusive_scan(s, &tot);
  if (SELF_BASE && totals && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) totals[row] = before + tot;
  uint32_t run = before + inc - s;
#pragma unroll
  for (int j = 0; j < kItems; ++j) {
    int64_t i = base + j;
    if (i < n) dst[i] = run;
    run += v[j];
  }
}

static inline int blocks_for(int64_t n) { return (int)((n + kTile - 1) / kTile); }
// scratch (uint32 count) needed for `rows` rows of length n
static inline int64_t scratch_words(int64_t n, int rows) { return (int64_t)blocks_for(n) * rows; }

// Queue the launches.  totals may be NULL.  in/out may alias only if
// row strides are equal (each element is read before it is written by the
// same thread in apply_kernel, and reduce_kernel has finished by then).
template <int T>
static inline hipError_t exclusive_scan(const uint32_t* in, int64_t n, int64_t row_stride, int rows,
                                        uint32_t* out, int64_t out_row_stride, uint32_t* scratch,
                                        uint32_t* totals, hipStream_t stream) {
  if (n <= 0 || rows <= 0) return hipSuccess;
  const int nb = blocks_for(n);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(reduce_kernel<T>), dim3(nb, rows), dim3(kThreads), 0, stream,
                     in, n, row_stride, scratch, nb);
  if (nb <= 4096) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(apply_kernel<T, true>), dim3(nb, rows), dim3(kThreads), 0, stream,
                       in, n, row_stride, scratch, nb, out, out_row_stride, totals);
  } else {
    hipLaunchKernelGGL(scan_sums_kernel, dim3(rows), dim3(kThreads), 0, stream, scratch, nb, totals);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(apply_kernel<T, false>), dim3(nb, rows), dim3(kThreads), 0, stream,
                       in, n, row_stride, scratch, nb, out, out_row_stride, (uint32_t*)nullptr);
  }
  return hipGetLastError();
}

}  // namespace ococc_scan
