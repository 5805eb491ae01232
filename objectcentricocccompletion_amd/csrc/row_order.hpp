// Neighbour-pattern buckets of the sub-manifold row order (sparse_conv_sorted.hip), shared with the geometry kernel
// that counts them while it writes the gather table (grid_geometry.hip).
#pragma once
#include "common.hpp"

namespace {

// buckets, heaviest class first: 3+ neighbours and 2 neighbours by their two lowest neighbour offsets a < b (triangular
// index b (b - 1) / 2 + a, 496 each), 1 neighbour by its offset (32), no neighbour (1).  The last 33 hold four rows out
// of five, and every counting workgroup adds to each of them; same-address atomics with a return take ~150 ns each, in a
// queue.  So the global counters keep them in kHotCopies copies side by side (copy = workgroup mod kHotCopies): a
// dozen atomics per counter and not 250 (750 from the emit kernel).  A workgroup's own (LDS) histogram needs no copies.
constexpr int kPairKeys = 32 * 31 / 2;
constexpr int kLocalBuckets = 2 * kPairKeys + 33;                // 1025: a workgroup's histogram
constexpr int kHotCopies = 64;
constexpr int kOrderBuckets = 2 * kPairKeys + 33 * kHotCopies;   // 3104: the global counters
constexpr int kOrderKeyBits = 12;                                // a row record's first word: bucket | place in the bucket << 12
constexpr int64_t kOrderMaxRows = 1ll << (32 - kOrderKeyBits);
static_assert(kOrderBuckets <= (1 << kOrderKeyBits), "bucket index must fit the key bits");

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

__device__ __forceinline__ uint32_t order_neighbours(uint32_t mask, int dense_k) {
  return dense_k >= 0 ? mask & ~(1u << dense_k) : mask;
}

__device__ __forceinline__ int order_key_local(uint32_t mask, int dense_k) {
  uint32_t m = order_neighbours(mask, dense_k);
  const int pc = __builtin_popcount(m);
  if (pc == 0) return 2 * kPairKeys + 32;
  const int a = __builtin_ctz(m);
  if (pc == 1) return 2 * kPairKeys + a;
  m &= m - 1;
  const int b = __builtin_ctz(m);
  return (pc == 2 ? kPairKeys : 0) + b * (b - 1) / 2 + a;
}
// a workgroup's bucket -> the global counter it adds to
__device__ __forceinline__ int order_global(int local, int copy) {
  return local < 2 * kPairKeys ? local : 2 * kPairKeys + (local - 2 * kPairKeys) * kHotCopies + copy;
}

// A counting workgroup's buckets -> its place in the global buckets: h[b] (rows of the workgroup in its bucket b) is
// replaced by the value the global counter had.  All atomics are in flight before the first result is used.
template <int THREADS>
__device__ __forceinline__ void order_reserve(uint32_t* h, uint32_t* __restrict__ hist, int copy) {
  constexpr int PER = (kLocalBuckets + THREADS - 1) / THREADS;
  uint32_t cnt[PER], got[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int b = threadIdx.x + THREADS * i;
    cnt[i] = b < kLocalBuckets ? h[b] : 0u;
    got[i] = 0u;
    if (cnt[i]) got[i] = atomicAdd(&hist[order_global(b, copy)], cnt[i]);
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int b = threadIdx.x + THREADS * i;
    if (cnt[i]) h[b] = got[i];
  }
}

}  // namespace
