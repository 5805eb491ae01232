// Neighbour-pattern buckets of the sub-manifold row order (sparse_conv_sorted.hip), shared with the geometry kernel
// that counts them while it writes the gather table (grid_geometry.hip).
#pragma once
#include "common.hpp"

namespace {

// buckets, heaviest class first: 3+ neighbours and 2 neighbours by their two lowest neighbour offsets a < b (triangular
// index b (b - 1) / 2 + a, 496 each), 1 neighbour by its offset (32), no neighbour (1).  (One bucket for the whole 3+
// class was tried: nothing lost at the benchmark's density, where the class is 4 % of the rows and its tiles walk
// nearly every offset anyway, but 70 -> 84 us at 2.5 pairs per row, where it is a third of them.)  Every counting
// workgroup reserves its share of a bucket with ONE atomic that returns the counter's value; same-address atomics with
// a return take ~130 ns each, in a queue, and the vector-memory counter retires in order, so whatever the kernel waits
// for next waits for them too (grid_emit_kernel: 4 us, measured by compiling them out).  The global counters therefore
// exist in copies side by side (copy = workgroup mod copies): kHotCopies for the 33 buckets every workgroup adds to (1
// and 0 neighbours: 82 % of the rows), kMidCopies / kHeavyCopies for the pair buckets (~50 + ~20 of them per workgroup
// of 500 rows: 40 atomics per counter without copies).  A workgroup's own (LDS) histogram needs no copies.
constexpr int kPairKeys = 32 * 31 / 2;
constexpr int kLocalMid = kPairKeys;                             // first 2-neighbour bucket of a workgroup's histogram
constexpr int kLocalHot = 2 * kPairKeys;                         // first 1-neighbour bucket
constexpr int kLocalBuckets = 2 * kPairKeys + 33;                // 1025
constexpr int kHotCopies = 32;
constexpr int kMidCopies = 4;
constexpr int kHeavyCopies = 2;
constexpr int kGlobalMid = kPairKeys * kHeavyCopies;             // end of the 3+ class in the global counters
constexpr int kGlobalHot = kGlobalMid + kPairKeys * kMidCopies;  // end of the 2-neighbour class
constexpr int kOrderBuckets = 4096;                              // the global counters: 4032 used (whole counters per scanning thread)
constexpr int kOrderKeyBits = 12;                                // a row record's first word: bucket | place in the bucket << 12
constexpr int64_t kOrderMaxRows = 1ll << (32 - kOrderKeyBits);
static_assert(kGlobalHot + 33 * kHotCopies <= kOrderBuckets && kOrderBuckets <= (1 << kOrderKeyBits), "bucket index must fit the key bits");

constexpr int kOrderCounterWords = kOrderBuckets;

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

// header the order carries (int32[8]): blocks of 16 slots up to which tiles have heavy_blocks / mid_blocks blocks (16
// after), total blocks, tiles
struct OrderHdr {
  int b_heavy, b_mid, b_total, tiles, n, heavy_blocks, mid_blocks, dense_k;
};

__device__ __forceinline__ uint32_t order_neighbours(uint32_t mask, int dense_k) {
  return dense_k >= 0 ? mask & ~(1u << dense_k) : mask;
}

__device__ __forceinline__ int order_key_local(uint32_t mask, int dense_k) {
  uint32_t m = order_neighbours(mask, dense_k);
  const int pc = __builtin_popcount(m);
  if (pc == 0) return kLocalHot + 32;
  const int a = __builtin_ctz(m);
  if (pc == 1) return kLocalHot + a;
  m &= m - 1;
  const int b = __builtin_ctz(m);
  return (pc == 2 ? kLocalMid : 0) + b * (b - 1) / 2 + a;
}
// a workgroup's bucket -> the global counter workgroup ``wg`` adds to, and back
__device__ __forceinline__ int order_global(int local, int wg) {
  if (local < kLocalMid) return local * kHeavyCopies + wg % kHeavyCopies;
  if (local < kLocalHot) return kGlobalMid + (local - kLocalMid) * kMidCopies + wg % kMidCopies;
  return kGlobalHot + (local - kLocalHot) * kHotCopies + wg % kHotCopies;
}
__device__ __forceinline__ int order_local(int global) {
  if (global < kGlobalMid) return global / kHeavyCopies;
  if (global < kGlobalHot) return kLocalMid + (global - kGlobalMid) / kMidCopies;
  return kLocalHot + (global - kGlobalHot) / kHotCopies;
}

// A counting workgroup's buckets -> its place in the global buckets: h[b] (rows of the workgroup in its bucket b) is
// replaced by the value the global counter had.  All atomics are in flight before the first result is used.
template <int THREADS>
__device__ __forceinline__ void order_reserve(uint32_t* h, uint32_t* __restrict__ hist, int wg) {
  constexpr int PER = (kLocalBuckets + THREADS - 1) / THREADS;
  uint32_t cnt[PER], got[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int b = threadIdx.x + THREADS * i;
    cnt[i] = b < kLocalBuckets ? h[b] : 0u;
    got[i] = 0u;
    if (cnt[i]) got[i] = atomicAdd(&hist[order_global(b, wg)], cnt[i]);
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int b = threadIdx.x + THREADS * i;
    if (cnt[i]) h[b] = got[i];
  }
}

// First slot of every global bucket: exclusive prefix of the counters into ``start[kOrderBuckets + 1]`` (LDS), every
// workgroup for itself -- no scan launch.  ``part``: THREADS / 64 words of LDS.  Ends with a barrier.
template <int THREADS, class Load>
__device__ __forceinline__ void order_scan_starts(Load load, uint32_t* start, uint32_t* part) {
  static_assert(kOrderBuckets % THREADS == 0, "whole buckets per thread");
  constexpr int PER = kOrderBuckets / THREADS;
  uint32_t v[PER], s = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    v[i] = load((int)threadIdx.x * PER + i);
    s += v[i];
  }
  // exclusive prefix over the threads: inside the wave with shuffles, across the waves through LDS
  uint32_t inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if ((threadIdx.x & 63) >= d) inc += t;
  }
  if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t run = inc - s;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += part[w];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    start[threadIdx.x * PER + i] = run;
    run += v[i];
  }
  if (threadIdx.x == THREADS - 1) start[kOrderBuckets] = run;
  __syncthreads();
}

__device__ __forceinline__ void order_write_hdr(const uint32_t* start, int64_t n, int heavy_blocks, int mid_blocks, int dense_k,
                                                OrderHdr* hdr) {
  const int e3 = (int)start[kGlobalMid], e2 = (int)start[kGlobalHot];   // ends of the 3+ and of the 2 neighbour class
  OrderHdr o;
  o.b_total = (int)((n + 15) >> 4);
  o.b_heavy = min((e3 + 15) >> 4, o.b_total);
  o.b_mid = min(max(o.b_heavy, (e2 + 15) >> 4), o.b_total);
  o.tiles = (o.b_heavy + heavy_blocks - 1) / heavy_blocks + (o.b_mid - o.b_heavy + mid_blocks - 1) / mid_blocks +
            (o.b_total - o.b_mid + 15) / 16;
  o.n = (int)n;
  o.heavy_blocks = heavy_blocks;
  o.mid_blocks = mid_blocks;
  o.dense_k = dense_k;
  *hdr = o;
}

}  // namespace
