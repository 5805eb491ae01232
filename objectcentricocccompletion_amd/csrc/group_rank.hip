// B6 in-group ranks: get_inner_win_inds / make_continuous_inds of the SST window machinery
// (mmdet3d/ops/sst/sst_ops.py:243-263 -> TorchEx ingroup_indices, source not vendored; python
// twin get_inner_win_inds_deprecated :194-241; make_continuous_inds :316-330).
// For keys[i] in [0, key_bound):
//   conti[i] = rank of keys[i] among the distinct keys (sorted)      == make_continuous_inds
//   inner[i] = number of j < i with keys[j] == keys[i]               == a valid ingroup index
//   counts[g] = size of group g, num_groups = number of distinct keys
// The reference accepts any order inside a group; ours is the stable one (by element index), so
// results are reproducible.  Same bitmap + popcount-prefix machinery as the voxel code: no sort.
// Window populations are bounded by the window volume (<= 512 for 8^3 windows), so the
// "count smaller members" pass is cheap.
#include "common.hpp"
#include "scan.hpp"

namespace {

__global__ void __launch_bounds__(256)
gr_mark_kernel(const int32_t* __restrict__ keys, int64_t n, int32_t bound, uint32_t* __restrict__ bitmap,
               int32_t* __restrict__ status) {
  // (whole waves walk the keys, so that neighbouring lanes can compare notes: voxels arrive in coordinate order and
  // the lanes of a wave name a handful of windows -- only the first lane of a run of equal keys marks it, csrc/grid_unique.hip)
  const int64_t n_up = (n + 63) & ~(int64_t)63;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_up; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t k = i < n ? keys[i] : -1;
    if (k >= bound) {
      *status = 1;
      k = -1;
    }
    const int32_t left = __shfl_up(k, 1, 64);
    if (k < 0 || ((threadIdx.x & 63) != 0 && left == k)) continue;
    // (most keys of a group find their bit set already: a plain read first keeps ~30 atomics per word off the L2 --
    // 40 -> a few us for 260 k window ids; a stale read only costs the atomic it would have saved)
    const uint32_t bit = 1u << (k & 31);
    if (!(__builtin_nontemporal_load(bitmap + (k >> 5)) & bit)) atomicOr(bitmap + (k >> 5), bit);
  }
}

// The same for key ranges whose bitmap fits a workgroup's LDS (window ids of a batch of grids: a few thousand words):
// every workgroup marks a CONTIGUOUS stretch of the keys in its own LDS bitmap and hands over only the words it
// touched.  Voxels arrive in coordinate order, a stretch names a handful of words: ~10^3 global atomics for 260 k
// window ids instead of one per run of equal keys on a few hundred words (same-address atomics queue: 41 us).
constexpr int kMarkLdsWords = 8192;   // 262 k keys: three drop levels of 35 k window ids and more
__global__ void __launch_bounds__(256)
gr_mark_lds_kernel(const int32_t* __restrict__ keys, int64_t n, int32_t bound, int64_t per_block,
                   uint32_t* __restrict__ bitmap, int32_t* __restrict__ status) {
  __shared__ uint32_t bits[kMarkLdsWords];
  const int words = (bound + 31) >> 5;
  for (int w = threadIdx.x; w < words; w += 256) bits[w] = 0u;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * per_block, hi = min(n, lo + per_block);
  bool bad = false;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const int32_t k = keys[i];
    if (k >= bound) bad = true;
    else if (k >= 0) atomicOr(&bits[k >> 5], 1u << (k & 31));
  }
  if (bad) *status = 1;
  __syncthreads();
  for (int w = threadIdx.x; w < words; w += 256) {
    const uint32_t v = bits[w];
    if (v && (__builtin_nontemporal_load(bitmap + w) & v) != v) atomicOr(bitmap + w, v);
  }
}

__global__ void __launch_bounds__(256)
gr_rank_count_kernel(const int32_t* __restrict__ keys, int64_t n, int32_t bound,
                     const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ prefix,
                     int32_t* __restrict__ conti, uint32_t* __restrict__ counts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t k = keys[i];
    int32_t r = -1;
    if (k >= 0 && k < bound) {
      const uint32_t w = bitmap[k >> 5];
      r = (int32_t)(prefix[k >> 5] + __popc(w & ((1u << (k & 31)) - 1u)));
      atomicAdd(counts + r, 1u);
    }
    conti[i] = r;
  }
}

__global__ void __launch_bounds__(256)
gr_members_kernel(const int32_t* __restrict__ conti, int64_t n, const uint32_t* __restrict__ offsets,
                  uint32_t* __restrict__ cursor, int32_t* __restrict__ members) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = conti[i];
    if (r >= 0) members[offsets[r] + atomicAdd(cursor + r, 1u)] = (int32_t)i;
  }
}

__global__ void __launch_bounds__(256)
gr_inner_kernel(const int32_t* __restrict__ conti, int64_t n, const uint32_t* __restrict__ offsets,
                const uint32_t* __restrict__ counts, const int32_t* __restrict__ members,
                int32_t* __restrict__ inner) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = conti[i];
    int32_t rank = -1;
    if (r >= 0) {
      const int32_t* m = members + offsets[r];
      const uint32_t c = counts[r];
      rank = 0;
      for (uint32_t j = 0; j < c; ++j) rank += m[j] < (int32_t)i ? 1 : 0;
    }
    inner[i] = rank;
  }
}

struct Layout {
  int64_t words, o_bitmap, o_prefix, o_counts, o_off, o_cursor, o_members, o_scratch, total;
};
inline bool make_layout(int64_t n, int64_t bound, Layout* L) {
  if (n < 0 || bound < 1 || bound > 0x7fffffffLL) return false;
  L->words = (bound + 31) / 32;
  int64_t off = 0;
  auto take = [&](int64_t b) { int64_t o = off; off += ococc_align_up(b > 0 ? b : 4, 256); return o; };
  // (bitmap | counts | cursor are what has to start at zero: contiguous, one memset)
  L->o_bitmap = take(L->words * 4);
  L->o_counts = take(n * 4);
  L->o_cursor = take(n * 4);
  L->o_prefix = take(L->words * 4);
  L->o_off = take(n * 4);
  L->o_members = take(n * 4);
  const int64_t s1 = ococc_scan::scratch_words(L->words, 1), s2 = ococc_scan::scratch_words(n > 0 ? n : 1, 1);
  L->o_scratch = take((s1 > s2 ? s1 : s2) * 4);
  L->total = off;
  return true;
}

}  // namespace

extern "C" int64_t ococc_group_rank_workspace_bytes(int64_t n, int64_t key_bound) {
  Layout L;
  if (!make_layout(n, key_bound, &L)) return -1;
  return L.total;
}

extern "C" int ococc_group_rank_i32(const int32_t* keys, int64_t n, int64_t key_bound, int32_t* conti,
                                    int32_t* inner, int32_t* counts, int32_t* num_groups, int32_t* status,
                                    void* workspace, int64_t workspace_bytes, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Layout L;
  OCOCC_REQUIRE(make_layout(n, key_bound, &L), "need 1 <= key_bound < 2^31");
  OCOCC_REQUIRE(num_groups && status, "null num_groups/status");
  if (status == num_groups + 1) {
    OCOCC_HIP(hipMemsetAsync(num_groups, 0, 8, stream));   // (the usual {num_groups, status} pair: one memset)
  } else {
    OCOCC_HIP(hipMemsetAsync(num_groups, 0, 4, stream));
    OCOCC_HIP(hipMemsetAsync(status, 0, 4, stream));
  }
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(keys && conti && inner, "null pointer");
  OCOCC_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small");
  char* ws = (char*)workspace;
  uint32_t* bitmap = (uint32_t*)(ws + L.o_bitmap);
  uint32_t* prefix = (uint32_t*)(ws + L.o_prefix);
  uint32_t* cnt = (uint32_t*)(ws + L.o_counts);
  uint32_t* offsets = (uint32_t*)(ws + L.o_off);
  uint32_t* cursor = (uint32_t*)(ws + L.o_cursor);
  int32_t* members = (int32_t*)(ws + L.o_members);
  uint32_t* scratch = (uint32_t*)(ws + L.o_scratch);
  const int g1 = ococc_grid_1d(n, 256);
  OCOCC_HIP(hipMemsetAsync(bitmap, 0, (size_t)(L.o_prefix - L.o_bitmap), stream));   // bitmap, counts, cursor
  if (key_bound <= (int64_t)kMarkLdsWords * 32) {
    const int64_t per = ococc_align_up(ococc_cdiv(n, 1024), 256);   // <= 1024 workgroups, whole passes of 256 keys
    hipLaunchKernelGGL(gr_mark_lds_kernel, dim3((unsigned)ococc_cdiv(n, per)), dim3(256), 0, stream, keys, n,
                       (int32_t)key_bound, per, bitmap, status);
  } else {
    hipLaunchKernelGGL(gr_mark_kernel, dim3(g1), dim3(256), 0, stream, keys, n, (int32_t)key_bound, bitmap, status);
  }
  OCOCC_CHECK_LAUNCH();
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::POPC>(bitmap, L.words, L.words, 1, prefix, L.words, scratch,
                                                         (uint32_t*)num_groups, stream));
  hipLaunchKernelGGL(gr_rank_count_kernel, dim3(g1), dim3(256), 0, stream, keys, n, (int32_t)key_bound, bitmap,
                     prefix, conti, cnt);
  OCOCC_CHECK_LAUNCH();
  OCOCC_HIP(ococc_scan::exclusive_scan<ococc_scan::IDENT>(cnt, n, n, 1, offsets, n, scratch, nullptr, stream));
  hipLaunchKernelGGL(gr_members_kernel, dim3(g1), dim3(256), 0, stream, conti, n, offsets, cursor, members);
  OCOCC_CHECK_LAUNCH();
  hipLaunchKernelGGL(gr_inner_kernel, dim3(g1), dim3(256), 0, stream, conti, n, offsets, cnt, members, inner);
  OCOCC_CHECK_LAUNCH();
  if (counts) OCOCC_HIP(hipMemcpyAsync(counts, cnt, n * 4, hipMemcpyDeviceToDevice, stream));
  return OCOCC_OK;
}
