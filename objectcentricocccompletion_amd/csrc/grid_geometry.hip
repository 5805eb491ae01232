// Geometry of a batch of per-object grids in three launches: what ococc_voxelize_scatter_mean_f32 (7 launches)
// followed by ococc_subm_rulebook_build_sorted (3 launches) produce, for the case the occupancy encoder lives in:
// points that arrive GROUPED BY GRID (batch index non-decreasing), fixed-capacity ("static") outputs, 3x3x3
// sub-manifold rulebook.
//
// Reference chain restated (bit for bit the same outputs as the two entry points above, which are pinned to it):
//   dynamic_voxelize            mmdet3d/ops/voxel/src/voxelization_cpu.cpp:8-41
//   DynamicScatter(mean)        mmdet3d/ops/voxel/src/scatter_points_cuda.cu:183-241
//   getIndicePairsSubM          mmdet3d/ops/spconv/include/spconv/geometry.h:247-297
//
// MI355X design.  One object grid is small: 40^3 cells = 8 KB of occupancy bits.  The general path keeps ONE
// bitmap for the whole batch in global memory and pays for it in scattered global atomics (marking), a
// device-wide scan, and 27 L2 probes per voxel (neighbour table): ten launches of 5-12 us each whose kernels
// wait on memory-side atomics and dependent L2 round trips.  Here a workgroup owns a grid and keeps its bitmap
// and popcount prefix in LDS:
//   A  grid_mark_count_kernel   one workgroup per grid: find the grid's point segment (two rounds of 1024 probes of
//                               batch_idx), mark cells with LDS atomics (the returned old word says whether a point
//                               is its cell's first arrival), popcount-scan the words in LDS, count -- per row slice
//                               and per kernel offset -- the voxels and their neighbours, 32 cells at a time: the
//                               bitmap ANDed with itself shifted by the offset (one v_alignbit) and with the masks
//                               of the cells whose neighbour would fall off the x / y edge;
//                               writes the bitmap, its local prefix, the per-point codes and a [grids x slices x 28]
//                               count table
//   S  geometry_bases_kernel    exclusive prefix of that table (28 columns): global row bases and pair bases
//   B  grid_emit_kernel         one workgroup per (grid, row slice) with the grid's bitmap back in LDS: voxel
//                               coordinates, the global prefix words, the means of the point features (first arrival
//                               plain-stores, later arrivals add with float atomics behind a workgroup barrier, the
//                               owner of a shared row divides), the offset-major neighbour table, the 16-row block
//                               masks, the reference-format pair lists in CPU-functor order, and the -1 / zero
//                               padding rows of the fixed-capacity form.
// Nothing leaves the chip between the phases of a kernel except the outputs; the only global atomics are the float
// adds of the 1.6 % of points that share a cell and the 32-bit ORs of the block masks.
#include "common.hpp"
#include "row_order.hpp"

namespace {

#ifdef OCOCC_GEO_STAMPS
// diagnostic build only (tools/probe/geo_stamps.py): wall-clock stamps per workgroup and phase into a buffer of their own
__device__ long long* g_stamps = nullptr;
#define STAMP(slot) do { if (threadIdx.x == 0 && g_stamps) g_stamps[(int64_t)blockIdx.x * 16 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

constexpr int kThreads = 1024;       // kernel A: one workgroup per grid
constexpr int kWaves = kThreads / 64;
constexpr int kEmitThreads = 256;    // kernel B: one workgroup per (grid, row slice), several per CU
constexpr int kEmitWaves = kEmitThreads / 64;
constexpr int kFusedBasesMaxEntries = 256;   // kernel-A workgroups up to which kernel B sums their count rows itself
#ifndef OCOCC_GEO_FUSED_BASES
#define OCOCC_GEO_FUSED_BASES 1
#endif
constexpr int kPadBlocks = 64;       // workgroups of kernel B that write the padding rows of the fixed-capacity form
constexpr int kCols = 28;            // 27 kernel offsets + the voxel count
constexpr int kMaxSlices = 16;
constexpr int kCodesPerThread = 8;   // point codes a thread of kernel B fetches in one round

// workgroup barrier for data exchanged through LDS only: does not wait for the wave's global stores to land
// (__syncthreads() does, with s_waitcnt vmcnt(0), which is what the places that hand GLOBAL data on need)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct GeoParams {
  float vx, vy, vz, xmin, ymin, zmin;
  int32_t gx, gy, gz, batch;
  int32_t words;   // bitmap words per grid (cells / 32)
  int32_t slices;  // row slices per grid (workgroups of kernel B per grid)
  int32_t wps;     // words per slice (a multiple of 64)
  int32_t asplit;  // workgroups of kernel A per grid; each counts for spa = ceil(slices / asplit) consecutive slices
  int32_t spa;
};

__device__ __forceinline__ int32_t lds_rank(const uint32_t* bm, const uint32_t* pf, int32_t cell) {
  const uint32_t w = bm[cell >> 5];
  const uint32_t bit = 1u << (cell & 31);
  return (w & bit) ? (int32_t)(pf[cell >> 5] + __popc(w & (bit - 1u))) : -1;
}

// exclusive popcount scan of `words` bitmap words held in LDS.  tmp: kWaves words
__device__ void lds_popc_scan(const uint32_t* bm, uint32_t* pf, int words, uint32_t* tmp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (int w0 = 0; w0 < words; w0 += kThreads) {
    const int w = w0 + threadIdx.x;
    const uint32_t v = w < words ? (uint32_t)__popc(bm[w]) : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int t = 0; t < kWaves; ++t) {
      const uint32_t s = tmp[t];
      if (t < wave) before += s;
      total += s;
    }
    if (w < words) pf[w] = carry + before + inc - v;
    carry += total;
    __syncthreads();
  }
}

// lo = first i with batch_idx[i] >= key, hi = first i with batch_idx[i] >= key + 1 (n if none), batch_idx
// non-decreasing: both searched together, a coarse round over kThreads chunks and a fine round inside the chunk
__device__ void segment_bounds(const int32_t* __restrict__ batch_idx, int64_t n, int32_t key, int64_t* s_two,
                               int64_t* lo_out, int64_t* hi_out) {
  const int64_t step = (n + kThreads - 1) / kThreads;
  if (threadIdx.x < 2) s_two[threadIdx.x] = n;
  __syncthreads();
  {
    const int64_t c0 = (int64_t)threadIdx.x * step;
    if (c0 < n) {
      const int32_t last = batch_idx[(c0 + step < n ? c0 + step : n) - 1];
      const int32_t prev = c0 > 0 ? batch_idx[c0 - 1] : INT32_MIN;
      // the chunk whose last element is the first one >= key' holds the boundary for key'
      if (last >= key && prev < key) s_two[0] = c0;
      if (last >= key + 1 && prev < key + 1) s_two[1] = c0;
    }
  }
  __syncthreads();
  const int64_t b0 = s_two[0], b1 = s_two[1];
  __syncthreads();
  if (threadIdx.x < 2) s_two[threadIdx.x] = n;
  __syncthreads();
  for (int64_t t = threadIdx.x; t < 2 * step; t += kThreads) {
    const int which = t >= step;
    const int64_t base = which ? b1 : b0;
    const int64_t i = base + (which ? t - step : t);
    if (base < n && i < n) {
      const int32_t kk = key + which;
      if (batch_idx[i] >= kk && (i == 0 || batch_idx[i - 1] < kk)) s_two[which] = i;
    }
  }
  __syncthreads();
  *lo_out = s_two[0];
  *hi_out = s_two[1];
  __syncthreads();
}

// ---- A: one workgroup per grid ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
grid_mark_count_kernel(const float* __restrict__ points, int nfeat, const int32_t* __restrict__ batch_idx, int64_t n,
                       GeoParams g, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ local_prefix,
                       int32_t* __restrict__ code_of, int32_t* __restrict__ table, int64_t* __restrict__ seg,
                       uint32_t* __restrict__ blockmask, int64_t mask_words, int32_t* __restrict__ inv,
                       int32_t* __restrict__ bad_flags, uint32_t* __restrict__ order_hist, int32_t* __restrict__ part_sums) {
  extern __shared__ uint32_t smem[];
  uint32_t* bm = smem;             // [words]
  uint32_t* pf = smem + g.words;   // [words]
  __shared__ uint32_t s_scan[kWaves];
  __shared__ int32_t s_cnt[kMaxSlices][kCols][16];  // per slice of this workgroup; 16 copies per counter: a wave's lanes spread over them
  __shared__ int64_t s_two[2];
  __shared__ int s_bad;
  // `asplit` workgroups per grid: each builds the grid's whole bitmap (cheap: LDS atomics) and its prefix, and then
  // counts and writes out its own slice of the words only; slice 0 alone decides which points arrived first
  const int b = blockIdx.x / g.asplit, my_sl = blockIdx.x % g.asplit;  // my_sl: which share of the grid's slices

  STAMP(0);
  for (int w = threadIdx.x; w < g.words; w += kThreads) bm[w] = 0u;
  for (int t = threadIdx.x; t < kMaxSlices * kCols * 16; t += kThreads) (&s_cnt[0][0][0])[t] = 0;
  if (threadIdx.x == 0) s_bad = 0;
  // the block masks are OR-ed by kernel B: cleared here, one slice per workgroup
  for (int64_t w = (int64_t)blockIdx.x * kThreads + threadIdx.x; w < mask_words; w += (int64_t)gridDim.x * kThreads) blockmask[w] = 0u;
  // ... and so are the bucket counters of the row order kernel B counts into
  if (order_hist && blockIdx.x == 0)
    for (int w = threadIdx.x; w < kOrderCounterWords; w += kThreads) order_hist[w] = 0u;
  int64_t lo, hi;
  segment_bounds(batch_idx, n, b, s_two, &lo, &hi);
  if (threadIdx.x == 0 && my_sl == 0) {
    seg[2 * b] = lo;
    seg[2 * b + 1] = hi;
  }
  // points in front of grid 0 (negative batch index) and behind the last grid (index >= batch): dropped
  if (b == 0 && my_sl == 0)
    for (int64_t i = threadIdx.x; i < lo; i += kThreads) inv[i] = -1;
  if (b == g.batch - 1 && my_sl == 0)
    for (int64_t i = hi + threadIdx.x; i < n; i += kThreads) {
      inv[i] = -1;
      s_bad = 1;  // benign race
    }
  __syncthreads();
  STAMP(1);

  const int32_t cells = g.words * 32;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
    const float* p = points + i * nfeat;
    // voxelization_cpu.cpp:8-41: floor((p - min) / voxel) in float, clamped to the grid
    int cx = (int)floorf((p[0] - g.xmin) / g.vx);
    int cy = (int)floorf((p[1] - g.ymin) / g.vy);
    int cz = (int)floorf((p[2] - g.zmin) / g.vz);
    cx = cx < 0 ? 0 : (cx >= g.gx ? g.gx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= g.gy ? g.gy - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= g.gz ? g.gz - 1 : cz);
    int32_t code = -2;
    if (batch_idx[i] == b) {  // (anything else: the batch indices are not sorted)
      const int32_t cell = (cz * g.gy + cy) * g.gx + cx;
      const uint32_t bit = 1u << (cell & 31);
      const uint32_t old = atomicOr(bm + (cell >> 5), bit);
      code = (b * cells + cell) * 2 + ((old & bit) ? 1 : 0);  // same code as voxel_mark_kernel: 2 * global cell + dup
    } else {
      s_bad = 1;
    }
    if (my_sl == 0) code_of[i] = code;
  }
  __syncthreads();
  STAMP(2);
  lds_popc_scan(bm, pf, g.words, s_scan);
  const int aw_lo = my_sl * g.spa * g.wps < g.words ? my_sl * g.spa * g.wps : g.words;
  const int aw_hi = aw_lo + g.spa * g.wps < g.words ? aw_lo + g.spa * g.wps : g.words;
  for (int w = aw_lo + threadIdx.x; w < aw_hi; w += kThreads) {
    bitmap[(int64_t)b * g.words + w] = bm[w];
    local_prefix[(int64_t)b * g.words + w] = pf[w];
  }
  STAMP(3);

  // counts per row slice: voxels, and for every kernel offset the voxels that have a neighbour there -- 32 cells at a
  // time.  Neighbour (dz,dy,dx) of cell c is cell c + off, off = (dz gy + dy) gx + dx, PROVIDED x + dx and y + dy stay
  // inside the grid (z takes care of itself: such a cell index falls outside [0, cells) and reads as empty).  So
  //   count(off) = popcount( bits(w) & inside_x(dx)(w) & inside_y(dy)(w) & (bits shifted down by off)(w) )  over the words
  const int plane = g.gx * g.gy;
  for (int w = aw_lo + threadIdx.x; w < aw_hi; w += kThreads) {
    const uint32_t bits = bm[w];
    int32_t* cnt = &s_cnt[(w - aw_lo) / g.wps][0][threadIdx.x & 15];
    if (!bits) continue;
    atomicAdd(cnt + 27 * 16, __popc(bits));
    // cells of this word whose x-1 / x+1 / y-1 / y+1 neighbour exists: runs of constant y inside the word
    uint32_t mxm = 0xffffffffu, mxp = 0xffffffffu, mym = 0u, myp = 0u;
    {
      const int c0 = w * 32;
      int xx = c0 % g.gx, yy = (c0 / g.gx) % g.gy;
      for (int pos = 0; pos < 32;) {
        const int len = (g.gx - xx < 32 - pos) ? g.gx - xx : 32 - pos;
        const uint32_t run = (len >= 32 ? 0xffffffffu : ((1u << len) - 1u)) << pos;
        if (yy != 0) mym |= run;
        if (yy != g.gy - 1) myp |= run;
        if (xx == 0) mxm &= ~(1u << pos);
        if (xx + len == g.gx) mxp &= ~(1u << (pos + len - 1));
        pos += len;
        xx = 0;
        yy = yy + 1 == g.gy ? 0 : yy + 1;
      }
    }
    // (rolled over the nine (dz, dy): these kernels run their code once per workgroup, so every instruction is an
    //  instruction-cache miss the first time -- a compact loop body beats 27 unrolled copies)
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int kz = t9 / 3, ky = t9 - kz * 3;
      uint32_t my = bits;
      if (ky == 0) my &= mym;
      if (ky == 2) my &= myp;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int off = (kz - 1) * plane + (ky - 1) * g.gx + (kx - 1);
        const int q = w + (off >> 5), r = off & 31;  // (arithmetic shift: floor division)
        const uint32_t lo32 = (unsigned)q < (unsigned)g.words ? bm[q] : 0u;
        const uint32_t hi32 = (r && (unsigned)(q + 1) < (unsigned)g.words) ? bm[q + 1] : 0u;
        uint32_t nb = __builtin_amdgcn_alignbit(hi32, lo32, (uint32_t)r) & my;  // bit i = bitmap bit (32 w + i + off)
        if (kx == 0) nb &= mxm;
        if (kx == 2) nb &= mxp;
        if (nb) atomicAdd(cnt + (t9 * 3 + kx) * 16, __popc(nb));
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < g.spa * kCols; t += kThreads) {
    const int ls = t / kCols, k = t % kCols, sl = my_sl * g.spa + ls;
    if (sl >= g.slices) continue;
    int32_t tot = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += s_cnt[ls][k][j];
    table[((int64_t)b * g.slices + sl) * kCols + k] = tot;
  }
  if (part_sums && threadIdx.x < kCols) {   // this workgroup's slices together: what kernel B sums instead of single slices
    int32_t tot = 0;
    for (int ls = 0; ls < g.spa && my_sl * g.spa + ls < g.slices; ++ls)
#pragma unroll
      for (int j = 0; j < 16; ++j) tot += s_cnt[ls][threadIdx.x][j];
    part_sums[(int64_t)blockIdx.x * kCols + threadIdx.x] = tot;
  }
  if (threadIdx.x == 0) bad_flags[blockIdx.x] = s_bad;  // (gathered into *status by the next launch: no memset)
  STAMP(4);
}

// ---- S: exclusive prefix of the count table: one 64-lane workgroup per column -------------------------------------
__global__ void __launch_bounds__(64)
geometry_bases_kernel(const int32_t* __restrict__ table, int64_t entries, int32_t* __restrict__ bases,
                      int32_t* __restrict__ totals, int32_t* __restrict__ indice_num, int32_t* __restrict__ num_voxels,
                      int64_t cap, const int32_t* __restrict__ bad_flags, int nflags, int32_t* __restrict__ status) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c == 27) {  // status = any workgroup of the first launch saw a point outside its grid's segment
    int bad = 0;
    for (int i = lane; i < nflags; i += 64) bad |= bad_flags[i];
    const unsigned long long m = __ballot(bad != 0);
    if (lane == 0) *status = m ? 1 : 0;
  }
  int32_t carry = 0;
  for (int64_t e0 = 0; e0 < entries; e0 += 64 * 8) {  // 8 entries per lane and round: their loads are independent
    int32_t v[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t e = e0 + (int64_t)lane * 8 + j;
      v[j] = e < entries ? table[e * kCols + c] : 0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
    int32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    int32_t run = carry + inc - sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t e = e0 + (int64_t)lane * 8 + j;
      if (e < entries) bases[e * kCols + c] = run;
      run += v[j];
    }
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) {
    totals[c] = carry;
    // indice_num[k] = pairs of offset k = entries of table column 26 - k (= column k, by symmetry); geometry.h order
    if (c < 27) indice_num[26 - c] = carry;
    else *num_voxels = (int32_t)(carry < cap ? carry : cap);
  }
}

__device__ __forceinline__ void put_feat_row(const float* __restrict__ src, float* __restrict__ dst_f32,
                                             uint16_t* __restrict__ dst_bf16, int c, float scale) {
  if ((c & 3) == 0) {
    for (int c0 = 0; c0 < c; c0 += 4) {
      float4 v = *(const float4*)(src + c0);
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
      *(float4*)(dst_f32 + c0) = v;
      if (dst_bf16) {
        uint2 q;
        q.x = (uint32_t)ococc_f32_to_bf16(v.x) | ((uint32_t)ococc_f32_to_bf16(v.y) << 16);
        q.y = (uint32_t)ococc_f32_to_bf16(v.z) | ((uint32_t)ococc_f32_to_bf16(v.w) << 16);
        *(uint2*)(dst_bf16 + c0) = q;
      }
    }
  } else {
    for (int ch = 0; ch < c; ++ch) {
      const float v = src[ch] * scale;
      dst_f32[ch] = v;
      if (dst_bf16) dst_bf16[ch] = ococc_f32_to_bf16(v);
    }
  }
}

// ---- B: one workgroup per (grid, row slice); the workgroups behind them write the padding rows -------------------
__global__ void __launch_bounds__(kEmitThreads)
grid_emit_kernel(const float* __restrict__ feats, int c, int64_t n, GeoParams g, const uint32_t* __restrict__ bitmap,
                 const uint32_t* __restrict__ local_prefix, uint32_t* __restrict__ prefix_out,
                 int32_t* __restrict__ code_of, const int64_t* __restrict__ seg, const int32_t* __restrict__ bases,
                 const int32_t* __restrict__ totals, int32_t* __restrict__ inv, int32_t* __restrict__ out_coors,
                 int32_t* __restrict__ counts, float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16, int64_t cap,
                 int32_t* __restrict__ nbr_t, uint32_t* __restrict__ blockmask, int32_t* __restrict__ pairs,
                 int emit_blocks, uint32_t* __restrict__ order_hist, i32x4_t* __restrict__ order_rowrec,
                 const int32_t* __restrict__ count_table, const int32_t* __restrict__ part_sums, int32_t* __restrict__ indice_num,
                 int32_t* __restrict__ num_voxels,
                 const int32_t* __restrict__ bad_flags, int nflags, int32_t* __restrict__ status) {
  extern __shared__ uint32_t smem[];
  // (part_sums != null: no geometry_bases_kernel ran -- few enough kernel-A workgroups that every workgroup here sums
  // their count rows in front of its own, plus the single slices of its own part, and workgroup 0 writes what that
  // kernel's lane 0s did)
  __shared__ int32_t s_part[3][kEmitWaves][32];
  // (order_hist != null: the rows' neighbour-pattern records of ococc_subm_row_order are written here, where the row's
  // 27 table entries sit in registers anyway -- the separate counting pass re-read the whole table, 14 us)
  __shared__ uint32_t s_oh[kLocalBuckets];
  if ((int)blockIdx.x >= emit_blocks) {
    // padding rows of the fixed-capacity form: -1 coordinates, zero count and features, no neighbours; the padding
    // workgroups take them in turns of 256
    int64_t total;
    if (part_sums) {   // rows of all grids: the voxel column of the part sums
      int32_t v = 0;
      for (int e = threadIdx.x; e < g.batch * g.asplit; e += kEmitThreads) v += part_sums[(int64_t)e * kCols + 27];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
      if ((threadIdx.x & 63) == 0) s_part[0][threadIdx.x >> 6][0] = v;
      __syncthreads();
      total = 0;
      for (int w = 0; w < kEmitWaves; ++w) total += s_part[0][w][0];
    } else {
      total = totals[27];
    }
    const int64_t turn = (int64_t)(gridDim.x - emit_blocks) * kEmitThreads;
    const int64_t r0 = total + (int64_t)(blockIdx.x - emit_blocks) * kEmitThreads;
    if (r0 >= cap) return;
    for (int64_t r = r0 + threadIdx.x; r - threadIdx.x < cap; r += turn) {
      const bool have = r < cap;
      if (have) {
        *(int4*)(out_coors + r * 4) = make_int4(-1, -1, -1, -1);
        counts[r] = 0;
        for (int ch = 0; ch < c; ++ch) {
          out_f32[r * c + ch] = 0.f;
          if (out_bf16) out_bf16[r * c + ch] = 0;
        }
        for (int k = 0; k < 27; ++k) nbr_t[(int64_t)k * cap + r] = -1;
      }
      if (order_hist) {
        // no offsets at all: the "no neighbour" bucket; the rows of a wave take consecutive places behind ONE atomic
        const int key = order_global(order_key_local(0u, 13), (int)blockIdx.x);
        const unsigned long long m = __ballot(have);
        if (m) {
          const int leader = __ffsll((long long)m) - 1;
          uint32_t first = 0u;
          if ((int)(threadIdx.x & 63) == leader) first = atomicAdd(&order_hist[key], (uint32_t)__popcll(m));
          first = __shfl(first, leader, 64);
          const uint32_t place = first + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (have) order_rowrec[r] = i32x4_t{(int)((uint32_t)key | (place << kOrderKeyBits)), 0, -1, -1};
        }
      }
    }
    return;
  }
  uint32_t* bm = smem;            // [words]
  uint32_t* pf = smem + g.words;  // [words] local prefix (rank inside the grid)
  __shared__ int32_t s_base[kCols];
  __shared__ int32_t s_wcnt[27][kEmitWaves];  // per round: valid entries per offset and wave, then their exclusive prefix
  __shared__ int32_t s_run[27];
  __shared__ int2 s_list[kCodesPerThread * kEmitThreads];  // (point - segment start, row) work items of one round
  __shared__ int s_n, s_n2;
  __shared__ int32_t s_cell[kEmitThreads];
  __shared__ int32_t s_code[kCodesPerThread * kEmitThreads];  // the round's point codes (thread t owns slots j * 256 + t)
  const int b = blockIdx.x / g.slices, sl = blockIdx.x % g.slices;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  STAMP(5);
  // (part sums: column c of kernel A's count rows, lane -> column, 8 interleaved parts; asked for FIRST -- they depend
  // on nothing, the point codes below wait for their segment's bounds -- and only summed behind the bitmap: the
  // vector-memory counter retires in order, so consumed here they would hold everything behind them back by a trip)
  constexpr int kPartLoads = kFusedBasesMaxEntries / (2 * kEmitWaves);
  constexpr int kOwnLoads = kMaxSlices / 4;   // slices of the own part in front of this one: at most spa - 1 < 4
  int32_t pv[kPartLoads], pw[kOwnLoads];
  const int pcol = threadIdx.x & 31, ppart = threadIdx.x >> 5;
  if (part_sums) {
#pragma unroll
    for (int u = 0; u < kPartLoads; ++u) {
      const int e = ppart + u * 2 * kEmitWaves;
      pv[u] = (pcol < kCols && e < g.batch * g.asplit) ? part_sums[(int64_t)e * kCols + pcol] : 0;
    }
#pragma unroll
    for (int u = 0; u < kOwnLoads; ++u) {   // the slices of the own part in front of this one
      const int s2 = (sl / g.spa) * g.spa + u;
      pw[u] = (pcol < kCols && ppart == 0 && s2 < sl) ? count_table[((int64_t)b * g.slices + s2) * kCols + pcol] : 0;
    }
  }
  // the point codes of the first round are requested before anything else: their latency hides behind the row loop
  const int64_t p_lo = seg[2 * b], p_hi = seg[2 * b + 1];
  {
    int32_t codes[kCodesPerThread];
#pragma unroll
    for (int j = 0; j < kCodesPerThread; ++j) {
      const int64_t i = p_lo + (int64_t)j * kEmitThreads + threadIdx.x;
      codes[j] = i < p_hi ? code_of[i] : -1;
    }
#pragma unroll
    for (int j = 0; j < kCodesPerThread; ++j) s_code[j * kEmitThreads + threadIdx.x] = codes[j];  // (own slots: read back by this thread)
  }
  for (int w = threadIdx.x; w < g.words; w += kEmitThreads) {
    bm[w] = bitmap[(int64_t)b * g.words + w];
    pf[w] = local_prefix[(int64_t)b * g.words + w];
  }
  if (part_sums) {
    // ... summed over what lies in front of this workgroup's slice (-> its bases), in front of its grid (-> the grid's
    // first row) and over everything (-> totals)
    const int mine = b * g.asplit + sl / g.spa, first = b * g.asplit;
    int32_t a_me = 0, a_grid = 0, a_all = 0;
#pragma unroll
    for (int u = 0; u < kOwnLoads; ++u) a_me += pw[u];
#pragma unroll
    for (int u = 0; u < kPartLoads; ++u) {
      const int e = ppart + u * 2 * kEmitWaves;
      a_me += e < mine ? pv[u] : 0;
      a_grid += e < first ? pv[u] : 0;
      a_all += pv[u];
    }
    a_me += __shfl_xor(a_me, 32, 64);
    a_grid += __shfl_xor(a_grid, 32, 64);
    a_all += __shfl_xor(a_all, 32, 64);
    if (lane < 32) {
      s_part[0][wave][lane] = a_me;
      s_part[1][wave][lane] = a_grid;
      s_part[2][wave][lane] = a_all;
    }
  } else if (threadIdx.x < kCols) {
    s_base[threadIdx.x] = bases[((int64_t)b * g.slices + sl) * kCols + threadIdx.x];
  }
  if (threadIdx.x < 27) s_run[threadIdx.x] = 0;
  if (order_hist)
    for (int i = threadIdx.x; i < kLocalBuckets; i += kEmitThreads) s_oh[i] = 0u;
  // global row of this grid's first voxel: the slice-0 base of the voxel column
  int32_t grid_base = part_sums ? 0 : bases[((int64_t)b * g.slices) * kCols + 27];
  __syncthreads();
  if (part_sums) {
#pragma unroll
    for (int w = 0; w < kEmitWaves; ++w) {
      grid_base += s_part[1][w][27];
    }
    if (threadIdx.x < kCols) {   // (s_base is first read behind the row loop's barriers)
      int32_t me = 0, all = 0;
#pragma unroll
      for (int w = 0; w < kEmitWaves; ++w) {
        me += s_part[0][w][threadIdx.x];
        all += s_part[2][w][threadIdx.x];
      }
      s_base[threadIdx.x] = me;
      if (blockIdx.x == 0) {
        // indice_num[k] = pairs of offset k = entries of table column 26 - k (= column k, by symmetry); geometry.h order
        if (threadIdx.x < 27) indice_num[26 - threadIdx.x] = all;
        else *num_voxels = (int32_t)(all < cap ? all : cap);
      }
    }
    if (blockIdx.x == 0 && wave == 1) {   // status = any workgroup of the first launch saw a point outside its grid's segment
      int bad = 0;
      for (int i = lane; i < nflags; i += 64) bad |= bad_flags[i];
      const unsigned long long m = __ballot(bad != 0);
      if (lane == 0) *status = m ? 1 : 0;
    }
  }
  const int w_lo = sl * g.wps < g.words ? sl * g.wps : g.words, w_hi = (w_lo + g.wps < g.words) ? w_lo + g.wps : g.words;
  const int32_t grid_rows = (int32_t)(pf[g.words - 1] + __popc(bm[g.words - 1]));
  const int32_t row_lo = grid_base + (w_lo < g.words ? (int32_t)pf[w_lo] : grid_rows);
  const int32_t row_hi = grid_base + (w_hi < g.words ? (int32_t)pf[w_hi] : grid_rows);
  // the global prefix words (rank of a cell in the whole batch): what ococc_grid_unique_workspace_layout describes
  for (int w = w_lo + threadIdx.x; w < w_hi; w += kEmitThreads) prefix_out[(int64_t)b * g.words + w] = (uint32_t)grid_base + pf[w];

  STAMP(6);
  // ---- rows of this slice, 256 at a time in row order: coordinates, neighbour table, masks, pairs ----
  const int nrows = row_hi - row_lo;
  const int plane = g.gx * g.gy;
  for (int r0 = 0; r0 < nrows; r0 += kEmitThreads) {
    const int rl = r0 + threadIdx.x;           // row inside the slice
    const bool have = rl < nrows;
    const int32_t row = row_lo + rl;            // global row
    // the cells of this round's rows, in row order: one pass over the slice's words (thread = word) into LDS
    for (int w = w_lo + threadIdx.x; w < w_hi; w += kEmitThreads) {
      uint32_t bits = bm[w];
      int idx = (int)pf[w] - (row_lo - grid_base) - r0;  // position of the word's first voxel in this round
      if (idx >= kEmitThreads || idx + 32 <= 0) continue;
      while (bits) {
        const int bb = __ffs(bits) - 1;
        bits &= bits - 1;
        if ((unsigned)idx < (unsigned)kEmitThreads) s_cell[idx] = w * 32 + bb;
        ++idx;
      }
    }
    LDS_BARRIER();
    int32_t cell = 0;
    int x = 0, y = 0, z = 0;
    if (have) {
      cell = s_cell[threadIdx.x];
      x = cell % g.gx; y = (cell / g.gx) % g.gy; z = cell / plane;
      if (row < cap) {
        *(int4*)(out_coors + (int64_t)row * 4) = make_int4(b, z, y, x);
        counts[row] = 1;
      }
    }
    // the three x-neighbours of one (dz, dy) are consecutive cells: ONE 64-bit window of the bitmap and ONE prefix word
    // serve all three.  (Unrolled on purpose: nine independent LDS round trips in flight; a rolled body, or a second
    // pass that recomputes the neighbours for the pair lists, measured 4 us slower per round.)
    int32_t nb[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) nb[k] = -1;
    if (have) {
#pragma unroll
      for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int zz = z + kz - 1, yy = y + ky - 1;
          if ((unsigned)zz >= (unsigned)g.gz || (unsigned)yy >= (unsigned)g.gy) continue;
          const int32_t c0 = cell + (kz - 1) * plane + (ky - 1) * g.gx - 1;  // the kx = 0 cell (-1 only at the grid's first cell)
          const int32_t cs = c0 < 0 ? 0 : c0;
          const int sh = (cs & 31) - (c0 < 0 ? 1 : 0);
          const int wq = cs >> 5;
          const uint32_t wl = bm[wq];
          const uint32_t wh = ((cs & 31) > 29 && wq + 1 < g.words) ? bm[wq + 1] : 0u;
          const unsigned long long both = ((unsigned long long)wh << 32) | wl;
          const uint32_t pre = pf[wq];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int pos = sh + kx;
            if ((unsigned)(x + kx - 1) < (unsigned)g.gx && pos >= 0 && ((both >> pos) & 1ull))
              nb[(kz * 3 + ky) * 3 + kx] = grid_base + (int32_t)(pre + (uint32_t)__popcll(both & ((1ull << pos) - 1ull)));
          }
        }
    }
    STAMP(11);
    // neighbour table (offset-major) and per-wave counts of valid entries per offset
    uint32_t mbits = 0;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      if (have && row < cap) nbr_t[(int64_t)k * cap + row] = nb[k];
      const unsigned long long m = __ballot(nb[k] >= 0);
      if (lane == 0) s_wcnt[k][wave] = (int32_t)__popcll(m);
      if (nb[k] >= 0) mbits |= 1u << k;
    }
    if (order_hist && have && row < cap) {
      int32_t e1 = -1, e2 = -1;
#pragma unroll
      for (int k = 0; k < 27; ++k)
        if (k != 13 && nb[k] >= 0) {
          if (e1 < 0) e1 = nb[k];
          else if (e2 < 0) e2 = nb[k];
        }
      const int key = order_key_local(mbits, 13);
      const uint32_t rank = atomicAdd(&s_oh[key], 1u);   // place inside this workgroup's share of the bucket
      order_rowrec[row] = i32x4_t{(int)((uint32_t)order_global(key, (int)blockIdx.x) | (rank << kOrderKeyBits)),
                                  (int)mbits, e1, e2};
    }
    STAMP(12);
    // 16-row block masks.  Rows are consecutive along the lanes, so a block is a run of lanes: segmented OR towards
    // the run's first lane, which ORs the result into the mask word (cleared by kernel A; a block that continues in
    // another wave, slice or grid gets one OR from each)
    {
      uint32_t v = mbits;
      const int32_t blk = have ? (row >> 4) : -1 - lane;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const uint32_t o = __shfl_down(v, d, 64);
        const int32_t ob = __shfl_down(blk, d, 64);
        if (lane + d < 64 && ob == blk) v |= o;
      }
      const int32_t pb = __shfl_up(blk, 1, 64);
      if (have && row < cap && v && (lane == 0 || pb != blk)) atomicOr(blockmask + blk, v);
    }
    STAMP(13);
    LDS_BARRIER();
    if (threadIdx.x < 27) {  // counts -> position of each wave's first entry (base of the slice + rounds so far + waves before)
      int32_t acc = s_base[threadIdx.x] + s_run[threadIdx.x];
      const int32_t start = acc;
      for (int t = 0; t < kEmitWaves; ++t) {
        const int32_t cnt = s_wcnt[threadIdx.x][t];
        s_wcnt[threadIdx.x][t] = acc;
        acc += cnt;
      }
      s_run[threadIdx.x] += acc - start;
    }
    LDS_BARRIER();
    STAMP(14);
    // pairs: for table column k the list of offset 26 - k holds (in = row, out = nb[k]) in ascending row order
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      if (nb[k] >= 0) {
        const unsigned long long m = __ballot(true);
        const int32_t p = s_wcnt[k][wave] + (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (p < cap) {
          pairs[((int64_t)(26 - k) * 2 + 0) * cap + p] = row;
          pairs[((int64_t)(26 - k) * 2 + 1) * cap + p] = nb[k];
        }
      }
    }
    LDS_BARRIER();
  }

  // the workgroup's share of every bucket starts where the bucket's counter stood: the atomics are asked for here and
  // their answers used at the very end of the kernel, behind the point phases
  constexpr int kOrderPer = (kLocalBuckets + kEmitThreads - 1) / kEmitThreads;
  uint32_t o_cnt[kOrderPer], o_got[kOrderPer];
  if (order_hist) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kOrderPer; ++i) {
      const int bk = threadIdx.x + kEmitThreads * i;
      o_cnt[i] = bk < kLocalBuckets ? s_oh[bk] : 0u;
      o_got[i] = 0u;
      if (o_cnt[i]) o_got[i] = atomicAdd(&order_hist[order_global(bk, (int)blockIdx.x)], o_cnt[i]);
    }
  }
  STAMP(7);
  // ---- points of this grid whose voxel row belongs to the slice, kCodesPerThread * 256 per round ----
  // Every phase first SCANS the round's points (codes -> rank -> "mine?") into a work list in LDS and then serves the
  // list cooperatively, one thread per (entry, 16-byte piece or channel): the loads of a round are all in flight
  // before the first store or atomic waits for one, and the few threads that own a shared row do not hold up a wave.
  const int32_t cells = g.words * 32;
  const int64_t per_round = (int64_t)kCodesPerThread * kEmitThreads;
  const bool one_round = p_hi - p_lo <= per_round;
  const int pieces = (c + 3) / 4;
  auto load_codes = [&](int64_t base) {
    int32_t codes[kCodesPerThread];
#pragma unroll
    for (int j = 0; j < kCodesPerThread; ++j) {
      const int64_t i = base + (int64_t)j * kEmitThreads + threadIdx.x;
      codes[j] = i < p_hi ? code_of[i] : -1;
    }
#pragma unroll
    for (int j = 0; j < kCodesPerThread; ++j) s_code[j * kEmitThreads + threadIdx.x] = codes[j];
  };
  // phase 1: inv for every owned point; first arrivals copy their features into the row (f32 and bf16)
  for (int64_t base = p_lo; base < p_hi; base += per_round) {
    if (base != p_lo) load_codes(base);
    if (threadIdx.x == 0) s_n = 0;
    LDS_BARRIER();
#pragma unroll 1
    for (int j = 0; j < kCodesPerThread; ++j) {
      const int64_t i = base + (int64_t)j * kEmitThreads + threadIdx.x;
      const int32_t code = s_code[j * kEmitThreads + threadIdx.x];
      if (i >= p_hi) continue;
      if (code < 0) {
        if (sl == 0 && code == -2) inv[i] = -1;  // (codes <= -3: a finaliser mark of an earlier call: never seen here)
        continue;
      }
      const int32_t r = grid_base + lds_rank(bm, pf, (code >> 1) - b * cells);
      if (r < row_lo || r >= row_hi) continue;
      inv[i] = r;
      if (!(code & 1) && r < cap) {
        const int slot = atomicAdd(&s_n, 1);
        s_list[slot] = make_int2((int)(i - p_lo), r);
      }
    }
    LDS_BARRIER();
    const int nl = s_n;
    for (int t = threadIdx.x; t < nl * pieces; t += kEmitThreads) {
      const int2 e = s_list[t / pieces];
      const int piece = t % pieces;
      const float* src = feats + (p_lo + e.x) * c;
      float* dst = out_f32 + (int64_t)e.y * c;
      uint16_t* dst16 = out_bf16 ? out_bf16 + (int64_t)e.y * c : nullptr;
      if ((c & 3) == 0) {
        const float4 v = *(const float4*)(src + piece * 4);
        *(float4*)(dst + piece * 4) = v;
        if (dst16) {
          uint2 q;
          q.x = (uint32_t)ococc_f32_to_bf16(v.x) | ((uint32_t)ococc_f32_to_bf16(v.y) << 16);
          q.y = (uint32_t)ococc_f32_to_bf16(v.z) | ((uint32_t)ococc_f32_to_bf16(v.w) << 16);
          *(uint2*)(dst16 + piece * 4) = q;
        }
      } else {
        for (int ch = piece * 4; ch < piece * 4 + 4 && ch < c; ++ch) {
          dst[ch] = src[ch];
          if (dst16) dst16[ch] = ococc_f32_to_bf16(src[ch]);
        }
      }
    }
    LDS_BARRIER();
  }
  __syncthreads();  // the row copies are complete (s_waitcnt vmcnt(0) in front of the barrier) before anything adds to them
  STAMP(8);
  // phase 2: later arrivals add their features to the row and count themselves; the one that takes a row's count from
  // 1 to 2 marks itself (code -3 - row) as the row's finaliser
  int32_t* s_nfin = &s_n2;
  if (threadIdx.x == 0) s_n2 = 0;
  for (int64_t base = p_lo; base < p_hi; base += per_round) {
    if (!one_round) load_codes(base);
    if (threadIdx.x == 0) s_n = 0;
    LDS_BARRIER();
#pragma unroll 1
    for (int j = 0; j < kCodesPerThread; ++j) {
      const int64_t i = base + (int64_t)j * kEmitThreads + threadIdx.x;
      const int32_t code = s_code[j * kEmitThreads + threadIdx.x];
      if (i >= p_hi || code < 0 || !(code & 1)) continue;
      const int32_t r = grid_base + lds_rank(bm, pf, (code >> 1) - b * cells);
      if (r < row_lo || r >= row_hi || r >= cap) continue;
      const int slot = atomicAdd(&s_n, 1);
      s_list[slot] = make_int2((int)(i - p_lo), r);
    }
    LDS_BARRIER();
    const int nl = s_n;
    for (int t = threadIdx.x; t < nl * c; t += kEmitThreads) {
      const int2 e = s_list[t / c];
      const int ch = t % c;
      atomicAdd(out_f32 + (int64_t)e.y * c + ch, feats[(p_lo + e.x) * c + ch]);
      if (ch == 0 && atomicAdd(counts + e.y, 1) == 1) {
        code_of[p_lo + e.x] = -3 - e.y;
        atomicAdd(s_nfin, 1);
      }
    }
    LDS_BARRIER();
  }
  __syncthreads();  // the adds and marks are complete
  STAMP(9);
  // phase 3: rows with more than one point: sum -> mean (read past L1: the adds went to L2)
  if (s_n2 > 0) {
    for (int64_t base = p_lo; base < p_hi; base += per_round) {
      load_codes(base);
      if (threadIdx.x == 0) s_n = 0;
      LDS_BARRIER();
#pragma unroll 1
      for (int j = 0; j < kCodesPerThread; ++j) {
        const int64_t i = base + (int64_t)j * kEmitThreads + threadIdx.x;
        const int32_t code = s_code[j * kEmitThreads + threadIdx.x];
        if (i >= p_hi || code > -3) continue;
        const int32_t r = -3 - code;
        if (r < row_lo || r >= row_hi) continue;  // another slice's finaliser
        const int slot = atomicAdd(&s_n, 1);
        s_list[slot] = make_int2((int)(i - p_lo), r);
      }
      LDS_BARRIER();
      const int nl = s_n;
      for (int t = threadIdx.x; t < nl * c; t += kEmitThreads) {
        const int64_t r = s_list[t / c].y;
        const int ch = t % c;
        const float cnt = (float)__hip_atomic_load(counts + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float v = __hip_atomic_load(out_f32 + r * c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / cnt;
        out_f32[r * c + ch] = v;
        if (out_bf16) out_bf16[r * c + ch] = ococc_f32_to_bf16(v);
      }
      LDS_BARRIER();
    }
  }
  STAMP(10);
  if (order_hist) {
    // this workgroup's share of every bucket starts where the counter stood when its atomic arrived
#pragma unroll
    for (int i = 0; i < kOrderPer; ++i) {
      const int bk = threadIdx.x + kEmitThreads * i;
      if (o_cnt[i]) s_oh[bk] = o_got[i];
    }
    __syncthreads();
    const int32_t hi = row_hi < cap ? row_hi : (int32_t)cap;
    // the records' first word (bucket | place inside the workgroup's share << 12) gets that start added;
    // ococc_subm_row_order_place moves them to their slots.  (Assigning the slots HERE, behind a barrier over the whole
    // grid, was built and measured: EXPERIMENTS.md (i) -- 81-136 us per geometry against 55.5 with the second launch.)
    for (int32_t r = row_lo + threadIdx.x; r < hi; r += kEmitThreads) {
      const uint32_t x = (uint32_t)order_rowrec[r].x;
      const int lk = order_local((int)(x & ((1u << kOrderKeyBits) - 1u)));   // global bucket -> this workgroup's
      order_rowrec[r].x = (int)(x + (s_oh[lk] << kOrderKeyBits));
    }
  }
}

struct GeoLayout {
  int64_t words_total, off_bitmap, off_prefix, off_code, off_lpre, off_table, off_bases, off_totals, off_seg, off_bad, total;
};

inline bool make_geo_layout(int64_t n, int32_t batch, const int32_t* grid_zyx, int32_t slices, GeoLayout* L,
                            int64_t* grid_unique_total) {
  if (n < 0 || batch < 1 || !grid_zyx || slices < 1 || slices > kMaxSlices) return false;
  int64_t cells = 1;
  for (int i = 0; i < 3; ++i) {
    if (grid_zyx[i] < 1) return false;
    cells *= grid_zyx[i];
  }
  if (cells % 32 != 0 || cells * batch >= (1LL << 30)) return false;
  const int64_t words = cells / 32;
  if (words * 8 > 128 * 1024) return false;  // bitmap + prefix of one grid in LDS
  L->words_total = words * batch;
  // the first part is exactly the grid_unique layout (bitmap | prefix | scan scratch), so that the tag
  // spconv.ops.get_indice_pairs reads (ococc_grid_unique_workspace_layout) describes this workspace too
  const int32_t dims[4] = {batch, grid_zyx[0], grid_zyx[1], grid_zyx[2]};
  const int64_t gu = ococc_grid_unique_workspace_bytes(4, dims);
  if (gu < 0) return false;
  *grid_unique_total = gu;
  L->off_bitmap = 0;
  L->off_prefix = ococc_align_up(L->words_total * 4, 256);
  L->off_code = gu;
  L->off_lpre = L->off_code + ococc_align_up(n * 4, 256);
  L->off_table = L->off_lpre + ococc_align_up(L->words_total * 4, 256);
  const int64_t entries = (int64_t)batch * slices;
  L->off_bases = L->off_table + ococc_align_up(entries * kCols * 4, 256);
  L->off_totals = L->off_bases + ococc_align_up(entries * kCols * 4, 256);
  L->off_seg = L->off_totals + 256;
  L->off_bad = L->off_seg + ococc_align_up((int64_t)batch * 16, 256);
  L->total = L->off_bad + ococc_align_up((int64_t)batch * kMaxSlices * 4, 256);
  return true;
}

}  // namespace

#ifdef OCOCC_GEO_STAMPS
extern "C" int ococc_geo_set_stamps(long long* dev_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dev_buffer, sizeof(dev_buffer)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int64_t ococc_object_grid_geometry_workspace_bytes(int64_t n, int32_t batch_size,
                                                              const int32_t host_grid_zyx[3], int32_t slices) {
  GeoLayout L;
  int64_t gu;
  if (!make_geo_layout(n, batch_size, host_grid_zyx, slices, &L, &gu)) return -1;
  return L.total;
}

extern "C" int ococc_object_grid_geometry_order_f32(const float* points, int32_t num_point_features, const int32_t* batch_idx,
                                                    int64_t n, const float* feats, int32_t c, const float host_voxel_size[3],
                                                    const float host_coors_range[6], int32_t batch_size,
                                                    const int32_t host_grid_zyx[3], int32_t slices, int32_t* voxel_coors,
                                                    int64_t capacity, int32_t* inv, int32_t* counts, float* voxel_feats,
                                                    uint16_t* voxel_feats_bf16, int32_t* num_voxels, int32_t* status,
                                                    int32_t* nbr_t, uint32_t* blockmask, int32_t* indice_pairs,
                                                    int32_t* indice_num, void* workspace, int64_t workspace_bytes,
                                                    void* order_counters, int32_t* order_rowrec, int32_t* order_rec,
                                                    int32_t* order_hdr, int32_t heavy_blocks, int32_t mid_blocks,
                                                    ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE((order_counters == nullptr) == (order_rowrec == nullptr), "order_counters and order_rowrec go together");
  OCOCC_REQUIRE((order_rec == nullptr) == (order_hdr == nullptr) && (!order_rec || order_rowrec),
                "order_rec and order_hdr go together, with order_counters and order_rowrec");
  OCOCC_REQUIRE(!order_rec || ((heavy_blocks == 4 || heavy_blocks == 8 || heavy_blocks == 16) &&
                               (mid_blocks == 4 || mid_blocks == 8 || mid_blocks == 16) && ((uintptr_t)order_rec & 15) == 0),
                "tile sizes must be 4, 8 or 16 blocks; order_rec 16-byte aligned");
  OCOCC_REQUIRE(!order_rowrec || (capacity < kOrderMaxRows && ((uintptr_t)order_rowrec & 15) == 0),
                "row records: 16-byte aligned, below 2^20 rows");
  GeoLayout L;
  int64_t gu;
  OCOCC_REQUIRE(n >= 1 && capacity >= 1 && c >= 1 && num_point_features >= 3, "bad sizes");
  OCOCC_REQUIRE(host_voxel_size && host_coors_range, "null voxel_size / coors_range");
  OCOCC_REQUIRE(make_geo_layout(n, batch_size, host_grid_zyx, slices, &L, &gu),
                "need batch >= 1, grid cells a multiple of 32 and <= 512 Ki per grid, batch * cells < 2^30, 1 <= slices <= 16");
  OCOCC_REQUIRE(num_voxels && status && status == num_voxels + 1, "num_voxels/status: one device int32[2]");
  OCOCC_REQUIRE(points && batch_idx && feats && voxel_coors && inv && counts && voxel_feats && nbr_t && blockmask &&
                    indice_pairs && indice_num, "null device pointer");
  OCOCC_REQUIRE(workspace && workspace_bytes >= L.total, "workspace too small");
  GeoParams g;
  for (int i = 0; i < 3; ++i) {
    OCOCC_REQUIRE(host_voxel_size[i] > 0.f, "voxel size must be positive");
    const int gi = (int)ceilf((host_coors_range[3 + i] - host_coors_range[i]) / host_voxel_size[i]);
    OCOCC_REQUIRE(gi == host_grid_zyx[2 - i], "grid_zyx does not match coors_range / voxel_size");
  }
  g.vx = host_voxel_size[0]; g.vy = host_voxel_size[1]; g.vz = host_voxel_size[2];
  g.xmin = host_coors_range[0]; g.ymin = host_coors_range[1]; g.zmin = host_coors_range[2];
  g.gx = host_grid_zyx[2]; g.gy = host_grid_zyx[1]; g.gz = host_grid_zyx[0];
  g.batch = batch_size;
  g.words = (int32_t)(L.words_total / batch_size);
  g.slices = slices;
  g.wps = (int32_t)ococc_align_up((g.words + slices - 1) / slices, 64);  // whole waves of words per slice
  g.asplit = slices < 4 ? slices : 4;  // (1024-thread workgroups: one per CU; 4 x 64 grids fill the chip once)
  g.spa = (slices + g.asplit - 1) / g.asplit;
  char* ws = (char*)workspace;
  uint32_t* bitmap = (uint32_t*)(ws + L.off_bitmap);
  uint32_t* prefix = (uint32_t*)(ws + L.off_prefix);
  int32_t* code_of = (int32_t*)(ws + L.off_code);
  uint32_t* lpre = (uint32_t*)(ws + L.off_lpre);
  int32_t* table = (int32_t*)(ws + L.off_table);
  int32_t* bases = (int32_t*)(ws + L.off_bases);
  int32_t* totals = (int32_t*)(ws + L.off_totals);
  int64_t* seg = (int64_t*)(ws + L.off_seg);
  int32_t* bad_flags = (int32_t*)(ws + L.off_bad);
  const size_t lds = (size_t)g.words * 8;
  const int64_t mask_words = ococc_cdiv(capacity, 16);
  if (lds > 48 * 1024) {
    OCOCC_HIP(hipFuncSetAttribute((const void*)grid_mark_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    OCOCC_HIP(hipFuncSetAttribute((const void*)grid_emit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // the prefix of the count table: inside kernel B for small batches (kernel A also leaves the sums of each of its
  // workgroups' slices; every workgroup of B reads those in front of its own: quadratic in the batch, one trip to L2),
  // a launch of its own for large ones
  const bool fused_bases = OCOCC_GEO_FUSED_BASES && (int64_t)batch_size * g.asplit <= kFusedBasesMaxEntries;
  int32_t* part_sums = fused_bases ? bases : nullptr;   // (the bases' own space: nobody writes them in this form)
  hipLaunchKernelGGL(grid_mark_count_kernel, dim3(batch_size * g.asplit), dim3(kThreads), lds, stream, points,
                     (int)num_point_features, batch_idx, n, g, bitmap, lpre, code_of, table, seg, blockmask, mask_words,
                     inv, bad_flags, (uint32_t*)order_counters, part_sums);
  OCOCC_CHECK_LAUNCH();
  const int64_t entries = (int64_t)batch_size * slices;
  if (!fused_bases) {
    hipLaunchKernelGGL(geometry_bases_kernel, dim3(kCols), dim3(64), 0, stream, table, entries, bases, totals,
                       indice_num, num_voxels, capacity, bad_flags, (int)(batch_size * g.asplit), status);
    OCOCC_CHECK_LAUNCH();
  }
  const int emit_blocks = (int)entries;
  // the padding workgroups index rows from the device-side total and take them in turns (worst case every row is
  // padding: an empty batch)
  const int64_t pad_need = ococc_cdiv(capacity, kEmitThreads);
  const int pad_blocks = (int)(pad_need < kPadBlocks ? pad_need : kPadBlocks);
  hipLaunchKernelGGL(grid_emit_kernel, dim3(emit_blocks + pad_blocks), dim3(kEmitThreads), lds, stream, feats, (int)c, n, g,
                     bitmap, lpre, prefix, code_of, seg, bases, totals, inv, voxel_coors, counts, voxel_feats,
                     voxel_feats_bf16, capacity, nbr_t, blockmask, indice_pairs, emit_blocks, (uint32_t*)order_counters,
                     (i32x4_t*)order_rowrec, (const int32_t*)table, (const int32_t*)part_sums, indice_num, num_voxels,
                     (const int32_t*)bad_flags, (int)(batch_size * g.asplit), status);
  OCOCC_CHECK_LAUNCH();
  if (order_rec)   // the slots: a launch of its own
    return ococc_subm_row_order_place(order_rowrec, 27, 13, capacity, heavy_blocks, mid_blocks, order_counters, order_rec,
                                      order_hdr, stream_);
  return OCOCC_OK;
}

extern "C" int ococc_object_grid_geometry_f32(const float* points, int32_t num_point_features, const int32_t* batch_idx,
                                              int64_t n, const float* feats, int32_t c, const float host_voxel_size[3],
                                              const float host_coors_range[6], int32_t batch_size,
                                              const int32_t host_grid_zyx[3], int32_t slices, int32_t* voxel_coors,
                                              int64_t capacity, int32_t* inv, int32_t* counts, float* voxel_feats,
                                              uint16_t* voxel_feats_bf16, int32_t* num_voxels, int32_t* status,
                                              int32_t* nbr_t, uint32_t* blockmask, int32_t* indice_pairs,
                                              int32_t* indice_num, void* workspace, int64_t workspace_bytes,
                                              ococc_stream_t stream_) {
  return ococc_object_grid_geometry_order_f32(points, num_point_features, batch_idx, n, feats, c, host_voxel_size,
                                              host_coors_range, batch_size, host_grid_zyx, slices, voxel_coors, capacity, inv,
                                              counts, voxel_feats, voxel_feats_bf16, num_voxels, status, nbr_t, blockmask,
                                              indice_pairs, indice_num, workspace, workspace_bytes, nullptr, nullptr, nullptr,
                                              nullptr, 0, 0, stream_);
}
