// B4  sub-manifold convolution over rows taken in NEIGHBOUR-PATTERN order.
// Same contract as ococc_sparse_conv_gather_gemm_bf16 (out[o] = sum_k feat[table[k][o]] @ W[k],
// replaces indiceConv / indiceConvBackward, spconv_ops.h:260-456); the result is bit-identical to it:
// a row's products are still added in ascending offset order into an f32 accumulator of its own.
//
// Why.  The output-stationary kernels (sparse_conv.hip) walk all kvol offsets for every 256-row workgroup and
// issue a 16-row MFMA block whenever ANY of its rows has a neighbour at the offset.  On sparse active sets (the
// benchmark: 0.76 neighbours per voxel besides itself) rows in voxel order have unrelated neighbour patterns:
// 13 305 (workgroup, offset) iterations -- each one weight slice through LDS and one barrier -- and 85 404 active
// 16-row blocks for 126 k rows, 6.1 x what the 222 k rulebook pairs need.  Nothing in the contract fixes the order
// in which OUTPUT rows are processed, so the rows are bucketed by pattern first: class (3+, 2, 1, 0 neighbours), then
// the two lowest neighbour offsets.  In that order a workgroup's rows share their offsets: 1 811 iterations and
// 17 913 active blocks (1.29 x the pairs) on the same grids.  Rows are addressed through the order (slot -> row),
// the centre tap becomes a gather and the output store a scatter of whole rows; nothing else changes.
//
// ococc_subm_row_order builds the order from the offset-major gather table (two small launches: count, place); it belongs to the rulebook and is shared by every layer and direction that uses the table.  A slot's record
// carries the row, its offset mask and the table entries of its two lowest neighbour offsets: rows with at most two
// neighbours (95 % of the benchmark's) never touch the table again.
// Rows with many neighbours need (nearly) all offsets whatever the order; they are dealt out first and in
// smaller tiles (64 rows, then 128, then 256) so that the long workgroups are many and start at once, and a wave with
// fewer than four 16-row blocks uses its row registers to gather several OFFSETS ahead instead.
// (Tried and dropped, tools/probe/sparse_conv_sorted_tasks.hip.txt: the many-neighbour blocks one by one on single
// waves with the weight fragments straight from L2 -- no LDS, no barrier, three offsets in flight.  16 KB of weights
// per 16 rows and offset is L2 bandwidth: 24 us for the 313 such blocks of the benchmark, ~9 TB/s at any size.)
#include "common.hpp"
#include "stream_ops.hpp"
#include "row_order.hpp"
#include "ln_math.hpp"
#include <type_traits>

namespace {

#ifdef OCOCC_SORTED_STAMPS
// diagnostic build only (tools/probe/sorted_stamps.py): wall-clock stamps (100 MHz) per tile and phase into a buffer of their own
__device__ long long* s_stamps = nullptr;
#define SSTAMP(tile, slot) do { if (threadIdx.x == 0 && s_stamps) s_stamps[(int64_t)(tile) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define SNOTE(tile, slot, v) do { if (threadIdx.x == 0 && s_stamps) s_stamps[(int64_t)(tile) * 8 + (slot)] = (long long)(v); } while (0)
#else
#define SSTAMP(tile, slot) do { } while (0)
#define SNOTE(tile, slot, v) do { } while (0)
#endif

constexpr int kSortThreads = 256;
#ifndef OCOCC_SORTED_LIGHT_BLOCKS
#define OCOCC_SORTED_LIGHT_BLOCKS 16
#endif
constexpr int kLightBlocks = OCOCC_SORTED_LIGHT_BLOCKS;   // 16-row blocks per tile of the rows with at most one neighbour
constexpr int kOrderRowsPerWg = 512;

// rowrec[r] = {bucket | place in the bucket << 11, offset mask, table entry at the lowest neighbour offset, at the
// second lowest}.  (For the benchmark's grids grid_emit_kernel writes these records while it writes the table.)
__global__ void __launch_bounds__(256)
order_count_kernel(const int32_t* __restrict__ table, int kvol, int dense_k, int64_t n, i32x4_t* __restrict__ rowrec,
                   uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[kLocalBuckets];
  for (int i = threadIdx.x; i < kLocalBuckets; i += 256) h[i] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kOrderRowsPerWg;
  const int copy = (int)blockIdx.x;   // (which of a bucket's counter copies: order_global takes it modulo)
  i32x4_t rr[kOrderRowsPerWg / 256];
  int lkey[kOrderRowsPerWg / 256];
#pragma unroll
  for (int u = 0; u < kOrderRowsPerWg / 256; ++u) {
    const int64_t r = base + u * 256 + threadIdx.x;
    rr[u] = i32x4_t{0, 0, -1, -1};
    if (r < n) {
      int32_t e[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) e[k] = k < kvol ? table[(int64_t)k * n + r] : -1;   // (all in flight at once)
      uint32_t m = 0;
      int32_t e1 = -1, e2 = -1;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (e[k] >= 0) {
          m |= 1u << k;
          if (k != dense_k) {
            if (e1 < 0) e1 = e[k];
            else if (e2 < 0) e2 = e[k];
          }
        }
      }
      lkey[u] = order_key_local(m, dense_k);
      const uint32_t rank = atomicAdd(&h[lkey[u]], 1u);   // place inside the workgroup's share of the bucket
      rr[u] = i32x4_t{(int)((uint32_t)order_global(lkey[u], copy) | (rank << kOrderKeyBits)), (int)m, e1, e2};
    }
  }
  __syncthreads();
  order_reserve<256>(h, hist, copy);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kOrderRowsPerWg / 256; ++u) {
    const int64_t r = base + u * 256 + threadIdx.x;
    if (r < n) {
      rr[u].x = (int)((uint32_t)rr[u].x + (h[lkey[u]] << kOrderKeyBits));
      rowrec[r] = rr[u];
    }
  }
}

// Slots.  Every workgroup scans the (small) bucket histogram itself -- no scan launch in between -- and moves its rows'
// records to first slot of the bucket + the row's place in the bucket, which the counting pass left in the record (a
// second round of atomics here, one per workgroup and bucket, took 12 of this kernel's 18 us; per-row atomics on the
// cursors of the few big buckets 666 us).  Workgroup 0 writes the header.  The counters are left as they are: every
// build clears them before it counts (no "who is last" atomic at the end of this kernel).
__global__ void __launch_bounds__(256)
order_place_kernel(const i32x4_t* __restrict__ rowrec, int64_t n, const uint32_t* __restrict__ hist,
                   int heavy_blocks, int mid_blocks, int dense_k, i32x4_t* __restrict__ rec, OrderHdr* __restrict__ hdr) {
  __shared__ uint32_t start[kOrderBuckets + 1];   // first slot of the bucket
  __shared__ uint32_t part[4];
  const int64_t base = (int64_t)blockIdx.x * kOrderRowsPerWg;
  i32x4_t rr[kOrderRowsPerWg / 256];
#pragma unroll
  for (int u = 0; u < kOrderRowsPerWg / 256; ++u) {   // (asked for first: in flight under the scan)
    const int64_t r = base + u * 256 + threadIdx.x;
    rr[u] = r < n ? rowrec[r] : i32x4_t{0, 0, 0, 0};
  }
  order_scan_starts<256>([&](int b) { return hist[b]; }, start, part);
  if (blockIdx.x == 0 && threadIdx.x == 0) order_write_hdr(start, n, heavy_blocks, mid_blocks, dense_k, hdr);
#pragma unroll
  for (int u = 0; u < kOrderRowsPerWg / 256; ++u) {
    const int64_t r = base + u * 256 + threadIdx.x;
    if (r < n) {
      const uint32_t x = (uint32_t)rr[u].x;
      rec[start[x & ((1u << kOrderKeyBits) - 1u)] + (x >> kOrderKeyBits)] = i32x4_t{(int)r, rr[u].y, rr[u].z, rr[u].w};
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The convolution.  The per-offset pipeline is the streamed-weights kernel's (sparse_conv.hip: weights of one offset
// through LDS by DMA; rows gathered ahead with buffer loads that return zeros for "no neighbour" without touching
// memory; hand-counted vmcnt waits, one barrier per offset), walked over the SET BITS of the workgroup's offset mask
// instead of 0..kvol-1, with the changes the short, latency-bound walks need (in-kernel stamps, tools/probe/
// sorted_stamps.py: every dependent trip to memory costs ~2 us with all workgroups in the same phase):
//  - the tile's table entries sit in LDS before the walk starts: from the slot records for rows with at most two
//    neighbours, one batch of table loads for the others (asked for one offset ahead inside the walk, every iteration
//    waited ~1 us for its entry);
//  - weights run two offsets ahead through three LDS buffers;
//  - a wave owns BPW = 4, 2 or 1 blocks of 16 rows; its 2 x 4 row-register sets hold 4 / BPW offsets each, gathered a
//    whole group (4 / BPW offsets) ahead: the 64-row tiles of the many-neighbour rows walk ~27 offsets with the rows
//    of offsets i+4 .. i+7 in flight.
// Workgroups are persistent and take tiles round-robin (the first tiles are the long ones); in the second round the
// workgroups behind the long tiles go first.
// LayerNorm (+ GELU) BACKWARD in the epilogue (LNB instantiations: the input-gradient convolution behind a
// conv -> LayerNorm -> act block, functional.LnBackwardLink): the finished f32 row of d(block output) is in the
// accumulators of the four lanes that share a slot, so the block's LayerNorm backward happens there and d(block's conv
// output) leaves instead -- the arithmetic and its order are those of ln_act_bwd_vec_kernel and of the tile kernel's LNB
// epilogue (ln_math.hpp; the row sums are combined over the 8-channel pieces in the same butterfly order), so the rows
// are bit-identical to both; the d gamma / d beta sums are grouped per workgroup (partials [gridDim.x][2 NC]).
struct SortedLn {
  const uint16_t* y;       // [n_out, NC] bf16: the block's conv output
  const float* mean_rstd;  // [n_out, 2]
  const float* gamma;
  const float* beta;
  int act;                 // 0 none, 1 GELU(erf)
  float* partials;         // [gridDim.x][2 NC]
  // LayerNorm (+ GELU) FORWARD epilogue (LNF instantiations: the conv -> norm -> act block of make_sparse_convmodule,
  // sparse_block.py:216-289, in one launch): the activation and the row statistics leave beside the conv output
  float eps;
  uint16_t* act_out;       // [n_out, NC] bf16
  float* stats_out;        // [n_out, 2] mean, rstd
};

template <int KD, int NC, bool LNB = false>
constexpr int sorted_lds_bytes(int kvol) {
  return 3 * NC * (KD / 8) * 16 + kvol * kSortThreads * 4 + 2 * (kSortThreads / 64) * 4 +
         (LNB ? (kSortThreads / 64) * 2 * NC * 4 : 0);
}

template <int KD, int NC, bool OUT_BF16, bool LNB = false, bool LNF = false>
__global__ void __launch_bounds__(kSortThreads, 2)
gather_gemm_sorted_kernel(const uint16_t* __restrict__ feat, uint32_t feat_bytes, const uint16_t* __restrict__ wn, int kvol,
                          const int32_t* __restrict__ table, const i32x4_t* __restrict__ rec,
                          const OrderHdr* __restrict__ hdr, int64_t n_out, const float* __restrict__ bias,
                          void* __restrict__ out_, SortedLn ln) {
  constexpr int KSTEPS = KD / 32;
  constexpr int NB = NC / 16;
  constexpr int PPR = KD / 8;             // 16-byte pieces per weight row
  constexpr int RPB = 256 / (KD * 2);     // weight rows per 256-byte LDS bank row
  constexpr int PIECES = NC * PPR;        // per offset
  constexpr int CHUNKS = PIECES / 64;     // 1 KB chunks, one global_load_lds_dwordx4 each
  constexpr int NWAVES = kSortThreads / 64;
  constexpr int CPW = (CHUNKS + NWAVES - 1) / NWAVES;
  constexpr int NG = 4 * KSTEPS;          // row pieces one group gathers (4 virtual blocks)
  static_assert(KD % 32 == 0 && PIECES % 64 == 0 && NB % 2 == 0 && KSTEPS <= 4, "sorted kernel shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* wl = (u32x4*)smem;                                // [3][PIECES] (layout and swizzle: see gather_gemm_stream_kernel)
  int32_t* sidx = (int32_t*)(wl + 3 * PIECES);             // [kvol][256]: table entry of offset k for the tile's slot
  uint32_t* wg_mask = (uint32_t*)(sidx + kvol * kSortThreads);   // [2][NWAVES]
  float* ln_sums = (float*)(wg_mask + 2 * NWAVES);               // LNB: [NWAVES][2 NC] d gamma | d beta of the wave's rows so far
  if constexpr (LNB) {
    for (int i = threadIdx.x; i < NWAVES * 2 * NC; i += kSortThreads) ln_sums[i] = 0.f;   // (each wave clears and uses its own strip)
  }

  const OrderHdr o = *hdr;
  const int hb = __builtin_amdgcn_readfirstlane(o.heavy_blocks), mb = __builtin_amdgcn_readfirstlane(o.mid_blocks);
  const int dense_k = __builtin_amdgcn_readfirstlane(o.dense_k);
  // One block per wave (the 64-row tiles of the many-neighbour rows: ~all offsets each) is bound by LDS reads: every
  // wave reads the offset's whole weight slice for its 16 rows, 64 KB per offset and workgroup.  Such a tile is
  // therefore taken by TWO workgroups, each with one half of the output columns: half the slice staged and read, half
  // the matrix instructions, the same rows gathered twice (a few KB), every output element still summed by one lane in
  // ascending offset order -- the results do not change.
  // (never with the LayerNorm-backward epilogue: a row's statistics need all of its columns in one place)
  constexpr bool kCanSplit = !LNB && !LNF && CPW % 2 == 0 && CHUNKS % (2 * NWAVES) == 0 && NB % 4 == 0;
  // ... while such tiles are FEW: they are then the launch's tail on slots that would idle (the benchmark: 79 tiles on
  // 512 slots).  Where they are many, two workgroups per tile double their fixed cost (table batch, first operands) for
  // nothing: 2.5 pairs per row took 103 us split against 70 us unsplit.
  const int t_heavy1 = (o.b_heavy + hb - 1) / hb;
  const bool split = kCanSplit && hb == NWAVES && t_heavy1 * 4 <= (int)gridDim.x;
  const int t_heavy = split ? 2 * t_heavy1 : t_heavy1;            // tiles of the many-neighbour class (halves count)
  const int t_mid = t_heavy + (o.b_mid - o.b_heavy + mb - 1) / mb;
  const int tiles = __builtin_amdgcn_readfirstlane(t_mid + (o.b_total - o.b_mid + kLightBlocks - 1) / kLightBlocks);

  i32x4 frs;  // raw buffer descriptor: base, stride 0, size in bytes, 32-bit data format
  frs.x = (int)(uint32_t)(uintptr_t)feat;
  frs.y = (int)(uint32_t)((uintptr_t)feat >> 32);
  frs.z = (int)feat_bytes;
  frs.w = 0x00020000;

  auto swz = [](int row) -> int { return (row / RPB) & (PPR - 1); };
  auto chan = [](int cb, int i) -> int { return (cb >> 1) * 32 + (i >> 2) * 8 + (cb & 1) * 4 + (i & 3); };

#pragma unroll 1
  for (int round = 0;; ++round) {
    // (second round: the workgroups behind those of the long many-neighbour tiles go first -- they hold the
    // two-neighbour tiles, which end earliest and most evenly; reversed order paired the second tiles with the
    // 2-3-offset light tiles, whose first operands take up to 4 us longer)
    const int G_ = (int)gridDim.x;
    const int shift = t_heavy < G_ ? t_heavy : 0;
    const int t = round * G_ + ((round & 1) ? ((int)blockIdx.x + G_ - shift) % G_ : (int)blockIdx.x);
    if (round * (int)gridDim.x >= tiles) break;
    if (t >= tiles) continue;   // (only in the last round; the workgroup leaves as a whole)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));   // (thread coordinates re-derived per tile: nothing address-like is carried across)
    const int lane = tid_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int lrow = lane & 15;
    const int kg = lane >> 4;
    // tile -> first block and blocks per wave
    int blk0, bpw, half = 0;
    if (t < t_heavy) {
      blk0 = (split ? t >> 1 : t) * hb;
      half = split ? (t & 1) : 0;
      bpw = hb / NWAVES;
    } else if (t < t_mid) {
      blk0 = o.b_heavy + (t - t_heavy) * mb;
      bpw = mb / NWAVES;
    } else {
      blk0 = o.b_mid + (t - t_mid) * kLightBlocks;
      bpw = kLightBlocks / NWAVES;
    }
    const int blk_end = t < t_heavy ? o.b_heavy : (t < t_mid ? o.b_mid : o.b_total);
    SSTAMP(t, 0);
    const int wblk = blk0 + wave * bpw;   // this wave's first block
    const int64_t slot = (int64_t)wblk * 16 + lane;
    const bool have = lane < bpw * 16 && (wblk + (lane >> 4)) < blk_end && slot < n_out;
    i32x4_t my = {-1, 0, -1, -1};   // row, offset mask, table entries at the two lowest neighbour offsets
    if (have) my = rec[slot];
    const int32_t myrow = my.x;
    const uint32_t mymask = (uint32_t)my.y;
    // 16-lane OR -> block masks; wave OR -> workgroup mask
    uint32_t bm = mymask;
    bm |= __shfl_xor(bm, 1, 64);
    bm |= __shfl_xor(bm, 2, 64);
    bm |= __shfl_xor(bm, 4, 64);
    bm |= __shfl_xor(bm, 8, 64);
    uint32_t m[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) m[rb] = __builtin_amdgcn_readlane(bm, rb * 16);
    const uint32_t wm = m[0] | m[1] | m[2] | m[3];
    if (lane == 0) wg_mask[(round & 1) * NWAVES + wave] = wm;
    // the tile's table entries -> the wave's own strip of sidx (no barrier between these writes and its reads)
    {
      const uint32_t nbm = order_neighbours(mymask, dense_k);
      const bool many = __builtin_popcount(nbm) > 2;
      const int lo1 = nbm ? __builtin_ctz(nbm) : -1;
      auto short_entry = [&](int k) -> int32_t { return k == dense_k ? myrow : (k == lo1 ? my.z : my.w); };
      if (__builtin_amdgcn_ballot_w64(many) != 0ull) {
        // rows with 3+ neighbours: every offset of the kernel at once, ONE latency ("nothing there": a load of
        // table[0], which all such lanes share)
        int32_t e[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const bool on = many && k < kvol && ((mymask >> k) & 1u);
          e[k] = stream_load_i32(on ? table + (int64_t)k * n_out + myrow : table);
        }
        stream_wait_vm<0>();
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          stream_tie(e[k]);
          if (k < kvol && ((wm >> k) & 1u))
            sidx[k * kSortThreads + tid_] = ((mymask >> k) & 1u) ? (many ? e[k] : short_entry(k)) : -1;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 32; ++k)
          if (k < kvol && ((wm >> k) & 1u)) sidx[k * kSortThreads + tid_] = ((mymask >> k) & 1u) ? short_entry(k) : -1;
      }
    }
    __syncthreads();
    SSTAMP(t, 1);
    uint32_t rem = 0u;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) rem |= wg_mask[(round & 1) * NWAVES + w];
    rem = __builtin_amdgcn_readfirstlane(rem);
    SNOTE(t, 6, __builtin_popcount(rem));
    SNOTE(t, 7, blockIdx.x);
    auto next_k = [&]() -> int {
      if (!rem) return -1;
      const int k = __builtin_ctz(rem);
      rem &= rem - 1;
      return k;
    };

    auto stage_w = [&](int k, int buf, auto cs_c) {
      constexpr int CS = decltype(cs_c)::value;   // 2: only this workgroup's half of the columns (chunks are whole weight rows)
      const int kk = k < 0 ? 0 : k;
#pragma unroll
      for (int u = 0; u < CPW / CS; ++u) {
        int c = (CS == 2 ? half * (CHUNKS / 2) : 0) + u * NWAVES + wave;
        if (CHUNKS % NWAVES != 0 && c >= CHUNKS) c = c % CHUNKS;
        const int q = c * 64 + lane;
        const int row = q / PPR, slot_ = q % PPR;
        const uint16_t* src = wn + (int64_t)kk * NC * KD + (row * PPR + (slot_ ^ swz(row))) * 8;
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(wl + buf * PIECES + c * 64);
        stream_dma_b128(src, __builtin_amdgcn_readfirstlane(dst));
      }
    };
    uint32_t wb[2], wh[2];
#pragma unroll
    for (int c1 = 0; c1 < 2; ++c1) {
      const int row = chan(c1, lrow);
      const int sw = swz(row);
      wb[c1] = (uint32_t)(row * PPR + (kg ^ (sw & 3))) * 16u;
      wh[c1] = (uint32_t)(sw & ~3) * 16u;
    }

    // the walk for a wave of BPW blocks: groups of G = 4 / BPW offsets; virtual block v = j * BPW + b is block b at
    // the group's j-th offset
    auto walk = [&](auto bpw_c, auto cs_c) {
      constexpr int BPW = decltype(bpw_c)::value;
      constexpr int CS = decltype(cs_c)::value;      // workgroups that share the tile's columns
      constexpr int NBL = NB / CS, CPWL = CPW / CS;  // column blocks and weight chunks per wave of THIS workgroup
      constexpr int G = 4 / BPW;
      const int cb0 = CS == 2 ? half * NBL : 0;      // first column block
      f32x4 acc[BPW][NBL];
#pragma unroll
      for (int b = 0; b < BPW; ++b)
#pragma unroll
        for (int cb = 0; cb < NBL; ++cb) acc[b][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      int K[2 * G + 1];   // offsets of this group, of the next, and one more (the weights run two offsets ahead)
#pragma unroll
      for (int i = 0; i < 2 * G + 1; ++i) K[i] = next_k();

      auto gather = [&](u32x4 (&x)[4][KSTEPS], int base) {
        int32_t ibv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int k = K[base + v / BPW];
          ibv[v] = sidx[(k < 0 ? 0 : k) * kSortThreads + wave * 64 + (v % BPW) * 16 + lrow];
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int k = K[base + v / BPW];
          const bool on = k >= 0 && ((m[v % BPW] >> (k & 31)) & 1u) && ibv[v] >= 0;
          const uint32_t off = on ? (uint32_t)ibv[v] * (KD * 2) + kg * 16 : 0xffffff00u;
          x[v][0] = stream_buffer_load<0>(frs, off);
          if constexpr (KSTEPS > 1) x[v][1] = stream_buffer_load<64>(frs, off);
          if constexpr (KSTEPS > 2) x[v][2] = stream_buffer_load<128>(frs, off);
          if constexpr (KSTEPS > 3) x[v][3] = stream_buffer_load<192>(frs, off);
        }
      };
      auto keep_rows = [&](const u32x4 (&x)[4][KSTEPS]) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) stream_keep(x[v][ks]);
      };
      auto mma = [&](u32x4 (&x)[4][KSTEPS], int j, int buf, int k) {
#pragma unroll
        for (int b = 0; b < BPW; ++b)
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) stream_tie(x[j * BPW + b][ks]);
        if (k < 0) return;
        stream_tie(wb[0]);
        stream_tie(wb[1]);
        stream_tie(wh[0]);
        stream_tie(wh[1]);
        uint32_t act = 0;
#pragma unroll
        for (int b = 0; b < BPW; ++b) act |= ((m[b] >> (k & 31)) & 1u) << b;
        if (!act) return;
        const char* wbuf = (const char*)(wl + buf * PIECES);
        constexpr int FG = (NBL >= 8) ? 1 : ((8 / NBL) < KSTEPS ? (8 / NBL) : KSTEPS);
#pragma unroll
        for (int k0 = 0; k0 < KSTEPS; k0 += FG) {
          bf16x8 w[FG][NBL];
#pragma unroll
          for (int f = 0; f < FG; ++f)
#pragma unroll
            for (int cb = 0; cb < NBL; ++cb) {
              // (cb0 is even: the pair (cb & 1) and the 32-row group (cb >> 1) of the global column block cb0 + cb)
              const uint32_t a = wb[cb & 1] + (uint32_t)(((cb0 + cb) >> 1) * 32 * PPR * 16) + ((uint32_t)((k0 + f) * 64) ^ wh[cb & 1]);
              w[f][cb] = __builtin_bit_cast(bf16x8, *(const u32x4*)(wbuf + a));
            }
#pragma unroll
          for (int b = 0; b < BPW; ++b) {
            if ((act >> b) & 1u) {
#pragma unroll
              for (int f = 0; f < FG; ++f)
#pragma unroll
                for (int cb = 0; cb < NBL; ++cb)
                  acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[f][cb], __builtin_bit_cast(bf16x8, x[j * BPW + b][k0 + f]),
                                                                       acc[b][cb], 0, 0, 0);
            }
          }
        }
      };

      u32x4 xa[4][KSTEPS], xb[4][KSTEPS];
      if (BPW == 4 && K[2] < 0) {
        // at most two offsets (most tiles of rows with 0 or 1 neighbours): everything is asked for at once, no loop
        stage_w(K[0], 0, cs_c);
        gather(xa, 0);
        if (K[1] >= 0) {
          stage_w(K[1], 1, cs_c);
          gather(xb, 1);
        }
        wait_vmcnt_barrier<0>();
        SSTAMP(t, 2);
        mma(xa, 0, 0, K[0]);
        if (K[1] >= 0) mma(xb, 0, 1, K[1]);
      } else {
        int bc = 0, bn = 1, b2 = 2;   // weight buffers of the offset at hand, the next, the one after
        stage_w(K[0], bc, cs_c);
        stage_w(K[1], bn, cs_c);
        gather(xa, 0);
        wait_vmcnt_barrier<0>();
        SSTAMP(t, 2);
        // one group: the next group's rows are asked for first, then offset by offset: weights two ahead, wait for this
        // wave's share of the NEXT offset's weights (and, older, this offset's rows), multiply, barrier (every wave's
        // share of the next offset's weights is in; buffer bc is free)
        auto group = [&](u32x4 (&xc)[4][KSTEPS], u32x4 (&xn)[4][KSTEPS]) {
          gather(xn, G);
#pragma unroll
          for (int j = 0; j < G; ++j) {
            stage_w(K[j + 2], b2, cs_c);
            if (j == 0) {
              stream_wait_vm<CPWL + NG>();
            } else {
              stream_wait_vm<CPWL>();
            }
            mma(xc, j, bc, K[j]);
            if (j == 0) {
              wait_vmcnt_barrier<CPWL + NG>();
            } else {
              wait_vmcnt_barrier<CPWL>();
            }
            const int b0 = bc;
            bc = bn;
            bn = b2;
            b2 = b0;
          }
#pragma unroll
          for (int i = 0; i <= G; ++i) K[i] = K[i + G];
#pragma unroll
          for (int i = G + 1; i < 2 * G + 1; ++i) K[i] = next_k();
        };
        while (K[0] >= 0) {   // ONE exit; the second group of a trip may be all padding (its loads are clamped, its MFMAs skipped)
          group(xa, xb);
          group(xb, xa);
        }
        stream_wait_vm<0>();
      }
      keep_rows(xa);
      keep_rows(xb);
      SSTAMP(t, 3);

      // ---- epilogue: lane holds channels 32p + 8kg .. +7 of slot lrow of block b in acc[b][2p], acc[b][2p+1] ----
      if constexpr (LNB) {
        static_assert(CS == 1 || !LNB, "the LayerNorm-backward epilogue needs whole rows");
        constexpr int NP = NB / 2;   // 8-channel pieces per lane: piece (p, kg) is the tile kernel's c8 = 4 p + kg
        // everything the rows need is asked for at once: the block's conv output and statistics, gamma, beta
        int32_t rr[BPW];
        u32x4 xin[BPW][NP];
        float mean[BPW], rstd[BPW];
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
          rr[b] = __shfl(myrow, b * 16 + lrow, 64);
          const int64_t rc = rr[b] < 0 ? 0 : rr[b];
#pragma unroll
          for (int p = 0; p < NP; ++p) xin[b][p] = *(const u32x4*)(ln.y + rc * NC + p * 32 + kg * 8);
          mean[b] = ln.mean_rstd[rc * 2];
          rstd[b] = ln.mean_rstd[rc * 2 + 1];
        }
        ln_f32x2 g[NP][4], bt[NP][4], dg[NP][4], db[NP][4];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const f32x4 g0 = *(const f32x4*)(ln.gamma + p * 32 + kg * 8), g1 = *(const f32x4*)(ln.gamma + p * 32 + kg * 8 + 4);
          const f32x4 b0 = *(const f32x4*)(ln.beta + p * 32 + kg * 8), b1 = *(const f32x4*)(ln.beta + p * 32 + kg * 8 + 4);
          g[p][0] = ln_f32x2{g0.x, g0.y}; g[p][1] = ln_f32x2{g0.z, g0.w}; g[p][2] = ln_f32x2{g1.x, g1.y}; g[p][3] = ln_f32x2{g1.z, g1.w};
          bt[p][0] = ln_f32x2{b0.x, b0.y}; bt[p][1] = ln_f32x2{b0.z, b0.w}; bt[p][2] = ln_f32x2{b1.x, b1.y}; bt[p][3] = ln_f32x2{b1.z, b1.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) dg[p][q] = db[p][q] = ln_f32x2{0.f, 0.f};
        }
        auto rb16 = [](float v) -> float { return ococc_bf16_to_f32(ococc_f32_to_bf16(v)); };
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
          if (rr[b] < 0) continue;   // (the four lanes of a slot agree: the exchanges below stay inside a slot)
          ln_f32x2 xv[NP][4], dzg[NP][4];
          float t1[NP], t2[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const f32x4 v0 = acc[b][2 * p], v1 = acc[b][2 * p + 1];
            ln_f32x2 dv[4];
            // (the incoming gradient rounded to bf16 first, as the unfused pair of kernels sees it)
            dv[0] = ln_f32x2{rb16(v0.x), rb16(v0.y)};
            dv[1] = ln_f32x2{rb16(v0.z), rb16(v0.w)};
            dv[2] = ln_f32x2{rb16(v1.x), rb16(v1.y)};
            dv[3] = ln_f32x2{rb16(v1.z), rb16(v1.w)};
            ln_unpack8(xin[b][p], xv[p]);
            if (ln.act == 1) ln_bwd_piece8<true>(xv[p], dv, mean[b], rstd[b], g[p], bt[p], dg[p], db[p], dzg[p], t1[p], t2[p]);
            else ln_bwd_piece8<false>(xv[p], dv, mean[b], rstd[b], g[p], bt[p], dg[p], db[p], dzg[p], t1[p], t2[p]);
          }
          // the row's two sums over its NC / 8 pieces, in the butterfly order of the other LayerNorm-backward kernels
          // (piece c8 with c8 ^ NC/16, ..., c8 ^ 1): the high bits of c8 are p (in this lane), the low two are kg
          float s1, s2;
          if constexpr (NP == 4) {
            s1 = (t1[0] + t1[2]) + (t1[1] + t1[3]);
            s2 = (t2[0] + t2[2]) + (t2[1] + t2[3]);
          } else if constexpr (NP == 2) {
            s1 = t1[0] + t1[1];
            s2 = t2[0] + t2[1];
          } else {
            s1 = t1[0];
            s2 = t2[0];
          }
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          s1 += __shfl_xor(s1, 16, 64);
          s2 += __shfl_xor(s2, 16, 64);
#pragma unroll
          for (int p = 0; p < NP; ++p)
            *(u32x4*)((uint16_t*)out_ + (int64_t)rr[b] * NC + p * 32 + kg * 8) =
                ln_bwd_finish8(xv[p], dzg[p], rstd[b], s1 * (1.f / NC), s2 * (1.f / NC));
        }
        // d gamma / d beta of the tile's rows: over the 16 slots of a lane group, then into this wave's strip
        float* mine = ln_sums + wave * 2 * NC;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a0 = dg[p][q].x, a1 = dg[p][q].y, c0 = db[p][q].x, c1 = db[p][q].y;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
              a0 += __shfl_xor(a0, d, 64);
              a1 += __shfl_xor(a1, d, 64);
              c0 += __shfl_xor(c0, d, 64);
              c1 += __shfl_xor(c1, d, 64);
            }
            if (lrow == 0) {
              const int ch = p * 32 + kg * 8 + 2 * q;
              mine[ch] += a0;
              mine[ch + 1] += a1;
              mine[NC + ch] += c0;
              mine[NC + ch + 1] += c1;
            }
          }
        return;
      }
      if constexpr (LNF) {
        // LayerNorm (+ GELU) forward on the finished rows.  The norm sees the bf16-rounded conv output, as the unfused pair
        // of launches does (conv store, ln_act_fwd_vec_kernel load), and the arithmetic is that kernel's, sums included:
        // there a lane holds ONE 8-channel piece c8 and the row sums run over the pieces as a butterfly (c8 ^ 8, ^ 4, ^ 2,
        // ^ 1); here a lane holds the pieces c8 = 4 p + kg, p = 0..3 -- the same butterfly is (p ^ 2), (p ^ 1) inside
        // the lane, then kg ^ 2 (lane ^ 32), kg ^ 1 (lane ^ 16).
        static_assert(CS == 1 || !LNF, "the LayerNorm epilogue needs whole rows");
        constexpr int NP = NB / 2;
        ln_f32x2 g[NP][4], bt[NP][4];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const f32x4 g0 = *(const f32x4*)(ln.gamma + p * 32 + kg * 8), g1 = *(const f32x4*)(ln.gamma + p * 32 + kg * 8 + 4);
          const f32x4 b0 = *(const f32x4*)(ln.beta + p * 32 + kg * 8), b1 = *(const f32x4*)(ln.beta + p * 32 + kg * 8 + 4);
          g[p][0] = ln_f32x2{g0.x, g0.y}; g[p][1] = ln_f32x2{g0.z, g0.w}; g[p][2] = ln_f32x2{g1.x, g1.y}; g[p][3] = ln_f32x2{g1.z, g1.w};
          bt[p][0] = ln_f32x2{b0.x, b0.y}; bt[p][1] = ln_f32x2{b0.z, b0.w}; bt[p][2] = ln_f32x2{b1.x, b1.y}; bt[p][3] = ln_f32x2{b1.z, b1.w};
        }
        auto over_pieces = [&](const float (&t)[NP]) -> float {
          float s;
          if constexpr (NP == 4) s = (t[0] + t[2]) + (t[1] + t[3]);
          else if constexpr (NP == 2) s = t[0] + t[1];
          else s = t[0];
          s += __shfl_xor(s, 32, 64);
          s += __shfl_xor(s, 16, 64);
          return s;
        };
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
          const int32_t r = __shfl(myrow, b * 16 + lrow, 64);
          if (r < 0) continue;   // (the four lanes of a slot agree: the exchanges below stay inside a slot)
          ln_f32x2 v[NP][4];
          float t[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const f32x4 v0 = acc[b][2 * p], v1 = acc[b][2 * p + 1];
            u32x4 q;
            q.x = ococc_pack_bf16x2(v0.x, v0.y);
            q.y = ococc_pack_bf16x2(v0.z, v0.w);
            q.z = ococc_pack_bf16x2(v1.x, v1.y);
            q.w = ococc_pack_bf16x2(v1.z, v1.w);
            *(u32x4*)((uint16_t*)out_ + (int64_t)r * NC + p * 32 + kg * 8) = q;
            ln_unpack8(q, v[p]);
            const ln_f32x2 sv = (v[p][0] + v[p][1]) + (v[p][2] + v[p][3]);
            t[p] = sv.x + sv.y;
          }
          const float mean = over_pieces(t) * (1.f / NC);
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            ln_f32x2 sq = {0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[p][c] = v[p][c] - mean;
              sq += v[p][c] * v[p][c];
            }
            t[p] = sq.x + sq.y;
          }
          const float rstd = rsqrtf(over_pieces(t) * (1.f / NC) + ln.eps);
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            u32x4 q;
            ln_f32x2 z[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              z[c] = (v[p][c] * rstd) * g[p][c] + bt[p][c];
              if (ln.act == 1) z[c] = ln_gelu2(z[c]);
            }
            q.x = ln_pack2(z[0]);
            q.y = ln_pack2(z[1]);
            q.z = ln_pack2(z[2]);
            q.w = ln_pack2(z[3]);
            *(u32x4*)(ln.act_out + (int64_t)r * NC + p * 32 + kg * 8) = q;
          }
          if (kg == 0) {
            ln.stats_out[(int64_t)r * 2] = mean;
            ln.stats_out[(int64_t)r * 2 + 1] = rstd;
          }
        }
        return;
      }
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        const int32_t r = __shfl(myrow, b * 16 + lrow, 64);
        if (r < 0) continue;
#pragma unroll
        for (int p = 0; p < NBL / 2; ++p) {
          const int ch = (cb0 / 2 + p) * 32 + kg * 8;
          f32x4 v0 = acc[b][2 * p], v1 = acc[b][2 * p + 1];
          if (bias) {
            v0 += *(const f32x4*)(bias + ch);
            v1 += *(const f32x4*)(bias + ch + 4);
          }
          if (OUT_BF16) {
            u32x4 q;
            q.x = (uint32_t)ococc_f32_to_bf16(v0.x) | ((uint32_t)ococc_f32_to_bf16(v0.y) << 16);
            q.y = (uint32_t)ococc_f32_to_bf16(v0.z) | ((uint32_t)ococc_f32_to_bf16(v0.w) << 16);
            q.z = (uint32_t)ococc_f32_to_bf16(v1.x) | ((uint32_t)ococc_f32_to_bf16(v1.y) << 16);
            q.w = (uint32_t)ococc_f32_to_bf16(v1.z) | ((uint32_t)ococc_f32_to_bf16(v1.w) << 16);
            *(u32x4*)((uint16_t*)out_ + (int64_t)r * NC + ch) = q;
          } else {
            *(f32x4*)((float*)out_ + (int64_t)r * NC + ch) = v0;
            *(f32x4*)((float*)out_ + (int64_t)r * NC + ch + 4) = v1;
          }
        }
      }
    };
    if (split && t < t_heavy) {   // (bpw == 1 there; a 4-block plan for the two-neighbour class has bpw == 1 too, unsplit)
      if constexpr (kCanSplit) walk(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
    } else if (bpw == 1) {
      walk(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    } else if (bpw == 2) {
      walk(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
    } else {
      walk(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
    }
    SSTAMP(t, 4);
  }
  if constexpr (LNB) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NC; i += kSortThreads) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) sum += ln_sums[w * 2 * NC + i];
      ln.partials[(int64_t)blockIdx.x * 2 * NC + i] = sum;
    }
  }
}

inline unsigned sorted_grid(int64_t n_out) {
  // persistent workgroups, two per CU; never more than the 64-row tiles there could be
  const int64_t most = ococc_cdiv(n_out, 64) + 3;
  return (unsigned)(most < 512 ? most : 512);
}

template <int KD, int NC>
int launch_sorted(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, const int32_t* table, const int32_t* rec,
                  const OrderHdr* hdr, int64_t n_out, const float* bias, void* out, int out_dtype, hipStream_t stream,
                  const SortedLn* ln = nullptr) {
  const dim3 grid(sorted_grid(n_out));
  if (ln && ln->act_out) {   // LayerNorm (+ GELU) forward epilogue
    const int lds = sorted_lds_bytes<KD, NC>(kvol);
    auto kern = gather_gemm_sorted_kernel<KD, NC, true, false, true>;
    OCOCC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, grid, dim3(kSortThreads), lds, stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, table,
                       (const i32x4_t*)rec, hdr, n_out, bias, out, *ln);
    OCOCC_CHECK_LAUNCH();
    return OCOCC_OK;
  }
  if (ln) {
    if constexpr (NC <= 64) {
      const int lds = sorted_lds_bytes<KD, NC, true>(kvol);
      auto kern = gather_gemm_sorted_kernel<KD, NC, true, true>;
      OCOCC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL(kern, grid, dim3(kSortThreads), lds, stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, table,
                         (const i32x4_t*)rec, hdr, n_out, bias, out, *ln);
      OCOCC_CHECK_LAUNCH();
      return OCOCC_OK;
    } else {
      return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "fused LayerNorm backward epilogue: 32 or 64 output columns");
    }
  }
  const int lds = sorted_lds_bytes<KD, NC>(kvol);
  auto kern = out_dtype == OCOCC_BF16 ? gather_gemm_sorted_kernel<KD, NC, true> : gather_gemm_sorted_kernel<KD, NC, false>;
  OCOCC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, grid, dim3(kSortThreads), lds, stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, table,
                     (const i32x4_t*)rec, hdr, n_out, bias, out, SortedLn{});
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

template <int KD>
int dispatch_sorted(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, int ncols, const int32_t* table,
                    const int32_t* rec, const OrderHdr* hdr, int64_t n_out, const float* bias, void* out, int out_dtype,
                    hipStream_t stream, const SortedLn* ln = nullptr) {
  switch (ncols) {
    case 32: return launch_sorted<KD, 32>(feat, n_in, wn, kvol, table, rec, hdr, n_out, bias, out, out_dtype, stream, ln);
    case 64: return launch_sorted<KD, 64>(feat, n_in, wn, kvol, table, rec, hdr, n_out, bias, out, out_dtype, stream, ln);
    case 128:
      if constexpr (KD <= 64) return launch_sorted<KD, 128>(feat, n_in, wn, kvol, table, rec, hdr, n_out, bias, out, out_dtype, stream, ln);
      break;
    default: break;
  }
  return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "kd x ncols must be one of {32,64,128} x {32,64,128} below 128 x 128");
}

constexpr int kOrderCounterBytes = (kOrderCounterWords * 4 + 15) / 16 * 16;

int launch_order_place(const i32x4_t* rowrec, int64_t n, int heavy_blocks, int mid_blocks, int dense_k, uint32_t* hist,
                       int32_t* rec, int32_t* hdr, hipStream_t stream) {
  const unsigned wgs = (unsigned)ococc_cdiv(n, kOrderRowsPerWg);
  hipLaunchKernelGGL(order_place_kernel, dim3(wgs > 0 ? wgs : 1), dim3(256), 0, stream, rowrec, n, (const uint32_t*)hist,
                     heavy_blocks, mid_blocks, dense_k, (i32x4_t*)rec, (OrderHdr*)hdr);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

int check_order_args(int32_t kvol, int32_t dense_k, int64_t n, int32_t heavy_blocks, int32_t mid_blocks, const void* counters,
                     const void* scratch, const void* rec, const void* hdr) {
  OCOCC_REQUIRE(n >= 0, "negative row count");
  if (n >= kOrderMaxRows) return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "the row records hold places below 2^20 rows");
  OCOCC_REQUIRE(kvol >= 1 && kvol <= 32, "kernel volume must be 1..32");
  OCOCC_REQUIRE(dense_k >= -1 && dense_k < kvol, "dense_k out of range");
  OCOCC_REQUIRE((heavy_blocks == 4 || heavy_blocks == 8 || heavy_blocks == 16) &&
                (mid_blocks == 4 || mid_blocks == 8 || mid_blocks == 16), "tile sizes must be 4, 8 or 16 blocks");
  OCOCC_REQUIRE(hdr && counters, "null pointer");
  OCOCC_REQUIRE(n == 0 || (rec && scratch), "null pointer");
  OCOCC_REQUIRE(((uintptr_t)scratch & 15) == 0 && ((uintptr_t)rec & 15) == 0, "scratch and rec must be 16-byte aligned");
  return OCOCC_OK;
}

}  // namespace

extern "C" int64_t ococc_subm_row_order_counter_bytes(void) { return kOrderCounterBytes; }

extern "C" int64_t ococc_subm_row_order_scratch_bytes(int64_t n) { return n * 16; }   // per-row records

extern "C" int ococc_subm_row_order(const int32_t* table, int32_t kvol, int32_t dense_k, int64_t n, int32_t heavy_blocks,
                                    int32_t mid_blocks, void* counters, void* scratch, int32_t* rec, int32_t* hdr,
                                    ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = check_order_args(kvol, dense_k, n, heavy_blocks, mid_blocks, counters, scratch, rec, hdr);
  if (rc != OCOCC_OK) return rc;
  OCOCC_REQUIRE(n == 0 || table, "null pointer");
  OCOCC_HIP(hipMemsetAsync(counters, 0, kOrderCounterBytes, stream));
  if (n > 0) {
    hipLaunchKernelGGL(order_count_kernel, dim3((unsigned)ococc_cdiv(n, kOrderRowsPerWg)), dim3(256), 0, stream, table,
                       (int)kvol, (int)dense_k, n, (i32x4_t*)scratch, (uint32_t*)counters);
    OCOCC_CHECK_LAUNCH();
  }
  return launch_order_place((const i32x4_t*)scratch, n, heavy_blocks, mid_blocks, dense_k, (uint32_t*)counters, rec, hdr, stream);
}

extern "C" int ococc_subm_row_order_place(const int32_t* rowrec, int32_t kvol, int32_t dense_k, int64_t n,
                                          int32_t heavy_blocks, int32_t mid_blocks, void* counters, int32_t* rec,
                                          int32_t* hdr, ococc_stream_t stream_) {
  const int rc = check_order_args(kvol, dense_k, n, heavy_blocks, mid_blocks, counters, rowrec, rec, hdr);
  if (rc != OCOCC_OK) return rc;
  return launch_order_place((const i32x4_t*)rowrec, n, heavy_blocks, mid_blocks, dense_k, (uint32_t*)counters, rec, hdr,
                            (hipStream_t)stream_);
}

namespace {
int sorted_entry(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn, int32_t kvol, int32_t ncols,
                 const int32_t* table, const int32_t* rec, const int32_t* hdr, int64_t n_out, const float* bias, void* out,
                 int32_t out_dtype, hipStream_t stream, const SortedLn* ln) {
  OCOCC_REQUIRE(n_in >= 0 && n_out >= 0, "negative row count");
  OCOCC_REQUIRE(kvol >= 1 && kvol <= 32, "kernel volume must be 1..32");
  OCOCC_REQUIRE(out_dtype == OCOCC_BF16 || out_dtype == OCOCC_F32, "out_dtype must be f32/bf16");
  if (n_out == 0) return OCOCC_OK;
  OCOCC_REQUIRE(wn && table && out && rec && hdr, "null pointer");
  OCOCC_REQUIRE(feat || n_in == 0, "null feat");
  OCOCC_REQUIRE(n_in * kd * 2 < 0xffffff00ll, "feat must stay below 4 GB (32-bit buffer offsets)");
  switch (kd) {
    case 32: return dispatch_sorted<32>(feat, n_in, wn, kvol, ncols, table, rec, (const OrderHdr*)hdr, n_out, bias, out, out_dtype, stream, ln);
    case 64: return dispatch_sorted<64>(feat, n_in, wn, kvol, ncols, table, rec, (const OrderHdr*)hdr, n_out, bias, out, out_dtype, stream, ln);
    case 128: return dispatch_sorted<128>(feat, n_in, wn, kvol, ncols, table, rec, (const OrderHdr*)hdr, n_out, bias, out, out_dtype, stream, ln);
    default: return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "kd must be 32/64/128");
  }
}
}  // namespace

extern "C" int ococc_sparse_conv_sorted_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn,
                                             int32_t kvol, int32_t ncols, const int32_t* table, const int32_t* rec,
                                             const int32_t* hdr, int64_t n_out, const float* bias, void* out,
                                             int32_t out_dtype, ococc_stream_t stream_) {
  return sorted_entry(feat, n_in, kd, wn, kvol, ncols, table, rec, hdr, n_out, bias, out, out_dtype, (hipStream_t)stream_,
                      nullptr);
}

extern "C" int ococc_sparse_conv_sorted_ln_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn,
                                                int32_t kvol, int32_t ncols, const int32_t* table, const int32_t* rec,
                                                const int32_t* hdr, int64_t n_out, const float* gamma, const float* beta,
                                                float eps, int32_t act, uint16_t* conv_out, uint16_t* y, float* mean_rstd,
                                                ococc_stream_t stream_) {
  OCOCC_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (gelu)");
  OCOCC_REQUIRE(n_out == 0 || (gamma && beta && conv_out && y && mean_rstd), "null pointer");
  SortedLn ln{};
  ln.gamma = gamma;
  ln.beta = beta;
  ln.act = act;
  ln.eps = eps;
  ln.act_out = y;
  ln.stats_out = mean_rstd;
  return sorted_entry(feat, n_in, kd, wn, kvol, ncols, table, rec, hdr, n_out, nullptr, conv_out, OCOCC_BF16,
                      (hipStream_t)stream_, &ln);
}

extern "C" int64_t ococc_sparse_conv_sorted_lnbwd_partial_rows(int64_t n_out) {
  return n_out < 0 ? -1 : (int64_t)sorted_grid(n_out > 0 ? n_out : 1);
}

extern "C" int ococc_sparse_conv_sorted_lnbwd_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn,
                                                   int32_t kvol, int32_t ncols, const int32_t* table, const int32_t* rec,
                                                   const int32_t* hdr, int64_t n_out, const uint16_t* block_conv_out,
                                                   const float* mean_rstd, const float* gamma, const float* beta,
                                                   int32_t act, uint16_t* d_conv_out, float* partials,
                                                   int64_t partial_rows, ococc_stream_t stream_) {
  OCOCC_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (gelu)");
  OCOCC_REQUIRE(ncols == 32 || ncols == 64, "fused LayerNorm backward: 32 or 64 output columns");
  OCOCC_REQUIRE(n_out == 0 || (gamma && beta && block_conv_out && mean_rstd && d_conv_out && partials), "null pointer");
  OCOCC_REQUIRE(partial_rows >= ococc_sparse_conv_sorted_lnbwd_partial_rows(n_out), "partials too small");
  const SortedLn ln{block_conv_out, mean_rstd, gamma, beta, act, partials, 0.f, nullptr, nullptr};
  return sorted_entry(feat, n_in, kd, wn, kvol, ncols, table, rec, hdr, n_out, nullptr, d_conv_out, OCOCC_BF16,
                      (hipStream_t)stream_, &ln);
}

#ifdef OCOCC_SORTED_STAMPS
extern "C" int ococc_sorted_set_stamps(long long* dev_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(s_stamps), &dev_buffer, sizeof(dev_buffer)) == hipSuccess ? 0 : -1;
}
#endif
