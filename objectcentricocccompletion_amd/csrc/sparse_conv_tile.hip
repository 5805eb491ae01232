// B4  sub-manifold convolution for SPARSE active sets: compact-then-multiply.
// Same contract as ococc_sparse_conv_gather_gemm_bf16 (out[o] = sum_k feat[table[k][o]] @ W[k],
// replaces indiceConv / indiceConvBackward, spconv_ops.h:260-456) for sub-manifold tables.
//
// Why a second kernel.  The output-stationary kernels in sparse_conv.hip give every wave 64 output
// rows and walk the kernel offsets; at offset k a 16-row MFMA block is issued if ANY of its rows has
// a neighbour.  On the benchmark's grids (2000 random points in 40^3 cells) a voxel has 0.75
// neighbours besides itself: a 16-row block has a neighbour at a given offset 37 % of the time and
// then carries 1.3 useful rows, i.e. 8 % of the MFMA work and of the weight traffic behind it is
// useful, and every wave re-reads every 16 KB weight slice from LDS for it.
//
// Here a workgroup (8 waves, one per CU) owns a tile of T output rows (512, or 256 for 128 output
// columns) with the f32 accumulators in LDS.  The centre offset (every row is its own neighbour) is a
// dense pass split over the waves.  Every other offset is taken by ONE wave: it reads the offset's
// table column for the whole tile, ranks the rows that have a neighbour with wave ballots (typically
// 15 of 512), gathers only those rows, multiplies them as full 16-row MFMA blocks by weight fragments
// it loaded straight from L2 into registers (each weight slice is read once per tile, by the one wave
// that needs it: no LDS staging, no per-offset barrier on the weights), and adds the products into
// the tile.  A wave carries four offsets at a time (table columns, first blocks' rows and the next
// offset's weights are all loaded ahead), eight waves work on eight offsets per round; their
// additions into the shared tile are taken in wave order with a barrier in between, so the sum order
// per output element is fixed (centre, then ascending offsets) and the result is deterministic, like
// the other kernels.
//
// Any table is handled correctly; on DENSE neighbourhoods the blocks past an offset's first and
// their ordered additions serialise and the output-stationary kernels are the better choice.
// Measured on configs[1] (126 k rows, 1.76 pairs per row; mean 15 rows per offset and tile, so the rows of
// the first TWO blocks of every offset are gathered ahead -- with one, a third of the offsets fell to the
// block-by-block path and stalled their round): 128 -> 64 channels 29.7 us against 50.1 us for the
// streamed-weights kernel; 64 -> 32 19.7 us and 32 -> 64 21.1 us against 21.3 / 21.2 us for the
// resident-weights kernel; 64 -> 128 (256-row tiles, two rounds of workgroups, spills) 65 us against 40 us.
// With one workgroup per CU every phase exposes its load latency: shapes up to 64 x 64 channels use 256-row tiles
// with ONE block per offset gathered ahead (mean 7.5 rows per offset and tile), fit 128 registers, and run two
// workgroups per CU (64 -> 32 16.5 us, 32 -> 64 17.6 us).  The host selects the kernel per shape
// (spconv/ops.py), or when told to.
#include "common.hpp"
#include "ln_math.hpp"

namespace {

#ifdef OCOCC_TILE_STAMPS
// diagnostic build only (tools/probe/tile_stamps.py): wall-clock stamps per workgroup and phase into a buffer of their own
__device__ long long* t_stamps = nullptr;
#define TSTAMP(slot) do { if (threadIdx.x == 0 && t_stamps) t_stamps[(int64_t)blockIdx.x * 16 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TSTAMP(slot) do { } while (0)
#endif

constexpr int kTileThreads = 512;
#ifndef OCOCC_TILE_LNB_ROWS
#define OCOCC_TILE_LNB_ROWS 512
#endif
// rows per tile of the 128-gathered-channel instantiation WITH the LayerNorm-backward epilogue
constexpr int kTileLnbRows = OCOCC_TILE_LNB_ROWS;
constexpr int kTileWaves = kTileThreads / 64;
constexpr int kTileOffsetsPerWave = 4;  // offsets a wave carries through one pass (8 x 4 = 32 >= 26)

// LayerNorm (+ GELU) of the enclosing conv -> norm -> act block in the epilogue: the finished f32 row sits in LDS, so
// the statistics cost three lane exchanges and the separate LN launch (one read of the conv output) disappears.
// As in the unfused pair of kernels the norm sees the bf16-rounded conv output.
struct TileLn {
  const float* gamma;
  const float* beta;
  float eps;
  int act;           // 0 none, 1 GELU(erf)
  uint16_t* y;       // [n_out, NC] bf16
  float* mean_rstd;  // [n_out, 2]
  // LayerNorm BACKWARD epilogue (LNB instantiations, the dgrad of the NEXT layer): y = this layer's conv output (read),
  // mean_rstd = its saved statistics (read), partials = [gridDim.x][2 NC] per-workgroup sums for d gamma | d beta
  float* partials;
};
__device__ __forceinline__ float tile_gelu(float z) { return ln_gelu1(z); }  // (ln_math.hpp: the GELU of every kernel here)
__device__ __forceinline__ float tile_round_bf16(float v) { return ococc_bf16_to_f32(ococc_f32_to_bf16(v)); }

template <int KD, int NC, int T, bool OUT_BF16, bool LN = false, bool LNB = false>
__global__ void __launch_bounds__(kTileThreads, (T <= 256 && NC <= 64) ? 4 : 1)
subm_tile_conv_kernel(const uint16_t* __restrict__ feat, uint32_t feat_bytes, const uint16_t* __restrict__ wn,
                      int kvol, int dense_k, const int32_t* __restrict__ table, int64_t n_out,
                      const float* __restrict__ bias, void* __restrict__ out_, TileLn ln) {
  constexpr int KSTEPS = KD / 32, NB = NC / 16, LDT = NC + 4, U = T / 64;
  constexpr int NW = kTileWaves, MAXO = (T <= 256 && KD >= 128) ? 2 : kTileOffsetsPerWave, DB = T / 16 / NW;  // DB: dense blocks per wave
  static_assert(KD % 32 == 0 && NC % 16 == 0 && T % (16 * NW) == 0, "tile kernel shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;  // [T][LDT] f32 accumulators
  // the 16 (input row, tile row) pairs of the MFMA block a wave is about to multiply, per offset slot
  constexpr int FB = (T <= 256 && NC <= 64) ? 1 : 2;  // blocks per offset whose rows are gathered ahead (mean 15 rows per offset and tile: a second
                          // block for a third of the offsets; past FB blocks an offset goes block by block)
  __shared__ int32_t sl_in[NW][MAXO][16 * FB];
  __shared__ uint16_t sl_row[NW][MAXO][16 * FB];
  __shared__ int s_cnt[NW];
  __shared__ uint32_t s_owner[T];  // row -> tag of the wave whose addition is next (see the additions below)
  __shared__ int s_more[2];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lrow = lane & 15, kg = lane >> 4;
  // XCD-aware: workgroups are dealt round-robin to the 8 XCDs; give each a contiguous eighth of the tiles
  const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t row0 = (int64_t)wg * T;
  if (row0 >= n_out) {
    if constexpr (LNB)   // (a padding workgroup still owns a row of the partial sums)
      for (int i = threadIdx.x; i < 2 * NC; i += kTileThreads) ln.partials[(int64_t)blockIdx.x * 2 * NC + i] = 0.f;
    return;
  }

  TSTAMP(0);
  for (int i = threadIdx.x; i < T; i += kTileThreads) s_owner[i] = 0u;
  uint32_t claim = 0;
  for (int i = threadIdx.x; i < T * (NC / 4); i += kTileThreads) {
    const int r = i / (NC / 4), c4 = i % (NC / 4);
    *(f32x4*)(tile + r * LDT + c4 * 4) = bias ? *(const f32x4*)(bias + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
  // fragment (cb, ks): lane (lrow, kg) holds W[k][channel 16 cb + lrow][32 ks + 8 kg .. +7], straight from L2.
  // wn is in FRAGMENT-major order (ococc_weight_prepare_bf16 mode + 4): [k][cb][ks][lane][8], so one load
  // instruction reads 1 KB of consecutive bytes (row-major rows made it 64 pieces of 16 bytes from 16 rows:
  // the sparse pass then took 31 of the kernel's 45 us)
  auto load_w = [&](bf16x8 (&w)[NB][KSTEPS], int k) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        w[cb][ks] = *(const bf16x8*)(wn + ((((int64_t)k * NB + cb) * KSTEPS + ks) * 64 + lane) * 8);
  };
  // the rows of one 16-row block: lane (lrow, kg) names the input row of slot lrow; a negative row gives an
  // out-of-range offset, for which the buffer unit returns zeros without touching memory
  auto gather = [&](bf16x8 (&x)[KSTEPS], int32_t in) {
    const uint32_t off = in >= 0 ? (uint32_t)in * (KD * 2) + kg * 16 : 0xffffff00u;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      x[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, off + ks * 64, 0, 0));
  };
  auto mma = [&](f32x4 (&acc)[NB], const bf16x8 (&w)[NB][KSTEPS], const bf16x8 (&x)[KSTEPS]) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][ks], x[ks], acc[cb], 0, 0, 0);
    }
  };
  // accumulator layout: lane (lrow, kg) holds channels 16 cb + 4 kg .. +3 of the row in slot lrow
  auto add_block = [&](const f32x4 (&acc)[NB], int row_local) {
    float* dst = tile + row_local * LDT + 4 * kg;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      f32x4 v = *(f32x4*)(dst + cb * 16);
      v += acc[cb];
      *(f32x4*)(dst + cb * 16) = v;
    }
  };
  auto table_col = [&](int32_t (&e)[U], int k) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + u * 64 + lane;
      e[u] = (k >= 0 && row < n_out) ? table[(int64_t)k * n_out + row] : -1;
    }
  };
  // ranks of the rows that have a neighbour (ballot prefix, row order); those ranked first .. first+span-1 leave
  // (input row, tile row) in slot j's list.  Returns how many rows have a neighbour.
  auto compact16 = [&](const int32_t (&e)[U], int j, int first, int span) -> int {
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long m = __ballot(e[u] >= 0);
      if (e[u] >= 0) {
        const int p = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) - first;
        if ((unsigned)p < (unsigned)span) {
          sl_in[wave][j][p] = e[u];
          sl_row[wave][j][p] = (uint16_t)(u * 64 + lane);
        }
      }
      cnt += (int)__popcll(m);
    }
    return cnt;
  };
  __syncthreads();
  TSTAMP(1);

  bf16x8 w[2][NB][KSTEPS];
  // ---- the dense offset: row r of the tile is slot r; waves own disjoint 16-row blocks, loads batched ----
  if (dense_k >= 0) {
    load_w(w[0], dense_k);
    int32_t in[DB];
    bf16x8 x[DB][KSTEPS];
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      const int64_t row = row0 + 16 * (wave + NW * d) + lrow;
      in[d] = row < n_out ? table[(int64_t)dense_k * n_out + row] : -1;
    }
#pragma unroll
    for (int d = 0; d < DB; ++d) gather(x[d], in[d]);
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      f32x4 acc[NB];
      mma(acc, w[0], x[d]);
      if (in[d] >= 0) add_block(acc, 16 * (wave + NW * d) + lrow);
    }
  }
  __syncthreads();
  TSTAMP(2);

  // ---- the other offsets: in pass p, round j, wave v takes the (p*MAXO + j)*NW + v -th of them ----
  const int nk = kvol - (dense_k >= 0 ? 1 : 0);
  for (int pass = 0; pass * MAXO * NW < nk; ++pass) {
    int kk[MAXO], cnt[MAXO];
    {
      int32_t e[MAXO][U];
#pragma unroll
      for (int j = 0; j < MAXO; ++j) {
        const int idx = (pass * MAXO + j) * NW + wave;
        kk[j] = idx < nk ? ((dense_k >= 0 && idx >= dense_k) ? idx + 1 : idx) : -1;
        table_col(e[j], kk[j]);
      }
#pragma unroll
      for (int j = 0; j < MAXO; ++j) cnt[j] = compact16(e[j], j, 0, 16 * FB);
    }
    if (pass == 0) TSTAMP(3);
    // (the lists are written and read by the same wave: LDS keeps a wave's operations in order)
    bf16x8 x[MAXO][FB][KSTEPS];
    int32_t in[MAXO][FB];
    int rowl[MAXO][FB];
#pragma unroll
    for (int j = 0; j < MAXO; ++j)
#pragma unroll
      for (int f = 0; f < FB; ++f) {
        const bool ok = 16 * f + lrow < cnt[j];
        in[j][f] = ok ? sl_in[wave][j][16 * f + lrow] : -1;
        rowl[j][f] = ok ? (int)sl_row[wave][j][16 * f + lrow] : 0;
        gather(x[j][f], in[j][f]);  // (no rows: every lane out of range, no memory access)
      }
    // weight fragments: two sets (the next offset's arrive during this one's work) where the registers allow,
    // else one set, re-requested right behind the MFMAs that read it
    constexpr int WB = (NB * KSTEPS * 4 * 2 + MAXO * FB * KSTEPS * 4 <= ((T <= 256 && NC <= 64) ? 80 : 200)) ? 2 : 1;
    if (cnt[0] > 0) load_w(w[0], kk[0]);
    if (pass == 0) TSTAMP(4);
#pragma unroll
    for (int j = 0; j < MAXO; ++j) {
      if (WB == 2 && j + 1 < MAXO && cnt[(j + 1) % MAXO] > 0) load_w(w[(j + 1) & 1], kk[(j + 1) % MAXO]);
      f32x4 acc[FB][NB];
#pragma unroll
      for (int f = 0; f < FB; ++f)
        if (16 * f < cnt[j]) mma(acc[f], w[WB == 2 ? (j & 1) : 0], x[j][f]);
      if (WB == 1 && j + 1 < MAXO && cnt[(j + 1) % MAXO] > 0 && cnt[j] <= 16 * FB) load_w(w[0], kk[(j + 1) % MAXO]);
      if (lane == 0) s_cnt[wave] = cnt[j];
      if (pass == 0 && j == 0) TSTAMP(5);
      __syncthreads();
      if (pass == 0 && j == 0) TSTAMP(6);
      int most = s_cnt[0];
#pragma unroll
      for (int t = 1; t < NW; ++t) most = s_cnt[t] > most ? s_cnt[t] : most;
      // Additions into the shared tile.  The sum order per output element stays "ascending offset" (= ascending wave
      // inside a round), but the eight waves no longer take turns: two waves of a round rarely hit the same row (a row
      // has 0.76 sparse contributions in all), so every wave claims its rows first -- LDS atomic max of a tag that grows
      // with the claim round and, inside one, is largest for the LOWEST wave -- and the waves whose claims stood add
      // together.  A wave that lost a row to an earlier offset claims again in the next claim round, after the winner's
      // addition: the order of the additions to any one row is exactly the serial one, the result bit for bit the same.
      {
        bool pend[FB];
#pragma unroll
        for (int f = 0; f < FB; ++f) pend[f] = in[j][f] >= 0;
        for (;;) {
          ++claim;
          const uint32_t tag = ((uint32_t)claim << 8) | (uint32_t)(255 - wave);
#pragma unroll
          for (int f = 0; f < FB; ++f)
            if (pend[f] && kg == 0) atomicMax(&s_owner[rowl[j][f]], tag);
          if (threadIdx.x == 0) s_more[claim & 1] = 0;
          __syncthreads();
          bool lost = false;
#pragma unroll
          for (int f = 0; f < FB; ++f)
            if (pend[f]) {
              if (s_owner[rowl[j][f]] == tag) {
                add_block(acc[f], rowl[j][f]);
                pend[f] = false;
              } else {
                lost = true;
              }
            }
          if (lost) s_more[claim & 1] = 1;  // benign race: every writer stores 1
          __syncthreads();
          if (!s_more[claim & 1]) break;  // (the flag of this claim round is cleared again two rounds later)
        }
      }
      if (pass == 0 && j == 0) TSTAMP(7);
      // offsets with more than 16 FB rows (dense neighbourhoods): the remaining blocks one by one
      for (int b = FB; b * 16 < most; ++b) {
        const bool have = b * 16 < cnt[j];
        int32_t in2 = -1;
        int rowl2 = 0;
        if (have) {
          int32_t e2[U];
          table_col(e2, kk[j]);
          compact16(e2, j, b * 16, 16);
          const bool ok = b * 16 + lrow < cnt[j];
          in2 = ok ? sl_in[wave][j][lrow] : -1;
          rowl2 = ok ? (int)sl_row[wave][j][lrow] : 0;
          bf16x8 x2[KSTEPS];
          gather(x2, in2);
          mma(acc[0], w[WB == 2 ? (j & 1) : 0], x2);
        }
#pragma unroll
        for (int t = 0; t < NW; ++t) {
          if (wave == t && in2 >= 0) add_block(acc[0], rowl2);
          __syncthreads();
        }
      }
      // (one fragment set and a long offset: its fragments were still needed above)
      if (WB == 1 && j + 1 < MAXO && cnt[(j + 1) % MAXO] > 0 && cnt[j] > 16 * FB) load_w(w[0], kk[(j + 1) % MAXO]);
    }
    __syncthreads();
  }

  TSTAMP(8);
  if constexpr (LNB) {
    // ---- epilogue of a dgrad whose output is the gradient of a conv -> LayerNorm -> act block's OUTPUT: the block's
    // LayerNorm (+ GELU) backward happens here, on the finished f32 row, and the gradient of the block's conv output
    // leaves instead (the separate LN-backward launch read this row back and wrote that one).  Arithmetic and its
    // order are those of ln_act_bwd_vec_kernel (the incoming gradient rounded to bf16 first, as the unfused pair
    // sees it): the rows are bit-identical; the per-workgroup sums for d gamma / d beta are grouped differently.
    constexpr int LPRC = NC / 8;
    const int c8 = threadIdx.x % LPRC;  // (kTileThreads is a multiple of LPRC: a thread keeps its channels)
    ln_f32x2 g[4], b[4], dg[4], db[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      g[p] = ln_f32x2{ln.gamma[c8 * 8 + 2 * p], ln.gamma[c8 * 8 + 2 * p + 1]};
      b[p] = ln_f32x2{ln.beta[c8 * 8 + 2 * p], ln.beta[c8 * 8 + 2 * p + 1]};
      dg[p] = db[p] = ln_f32x2{0.f, 0.f};
    }
    // the block's conv output and statistics for ALL of this thread's rows are requested at once
    constexpr int IT = T * LPRC / kTileThreads;
    static_assert(T * LPRC % kTileThreads == 0, "rows per thread");
    u32x4 xin_[IT];
    float mean_[IT], rstd_[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int64_t row = row0 + (it * kTileThreads + threadIdx.x) / LPRC;
      const int64_t rc = row < n_out ? row : n_out - 1;
      xin_[it] = *(const u32x4*)(ln.y + rc * NC + c8 * 8);
      mean_[it] = ln.mean_rstd[rc * 2];
      rstd_[it] = ln.mean_rstd[rc * 2 + 1];
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int r = (it * kTileThreads + threadIdx.x) / LPRC;
      const int64_t row = row0 + r;
      if (row >= n_out) continue;  // (whole rows leave together)
      const f32x4 v0 = *(const f32x4*)(tile + r * LDT + c8 * 8), v1 = *(const f32x4*)(tile + r * LDT + c8 * 8 + 4);
      ln_f32x2 xv[4], dv[4], dzg[4];
      ln_unpack8(xin_[it], xv);
      dv[0] = ln_f32x2{tile_round_bf16(v0.x), tile_round_bf16(v0.y)};
      dv[1] = ln_f32x2{tile_round_bf16(v0.z), tile_round_bf16(v0.w)};
      dv[2] = ln_f32x2{tile_round_bf16(v1.x), tile_round_bf16(v1.y)};
      dv[3] = ln_f32x2{tile_round_bf16(v1.z), tile_round_bf16(v1.w)};
      float s1, s2;
      if (ln.act == 1) ln_bwd_piece8<true>(xv, dv, mean_[it], rstd_[it], g, b, dg, db, dzg, s1, s2);
      else ln_bwd_piece8<false>(xv, dv, mean_[it], rstd_[it], g, b, dg, db, dzg, s1, s2);
#pragma unroll
      for (int d = LPRC >> 1; d >= 1; d >>= 1) {
        s1 += __shfl_xor(s1, d, 64);
        s2 += __shfl_xor(s2, d, 64);
      }
      *(u32x4*)((uint16_t*)out_ + row * NC + c8 * 8) = ln_bwd_finish8(xv, dzg, rstd_[it], s1 * (1.f / NC), s2 * (1.f / NC));
    }
    __syncthreads();  // the tile is read; its memory now carries the column sums of the thread groups
    constexpr int GROUPS = kTileThreads / LPRC;
    static_assert(GROUPS * 2 * NC <= T * LDT, "column sums fit the tile");
    float* mine = tile + (threadIdx.x / LPRC) * 2 * NC;
    *(f32x4*)(mine + c8 * 8) = f32x4{dg[0].x, dg[0].y, dg[1].x, dg[1].y};
    *(f32x4*)(mine + c8 * 8 + 4) = f32x4{dg[2].x, dg[2].y, dg[3].x, dg[3].y};
    *(f32x4*)(mine + NC + c8 * 8) = f32x4{db[0].x, db[0].y, db[1].x, db[1].y};
    *(f32x4*)(mine + NC + c8 * 8 + 4) = f32x4{db[2].x, db[2].y, db[3].x, db[3].y};
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NC; i += kTileThreads) {
      float sum = 0.f;
#pragma unroll 4
      for (int gq = 0; gq < GROUPS; ++gq) sum += tile[gq * 2 * NC + i];
      ln.partials[(int64_t)blockIdx.x * 2 * NC + i] = sum;
    }
    TSTAMP(9);
    return;
  }
  // ---- epilogue: tile -> global, 8 channels (16 bytes of bf16) per thread; a row is NC / 8 consecutive lanes ----
  for (int i = threadIdx.x; i < T * (NC / 8); i += kTileThreads) {
    const int r = i / (NC / 8), c8 = i % (NC / 8);
    const int64_t row = row0 + r;
    if (row >= n_out) continue;  // (whole rows leave together: the lane exchanges below stay inside a row)
    f32x4 v0 = *(const f32x4*)(tile + r * LDT + c8 * 8), v1 = *(const f32x4*)(tile + r * LDT + c8 * 8 + 4);
    if (OUT_BF16) {
      u32x4 q;
      q.x = (uint32_t)ococc_f32_to_bf16(v0.x) | ((uint32_t)ococc_f32_to_bf16(v0.y) << 16);
      q.y = (uint32_t)ococc_f32_to_bf16(v0.z) | ((uint32_t)ococc_f32_to_bf16(v0.w) << 16);
      q.z = (uint32_t)ococc_f32_to_bf16(v1.x) | ((uint32_t)ococc_f32_to_bf16(v1.y) << 16);
      q.w = (uint32_t)ococc_f32_to_bf16(v1.z) | ((uint32_t)ococc_f32_to_bf16(v1.w) << 16);
      *(u32x4*)((uint16_t*)out_ + row * NC + c8 * 8) = q;
    } else {
      *(f32x4*)((float*)out_ + row * NC + c8 * 8) = v0;
      *(f32x4*)((float*)out_ + row * NC + c8 * 8 + 4) = v1;
    }
    if constexpr (LN) {
      float z[8];
      float rs = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        z[j] = tile_round_bf16(v0[j]);
        z[4 + j] = tile_round_bf16(v1[j]);
        rs += z[j] + z[4 + j];
      }
#pragma unroll
      for (int d = 1; d < NC / 8; d <<= 1) rs += __shfl_xor(rs, d, 64);
      const float mean = rs * (1.f / NC);
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dlt = z[j] - mean;
        sq += dlt * dlt;
      }
#pragma unroll
      for (int d = 1; d < NC / 8; d <<= 1) sq += __shfl_xor(sq, d, 64);
      const float rstd = rsqrtf(sq * (1.f / NC) + ln.eps);
      const f32x4 g0 = *(const f32x4*)(ln.gamma + c8 * 8), g1 = *(const f32x4*)(ln.gamma + c8 * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(ln.beta + c8 * 8), b1 = *(const f32x4*)(ln.beta + c8 * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t0 = (z[j] - mean) * rstd * g0[j] + b0[j];
        const float t1 = (z[4 + j] - mean) * rstd * g1[j] + b1[j];
        z[j] = ln.act == 1 ? tile_gelu(t0) : t0;
        z[4 + j] = ln.act == 1 ? tile_gelu(t1) : t1;
      }
      u32x4 q;
      q.x = (uint32_t)ococc_f32_to_bf16(z[0]) | ((uint32_t)ococc_f32_to_bf16(z[1]) << 16);
      q.y = (uint32_t)ococc_f32_to_bf16(z[2]) | ((uint32_t)ococc_f32_to_bf16(z[3]) << 16);
      q.z = (uint32_t)ococc_f32_to_bf16(z[4]) | ((uint32_t)ococc_f32_to_bf16(z[5]) << 16);
      q.w = (uint32_t)ococc_f32_to_bf16(z[6]) | ((uint32_t)ococc_f32_to_bf16(z[7]) << 16);
      *(u32x4*)(ln.y + row * NC + c8 * 8) = q;
      if (c8 == 0) {
        ln.mean_rstd[row * 2] = mean;
        ln.mean_rstd[row * 2 + 1] = rstd;
      }
    }
  }
  TSTAMP(9);
}


template <int KD, int NC, int T>
int launch_tile_t(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, int dense_k, const int32_t* table,
                  int64_t n_out, const float* bias, void* out, int out_dtype, hipStream_t stream,
                  const TileLn* ln = nullptr) {
  constexpr size_t lds = (size_t)T * (NC + 4) * 4;
  const dim3 grid((unsigned)ococc_align_up(ococc_cdiv(n_out, T), 8));
  if (ln && ln->partials) {
    if constexpr (NC <= 64) {
      auto fn = subm_tile_conv_kernel<KD, NC, T, true, false, true>;
      OCOCC_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(fn, grid, dim3(kTileThreads), lds, stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, dense_k,
                         table, n_out, bias, out, *ln);
    } else {
      return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "fused LayerNorm backward epilogue: up to 64 output columns");
    }
  } else if (ln) {
    if constexpr (NC <= 64 && KD <= 64) {
      auto fn = subm_tile_conv_kernel<KD, NC, T, true, true>;
      OCOCC_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(fn, grid, dim3(kTileThreads), lds, stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, dense_k,
                         table, n_out, bias, out, *ln);
    } else {
      return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "fused LayerNorm epilogue: shapes up to 64 x 64 channels");
    }
  } else if (out_dtype == OCOCC_BF16) {
    auto fn = subm_tile_conv_kernel<KD, NC, T, true>;
    OCOCC_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, grid, dim3(kTileThreads), lds, stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, dense_k,
                       table, n_out, bias, out, TileLn{});
  } else {
    auto fn = subm_tile_conv_kernel<KD, NC, T, false>;
    OCOCC_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, grid, dim3(kTileThreads), lds, stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, dense_k,
                       table, n_out, bias, out, TileLn{});
  }
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

template <int KD, int NC>
int launch_tile(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, int dense_k, const int32_t* table,
                int64_t n_out, const float* bias, void* out, int out_dtype, hipStream_t stream,
                const TileLn* ln = nullptr) {
  if constexpr (NC >= 128)
    return launch_tile_t<KD, NC, 256>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
  else if constexpr (KD <= 64) {
    // 256-row tiles, one block per offset gathered ahead: <= 128 registers, so TWO workgroups share a CU and one's
    // load latency hides behind the other's phases (64 -> 32: 19.2 -> 16.5 us, 32 -> 64: 21.1 -> 17.6 us)
    return launch_tile_t<KD, NC, 256>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
  } else {
    // (128 input channels at 256 rows: fits 128 registers only with two offsets per wave and pass and one set of
    // weight fragments -- 30.6 us against 29.5 us for the 512-row tile at one workgroup per CU)
    if (ln && ln->partials)   // LayerNorm-backward epilogue (10 us of arithmetic per 512-row tile): see kTileLnbRows
      return launch_tile_t<KD, NC, kTileLnbRows>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
    return launch_tile_t<KD, NC, 512>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
  }
}

template <int KD>
int dispatch_tile_nc(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, int ncols, int dense_k,
                     const int32_t* table, int64_t n_out, const float* bias, void* out, int out_dtype,
                     hipStream_t stream, const TileLn* ln = nullptr) {
  switch (ncols) {
    case 32: return launch_tile<KD, 32>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
    case 64: return launch_tile<KD, 64>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
    case 128:
      // (128 x 128: two sets of weight fragments alone are 256 registers)
      if constexpr (KD <= 64)
        return launch_tile<KD, 128>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
      else
        return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "128 x 128 channels: use ococc_sparse_conv_gather_gemm_bf16");
    default: return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "ncols must be 32/64/128");
  }
}

}  // namespace

namespace {
int tile_entry(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn, int32_t kvol, int32_t ncols,
               const int32_t* table, int32_t dense_k, int64_t n_out, const float* bias, void* out, int32_t out_dtype,
               hipStream_t stream, const TileLn* ln) {
  OCOCC_REQUIRE(n_in >= 0 && n_out >= 0, "negative row count");
  OCOCC_REQUIRE(kvol >= 1, "kernel volume must be >= 1");
  OCOCC_REQUIRE(dense_k >= -1 && dense_k < kvol, "dense_k must be -1 or an offset index");
  OCOCC_REQUIRE(out_dtype == OCOCC_BF16 || out_dtype == OCOCC_F32, "out_dtype must be f32/bf16");
  if (n_out == 0) return OCOCC_OK;
  OCOCC_REQUIRE(wn && table && out, "null pointer");
  OCOCC_REQUIRE(feat || n_in == 0, "null feat");
  OCOCC_REQUIRE(n_in * kd * 2 < 0xffffff00ll, "feat too large for the 32-bit buffer offsets of the gathers");
  switch (kd) {
    case 32: return dispatch_tile_nc<32>(feat, n_in, wn, kvol, ncols, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
    case 64: return dispatch_tile_nc<64>(feat, n_in, wn, kvol, ncols, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
    case 128: return dispatch_tile_nc<128>(feat, n_in, wn, kvol, ncols, dense_k, table, n_out, bias, out, out_dtype, stream, ln);
    default: return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "kd must be 32/64/128");
  }
}
}  // namespace

#ifdef OCOCC_TILE_STAMPS
extern "C" int ococc_tile_set_stamps(long long* dev_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(t_stamps), &dev_buffer, sizeof(dev_buffer)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int ococc_sparse_conv_tile_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn,
                                           int32_t kvol, int32_t ncols, const int32_t* table, int32_t dense_k,
                                           int64_t n_out, const float* bias, void* out, int32_t out_dtype,
                                           ococc_stream_t stream_) {
  return tile_entry(feat, n_in, kd, wn, kvol, ncols, table, dense_k, n_out, bias, out, out_dtype,
                    (hipStream_t)stream_, nullptr);
}

extern "C" int ococc_sparse_conv_tile_ln_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn,
                                              int32_t kvol, int32_t ncols, const int32_t* table, int32_t dense_k,
                                              int64_t n_out, const float* gamma, const float* beta, float eps,
                                              int32_t act, uint16_t* conv_out, uint16_t* y, float* mean_rstd,
                                              ococc_stream_t stream_) {
  OCOCC_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (gelu)");
  OCOCC_REQUIRE(n_out == 0 || (gamma && beta && y && mean_rstd), "null pointer");
  const TileLn ln{gamma, beta, eps, act, y, mean_rstd, nullptr};
  return tile_entry(feat, n_in, kd, wn, kvol, ncols, table, dense_k, n_out, nullptr, conv_out, OCOCC_BF16,
                    (hipStream_t)stream_, &ln);
}

namespace {
// rows of the tile the launcher picks for a shape (launch_tile above)
inline int tile_rows_for(int kd, int ncols) { return (ncols >= 128 || kd <= 64) ? 256 : kTileLnbRows; }   // (only the LNB entry asks)
}  // namespace

extern "C" int64_t ococc_sparse_conv_tile_lnbwd_partial_rows(int64_t n_out, int32_t kd, int32_t ncols) {
  if (n_out < 0 || (kd != 32 && kd != 64 && kd != 128) || (ncols != 32 && ncols != 64)) return -1;
  return ococc_align_up(ococc_cdiv(n_out > 0 ? n_out : 1, tile_rows_for(kd, ncols)), 8);
}

extern "C" int ococc_sparse_conv_tile_lnbwd_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn,
                                                 int32_t kvol, int32_t ncols, const int32_t* table, int32_t dense_k,
                                                 int64_t n_out, const uint16_t* block_conv_out,
                                                 const float* mean_rstd, const float* gamma, const float* beta,
                                                 int32_t act, uint16_t* d_conv_out, float* partials,
                                                 int64_t partial_rows, ococc_stream_t stream_) {
  OCOCC_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (gelu)");
  OCOCC_REQUIRE(ncols == 32 || ncols == 64, "fused LayerNorm backward: 32 or 64 output columns");
  OCOCC_REQUIRE(n_out == 0 || (gamma && beta && block_conv_out && mean_rstd && d_conv_out && partials), "null pointer");
  OCOCC_REQUIRE(partial_rows >= ococc_sparse_conv_tile_lnbwd_partial_rows(n_out, kd, ncols), "partials too small");
  const TileLn ln{gamma, beta, 0.f, act, const_cast<uint16_t*>(block_conv_out), const_cast<float*>(mean_rstd), partials};
  return tile_entry(feat, n_in, kd, wn, kvol, ncols, table, dense_k, n_out, nullptr, d_conv_out, OCOCC_BF16,
                    (hipStream_t)stream_, &ln);
}
