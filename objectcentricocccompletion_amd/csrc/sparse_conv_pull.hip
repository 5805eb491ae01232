// B4  sub-manifold convolution for SPARSE active sets and WIDE outputs: compact, multiply, pull.
// Same contract as ococc_sparse_conv_tile_bf16 (out[o] = sum_k feat[table[k][o]] @ W[k] over a sub-manifold table;
// replaces indiceConv, spconv_ops.h:260-456), written for the shape the other two kernels are bad at: 64 -> 128
// channels on configs[1]'s grids (0.76 neighbours per voxel besides itself).
//
//   * gather_gemm_stream_kernel (sparse_conv.hip) keeps the f32 accumulators in the registers of the wave that owns
//     64 rows -- which is the only place the 64.5 MB of accumulators of 126 k x 128 outputs fit in ONE round of
//     workgroups (the chip has 40 MB of LDS and 128 MB of registers) -- but multiplies whole 16-row blocks in place:
//     a block has a neighbour at a given offset 37 % of the time and then carries 1.3 useful rows, 6.3 x more matrix
//     instructions than rows need, and v_mfma_f32_16x16x32_bf16 issues at half the rate of the 32 x 32 shape on this
//     part (tools/probe/mfma_rate.hip): 19 of its 38.7 us are matrix issue.
//   * subm_tile_conv_kernel (sparse_conv_tile.hip) compacts the rows of an offset over the tile before it multiplies
//     (7.5 rows per offset and 256-row tile -> ONE block), 4 x fewer matrix instructions, but accumulates in LDS: at
//     128 columns a 256-row tile takes 128 KB, one workgroup per CU, two rounds of workgroups, 62 us.
//
// Here both: accumulators in registers (a wave owns 64 rows x 64 columns: 64 registers), products compacted.  The rows
// of a 256-row tile that have a neighbour at offset k are ranked once into LDS lists (position <-> row).  Per round
// four offsets are multiplied -- a PAIR of waves (one per column half) takes an offset: gathers its <= 16 listed
// rows, multiplies them by the offset's weight fragments straight from L2 (fragment-major order, as the tile kernel),
// leaves the 16 x 128 f32 products in an LDS slot -- then every wave PULLS: for each of its row blocks it looks up
// whether a row has an entry in the offset's list and adds that product row to its accumulators.  Additions to a row
// happen in ascending offset order: deterministic.  Eight waves, 120 registers, 74 KB of LDS: two workgroups per CU,
// four waves per SIMD to hide the latencies that a round exposes.
#include "common.hpp"

namespace {

constexpr int kPullThreads = 512;
constexpr int kPullWaves = 8;
constexpr int kPullMaxVol = 27;     // kernel offsets (3 x 3 x 3)
constexpr int kPullQ = 4;           // offsets per round

template <int KD, int NC, bool OUT_BF16, int T>
__global__ void __launch_bounds__(kPullThreads, T == 256 ? 4 : 2)
subm_pull_conv_kernel(const uint16_t* __restrict__ feat, uint32_t feat_bytes, const uint16_t* __restrict__ wn, int kvol,
                      int dense_k, const int32_t* __restrict__ table, int64_t n_out, const float* __restrict__ bias,
                      void* __restrict__ out_, int dbg) {
  constexpr int KSTEPS = KD / 32, NB = NC / 16, HB = NB / 2, RB = T / 64, WR = T / 4;   // WR: rows per wave
  constexpr int OPW = (kPullMaxVol + kPullWaves - 1) / kPullWaves;   // offsets a wave ranks
  static_assert(KD % 32 == 0 && NC % 32 == 0, "pull kernel shape");
  // LDS, carved by hand so that the epilogue can reuse all of it as the bf16 staging tile
  constexpr int kInBytes = kPullMaxVol * T * 4;                 // s_in  [27][T] int32: offset k's input rows, compacted
  constexpr int kPosBytes = kPullMaxVol * T * 2;                // s_pos [27][T] uint16: tile row -> position in the list
  constexpr int kSlotBytes = kPullQ * NB * 16 * 4 * 16;         // slot  [4][NB][16][4] f32x4: products of a round
  constexpr int kMaskBytes = T * 4;                             // s_mask[T] uint32: bit k = row has a neighbour at offset k
  constexpr int kStageLd = NC + 8;                              // (bf16 elements; 16 bytes of pad per row)
  constexpr int kStageBytes = OUT_BF16 ? T * kStageLd * 2 : 0;
  constexpr int kCarved = kInBytes + kPosBytes + kSlotBytes + kMaskBytes + 128;
  constexpr int kLds = kCarved > kStageBytes ? kCarved : kStageBytes;
  static_assert(kLds <= (T == 256 ? 80 : 160) * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem[kLds];
  int32_t(*s_in)[T] = (int32_t(*)[T])smem;
  uint16_t(*s_pos)[T] = (uint16_t(*)[T])(smem + kInBytes);
  f32x4(*slot)[NB][16][4] = (f32x4(*)[NB][16][4])(smem + kInBytes + kPosBytes);
  uint32_t* s_mask = (uint32_t*)(smem + kInBytes + kPosBytes + kSlotBytes);
  int* s_cnt = (int*)(smem + kInBytes + kPosBytes + kSlotBytes + kMaskBytes);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lrow = lane & 15, kg = lane >> 4;
  const int wrow = wave & 3;     // the wave's 64 rows of the tile (and the offset of a round it multiplies)
  const int half = wave >> 2;    // its column half: blocks HB half .. HB half + HB - 1
  // XCD-aware: workgroups are dealt round-robin to the 8 XCDs; give each a contiguous eighth of the tiles
  const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t row0 = (int64_t)wg * T;
  if (row0 >= n_out) return;

  const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)feat_bytes, 0x00020000);
  // a negative row gives an out-of-range offset, for which the buffer unit returns zeros without touching memory
  auto gather = [&](bf16x8 (&x)[KSTEPS], int32_t in) {
    const uint32_t off = in >= 0 ? (uint32_t)in * (KD * 2) + kg * 16 : 0xffffff00u;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      x[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, off + ks * 64, 0, 0));
  };
  // fragment (cb, ks) of offset k: lane (lrow, kg) holds W[k][channel 16 cb + lrow][32 ks + 8 kg .. +7]; wn is in
  // fragment-major order (ococc_weight_prepare_bf16 mode + 4): one load instruction = 1 KB of consecutive bytes
  auto load_w = [&](bf16x8 (&w)[HB][KSTEPS], int k) {
#pragma unroll
    for (int c = 0; c < HB; ++c)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        w[c][ks] = *(const bf16x8*)(wn + ((((int64_t)k * NB + HB * half + c) * KSTEPS + ks) * 64 + lane) * 8);
  };

  // ---- everything the first phases wait for is requested up front: the table columns this wave ranks (offsets
  // wave, wave + 8, ...: OPW x 4 loads), the dense offset's weights and this wave's own rows
  int32_t tv[OPW][T / 64];
#pragma unroll
  for (int o = 0; o < OPW; ++o) {
    const int k = wave + kPullWaves * o;
#pragma unroll
    for (int u = 0; u < T / 64; ++u) {
      const int64_t row = row0 + 64 * u + lane;
      tv[o][u] = (k < kvol && k != dense_k && row < n_out) ? table[(int64_t)k * n_out + row] : -1;
    }
  }
  bf16x8 w[HB][KSTEPS];
  int32_t own[WR / 64];   // this wave's rows at the dense offset (lane -> row WR wrow + 64 i + lane)
  if (dense_k >= 0) {
    load_w(w, dense_k);
#pragma unroll
    for (int i = 0; i < WR / 64; ++i) {
      const int64_t row = row0 + WR * wrow + 64 * i + lane;
      own[i] = row < n_out ? table[(int64_t)dense_k * n_out + row] : -1;
    }
  }
  if (threadIdx.x < T) s_mask[threadIdx.x] = 0u;
  __syncthreads();

  // ---- lists
#pragma unroll
  for (int o = 0; o < OPW; ++o) {
    const int k = wave + kPullWaves * o;
    if (k >= kvol || k == dense_k) continue;
    int base = 0;
#pragma unroll
    for (int u = 0; u < T / 64; ++u) {
      const bool has = tv[o][u] >= 0;
      const uint64_t m = __ballot(has);
      const int rank = base + __popcll(m & ((1ull << lane) - 1ull));
      if (has) {
        s_pos[k][64 * u + lane] = (uint16_t)rank;
        s_in[k][rank] = tv[o][u];
        atomicOr(&s_mask[64 * u + lane], 1u << k);
      }
      base += __popcll(m);
    }
    if (lane == 0) s_cnt[k] = base;
  }

  // ---- accumulators: lane (lrow, kg) holds channels 16 (HB half + c) + 4 kg .. +3 of tile row 64 wrow + 16 rb + lrow
  f32x4 acc[RB][HB];
#pragma unroll
  for (int c = 0; c < HB; ++c) {
    const f32x4 b = bias ? *(const f32x4*)(bias + 16 * (HB * half + c) + 4 * kg) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb][c] = b;
  }
  // ---- the dense offset: every row block in place
  if (dense_k >= 0 && !(dbg & 2)) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      bf16x8 x[KSTEPS];
      gather(x, __shfl(own[rb / 4], 16 * (rb % 4) + lrow, 64));
#pragma unroll
      for (int c = 0; c < HB; ++c)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
          acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][ks], x[ks], acc[rb][c], 0, 0, 0);
    }
  }
  __syncthreads();   // the lists are complete
  uint32_t rmask[RB];   // which offsets add to this lane's rows
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) rmask[rb] = s_mask[WR * wrow + 16 * rb + lrow];

  // ---- rounds of kPullQ sparse offsets (ascending; offset index so skips the dense one).  The wave multiplies
  // offset number wrow of a round; its weights and its first block's rows are requested a round ahead.
  const int nsparse = (dbg & 1) ? 0 : (dense_k >= 0 ? kvol - 1 : kvol);
  auto offset_of = [&](int so) { return (dense_k >= 0 && so >= dense_k) ? so + 1 : so; };
  bf16x8 x[KSTEPS];
  // Weights and first rows of the offset this wave takes in the round starting at r0.  ALWAYS issued (past the end: the
  // last offset again, never used): a conditional definition of 40 loop-carried registers is more than the register
  // allocator handles -- it spilled the accumulators.
  auto prefetch = [&](int r0, int lrow) {
    int so = r0 + wrow;
    so = so < nsparse ? so : nsparse - 1;
    const int k = offset_of(so);
    const int c = __builtin_amdgcn_readfirstlane(s_cnt[k]);
    load_w(w, k);
    gather(x, lrow < c ? s_in[k][lrow] : -1);
  };
  auto produce = [&](int lrow, int kg) {   // the products of the 16 rows in x -> the wave's half of its offset's slot
#pragma unroll
    for (int c = 0; c < HB; ++c) {
      f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][ks], x[ks], p, 0, 0, 0);
      slot[wrow][HB * half + c][lrow][kg] = p;
    }
  };
  auto pull = [&](int r0, int b0, int lrow, int kg) {   // every wave: the round's products of listed rows b0 .. b0 + 15
#pragma unroll 1
    for (int q = 0; q < kPullQ; ++q) {   // (rolled: the round's code stays small)
      const int so = r0 + q;
      if (so >= nsparse) break;
      const int k = offset_of(so);
      if (b0 >= __builtin_amdgcn_readfirstlane(s_cnt[k])) continue;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const bool has = (rmask[rb] >> k) & 1u;
        if (__ballot(has)) {   // (wave-uniform: a row block without an entry at this offset costs one test)
          const uint32_t j = s_pos[k][WR * wrow + 16 * rb + lrow];   // (rows without an entry read a stale position)
          const bool hit = has && (int)(j & ~15u) == b0;
#pragma unroll
          for (int c = 0; c < HB; ++c) {
            const f32x4 p = slot[q][HB * half + c][j & 15u][kg];
            acc[rb][c] = acc[rb][c] + (hit ? p : f32x4{0.f, 0.f, 0.f, 0.f});
          }
        }
      }
    }
  };
  if (nsparse > 0) prefetch(0, lrow);
#pragma unroll 1
  for (int r0 = 0; r0 < nsparse; r0 += kPullQ) {
    // (the LDS addresses of a round are loop invariant per lane; left alone the compiler computes all of them ahead of the
    // loop and spills them around the accumulators: re-derive the lane's coordinates from an opaque copy every round)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lrow = tid_ & 15, kg = (tid_ >> 4) & 3;
    int maxc = 0, myk = 0, myc = 0;   // maxc: the longest list of the round (workgroup-uniform); my*: this wave's offset
#pragma unroll
    for (int q = 0; q < kPullQ; ++q) {
      const int so = r0 + q;
      const int k = offset_of(so < nsparse ? so : nsparse - 1);
      const int c = so < nsparse ? __builtin_amdgcn_readfirstlane(s_cnt[k]) : 0;
      maxc = c > maxc ? c : maxc;
      if (q == wrow) {
        myk = k;
        myc = c;
      }
    }
    if (myc > 0) produce(lrow, kg);
    if (maxc <= 16) prefetch(r0 + kPullQ, lrow);   // (the usual case) next round's operands fly during the pull
    __syncthreads();
    if (maxc > 0) pull(r0, 0, lrow, kg);
    __syncthreads();
    if (maxc > 16) {   // dense neighbourhoods: further blocks of 16 listed rows, one at a time
      for (int b0 = 16; b0 < maxc; b0 += 16) {
        if (b0 < myc) {
          gather(x, b0 + lrow < myc ? s_in[myk][b0 + lrow] : -1);
          produce(lrow, kg);
        }
        __syncthreads();
        pull(r0, b0, lrow, kg);
        __syncthreads();
      }
      prefetch(r0 + kPullQ, lrow);
    }
  }

  // ---- epilogue
  if (dbg & 4) return;
  if constexpr (OUT_BF16) {
    // through LDS (all of it is free now): a lane holds 4 channels of a row, the stores want whole rows
    uint16_t* stage = (uint16_t*)smem;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int c = 0; c < HB; ++c) {
        u32x2 o;
        o.x = (uint32_t)ococc_f32_to_bf16(acc[rb][c][0]) | ((uint32_t)ococc_f32_to_bf16(acc[rb][c][1]) << 16);
        o.y = (uint32_t)ococc_f32_to_bf16(acc[rb][c][2]) | ((uint32_t)ococc_f32_to_bf16(acc[rb][c][3]) << 16);
        *(u32x2*)(stage + (WR * wrow + 16 * rb + lrow) * kStageLd + 16 * (HB * half + c) + 4 * kg) = o;
      }
    __syncthreads();
    constexpr int PPR = NC / 8;   // 16-byte pieces per row
    for (int p = threadIdx.x; p < T * PPR; p += kPullThreads) {
      const int r = p / PPR, q = p - r * PPR;
      if (row0 + r < n_out) *(u32x4*)((uint16_t*)out_ + (row0 + r) * NC + q * 8) = *(const u32x4*)(stage + r * kStageLd + q * 8);
    }
  } else {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int64_t row = row0 + WR * wrow + 16 * rb + lrow;
      if (row >= n_out) continue;
#pragma unroll
      for (int c = 0; c < HB; ++c) *(f32x4*)((float*)out_ + row * NC + 16 * (HB * half + c) + 4 * kg) = acc[rb][c];
    }
  }
}

int g_pull_dbg = 0;    // probe only (tools/probe/pull_conv_bench.py): bit 0 skip the sparse rounds, 1 the dense pass, 2 the stores

int g_pull_rows = 256;   // rows per tile: 256 (two workgroups per CU at 128 registers) or 512 (one, 210 registers); probe: ococc_sparse_conv_pull_probe(mask | rows << 8)

template <int KD, int NC, int T>
int launch_pull_t(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, int dense_k, const int32_t* table,
                  int64_t n_out, const float* bias, void* out, int out_dtype, hipStream_t stream) {
  // (grid rounded up to a multiple of 8 so that the XCD permutation inside the kernel is a bijection)
  const dim3 grid((unsigned)ococc_align_up(ococc_cdiv(n_out, T), 8));
  if (out_dtype == OCOCC_BF16)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(subm_pull_conv_kernel<KD, NC, true, T>), grid, dim3(kPullThreads), 0, stream, feat,
                       (uint32_t)(n_in * KD * 2), wn, kvol, dense_k, table, n_out, bias, out, g_pull_dbg);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(subm_pull_conv_kernel<KD, NC, false, T>), grid, dim3(kPullThreads), 0, stream, feat,
                       (uint32_t)(n_in * KD * 2), wn, kvol, dense_k, table, n_out, bias, out, g_pull_dbg);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
template <int KD, int NC>
int launch_pull(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, int dense_k, const int32_t* table,
                int64_t n_out, const float* bias, void* out, int out_dtype, hipStream_t stream) {
  if (g_pull_rows == 256)
    return launch_pull_t<KD, NC, 256>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream);
  return launch_pull_t<KD, NC, 512>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream);
}

}  // namespace

extern "C" void ococc_sparse_conv_pull_probe(int mask) {
  g_pull_dbg = mask & 0xff;
  if (mask >> 8) g_pull_rows = mask >> 8;
}

extern "C" int ococc_sparse_conv_pull_bf16(const uint16_t* feat, int64_t n_in, int32_t kd, const uint16_t* wn,
                                           int32_t kvol, int32_t ncols, const int32_t* table, int32_t dense_k,
                                           int64_t n_out, const float* bias, void* out, int32_t out_dtype,
                                           ococc_stream_t stream_) {
  OCOCC_REQUIRE(n_in >= 0 && n_out >= 0, "negative row count");
  OCOCC_REQUIRE(kvol >= 1 && kvol <= kPullMaxVol, "kernel volume must be 1..27");
  OCOCC_REQUIRE(dense_k >= -1 && dense_k < kvol, "dense_k must be -1 or an offset index");
  OCOCC_REQUIRE(out_dtype == OCOCC_BF16 || out_dtype == OCOCC_F32, "out_dtype must be f32/bf16");
  if (n_out == 0) return OCOCC_OK;
  OCOCC_REQUIRE(wn && table && out, "null pointer");
  OCOCC_REQUIRE(feat || n_in == 0, "null feat");
  OCOCC_REQUIRE(n_in * kd * 2 < 0xffffff00ll, "feat too large for the 32-bit buffer offsets of the gathers");
  hipStream_t stream = (hipStream_t)stream_;
  if (kd == 64 && ncols == 128) return launch_pull<64, 128>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream);
  if (kd == 64 && ncols == 64) return launch_pull<64, 64>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream);
  if (kd == 32 && ncols == 64) return launch_pull<32, 64>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream);
  if (kd == 128 && ncols == 64) return launch_pull<128, 64>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream);
  if (kd == 128 && ncols == 128) return launch_pull<128, 128>(feat, n_in, wn, kvol, dense_k, table, n_out, bias, out, out_dtype, stream);
  return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "shapes: 64 -> 128, 64 -> 64, 32 -> 64, 128 -> 64, 128 -> 128 channels");
}
