// B4 sparse convolution on gfx950: forward / dgrad (gather-GEMM) and wgrad.
//
// Reference formulation (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-456):
// for each of the K^3 kernel offsets run a gather kernel (reordering.cu.h:21-97),
// a dense GEMM and a scatter-add kernel (reordering.cu.h:99-157), moving
// P*(Cin+2*Cout) elements through HBM and syncing with the host once per layer.
//
// MI355X formulation (one launch, compulsory traffic only):
//   out[o] = sum_k feat[table[k][o]] @ W[k]        (output stationary)
//   * a wave owns 16*RB consecutive output rows; accumulators never leave
//     registers, so there is no scatter and no atomic: results are
//     deterministic and each output row is written exactly once;
//   * all kvol weight slices of a column slice are staged ONCE per workgroup
//     into LDS (<= 144 KB of the 160 KB) and stay there while the workgroup
//     walks its row tiles -- the k loop contains no barrier;
//   * gathered rows go from L2/HBM straight into MFMA operand registers: the
//     16x16x32 bf16 MFMA wants, per lane, 8 consecutive k of one row, which is
//     exactly one 16-byte load from the gathered feature row, so no LDS
//     staging or transposition of the activations is needed;
//   * the MFMA is issued with the weights as the A operand and the gathered
//     rows as B, so that a lane ends up with 4 consecutive output channels of
//     one voxel and stores 8 B (bf16) / 16 B (f32) at a time;
//   * per 16-row block a 32-bit mask (built with wave ballots by the rulebook
//     kernel) says which offsets have any neighbour; the others are skipped
//     with scalar branches.
// dgrad is the same kernel run on dY with the gather table of the other side
// and the weights in their stored [cin][cout] orientation.
//
// wgrad: dW[k] = sum_p x[in_p]^T dy[out_p] contracts over rulebook pairs, so it
// walks the compacted pair lists (32 pairs per MFMA k-step, all useful).  Rows
// are gathered into LDS row-major and read back with ds_read_b64_tr_b16, the
// gfx950 transposing LDS read, which yields MFMA operands whose k index runs
// over pairs.  Partial sums are written to per-workgroup slabs and added in a
// fixed order (deterministic, no float atomics).
#include "common.hpp"
#include "weight_layout.hpp"
#include "ln_math.hpp"
#include "param_reduce.hpp"
#include "stream_ops.hpp"
#include <cstdlib>

namespace {

constexpr int kConvThreads = 512;  // 8 waves, 2 per SIMD
constexpr int kConvWaves = kConvThreads / 64;
constexpr int kLdsBudget = 144 * 1024;

// A row of zeros in device memory: rows without a neighbour gather from here, so no per-element
// zero-select is needed after the loads (it stays L1 resident).
__device__ __attribute__((aligned(256))) uint16_t g_zero_row[256];

// Optional fused epilogue: LayerNorm (+ exact GELU) over the output row, the norm/act pair that
// make_sparse_convmodule puts behind the convolution (ops/sparse_block.py:216-289).  The row statistics
// are taken from the bf16-rounded conv output, exactly what the separate LN kernel would read.
struct LnArgs {
  const float* gamma;
  const float* beta;
  float eps;
  int act;             // 0 none, 1 GELU(erf)
  uint16_t* y;         // [n_out, ncols] bf16
  float* mean_rstd;    // [n_out, 2]
};

__device__ __forceinline__ float conv_gelu(float z) { return ln_gelu1(z); }  // (ln_math.hpp)
__device__ __forceinline__ float round_bf16(float v) { return ococc_bf16_to_f32(ococc_f32_to_bf16(v)); }
// sum over the 4 lanes (kg = 0..3) that share an output row
__device__ __forceinline__ float row_sum4(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ bf16x8 zero_bf16x8() {
  u32x4 z = {0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8, z);
}

template <int KD, int CS, int RB, int G, bool OUT_BF16, bool LN = false>
__global__ void __launch_bounds__(kConvThreads)
gather_gemm_kernel(const uint16_t* __restrict__ feat, const uint16_t* __restrict__ wn, int kvol,
                   int ncols, const int32_t* __restrict__ table,
                   const uint32_t* __restrict__ blockmask, int64_t n_out,
                   const float* __restrict__ bias, void* __restrict__ out_, LnArgs ln) {
  constexpr int KSTEPS = (KD + 31) / 32;
  constexpr int NB = CS / 16;
  constexpr int LDW = KD + 8;  // LDS row stride in elements (16 B pad)
  static_assert(RB * 16 <= 64, "a wave loads the indices of its whole tile with one instruction");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* wl = (uint16_t*)smem;

  const int cs0 = blockIdx.y * CS;
  // ---- stage this column slice of all kvol weight matrices into LDS ----
  // Loads are issued in groups of 8 per thread before any of them is consumed: one memory
  // latency per 64 KB instead of one per 8 KB.
  {
    constexpr int PPR = KD / 8;  // 16-byte pieces per row
    constexpr int UN = 8;
    const int total = kvol * CS * PPR;
    for (int base = 0; base < total; base += kConvThreads * UN) {
      u32x4 tmp[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * kConvThreads + threadIdx.x;
        tmp[u] = u32x4{0u, 0u, 0u, 0u};
        if (idx < total) {
          const int piece = idx % PPR, row = (idx / PPR) % CS, k = idx / (PPR * CS);
          tmp[u] = *(const u32x4*)(wn + ((int64_t)(k * ncols + cs0 + row)) * KD + piece * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * kConvThreads + threadIdx.x;
        if (idx < total) {
          const int piece = idx % PPR, row = (idx / PPR) % CS, k = idx / (PPR * CS);
          *(u32x4*)(wl + (k * CS + row) * LDW + piece * 8) = tmp[u];
        }
      }
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lrow = lane & 15;       // voxel within the 16-row block / weight row
  const int kg = lane >> 4;         // k group: elements 8*kg .. 8*kg+7 of a 32-deep k-step
  const int64_t n_blocks = (n_out + 15) >> 4;
  const int64_t n_tiles = (n_blocks + RB - 1) / RB;
  const uint32_t all_mask = kvol >= 32 ? 0xffffffffu : ((1u << kvol) - 1u);

  for (int64_t tile = (int64_t)blockIdx.x * kConvWaves + wave; tile < n_tiles;
       tile += (int64_t)gridDim.x * kConvWaves) {
    uint32_t m[RB];
    uint32_t um = 0;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int64_t blk = tile * RB + rb;
      uint32_t v = 0;
      if (blk < n_blocks) v = blockmask ? blockmask[blk] : all_mask;
      m[rb] = __builtin_amdgcn_readfirstlane(v);
      um |= m[rb];
    }
    f32x4 acc[RB][NB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Offsets are consumed in batches of G.  One coalesced load per offset brings the
    // gather indices of the whole tile (lane -> row); the next batch's indices are in
    // flight while this batch's rows are gathered and multiplied, and all row gathers
    // of a batch (up to G*RB*KSTEPS 16-byte loads per lane) are issued back to back.
    const int64_t row_l = tile * (RB * 16) + lane;
    const bool row_ok = lane < RB * 16 && row_l < n_out;
    int kcur[G], knxt[G];
    int32_t icur[G], inxt[G];
#pragma unroll
    for (int s = 0; s < G; ++s) {
      kcur[s] = -1;
      if (um) { kcur[s] = __builtin_ctz(um); um &= um - 1; }
      int32_t v = -1;
      if (kcur[s] >= 0 && row_ok) v = table[(int64_t)kcur[s] * n_out + row_l];
      icur[s] = v;
    }
    while (kcur[0] >= 0) {
      // x[s][rb][*] is written and read only under the same wave-uniform predicate
      // on(s, rb), so inactive (offset, block) pairs cost no instruction at all.
      bf16x8 x[G][RB][KSTEPS];
      uint32_t onmask = 0;  // bit s*RB+rb
#pragma unroll
      for (int s = 0; s < G; ++s) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if (kcur[s] >= 0 && ((m[rb] >> (kcur[s] & 31)) & 1u)) onmask |= 1u << (s * RB + rb);
        }
      }
      // Pass 1: issue every gather of the batch.  Loads are unconditional inside an active
      // (offset, block) pair -- rows without a neighbour read row 0 and are zeroed later -- so no
      // select or exec-masked move touches a register with a load in flight and the compiler
      // places ONE wait for the whole batch instead of one per load.
      // The index shuffles run unconditionally and first: they are the only consumers of the
      // (possibly still loading) index registers, so the single wait for them sits here, in
      // straight-line code, and not in front of every gather.
      int32_t ibv[G][RB];
#pragma unroll
      for (int s = 0; s < G; ++s) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          ibv[s][rb] = __shfl(icur[s], rb * 16 + lrow, 64);
        }
      }
#pragma unroll
      for (int s = 0; s < G; ++s) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if (onmask & (1u << (s * RB + rb))) {
            const int32_t ib = ibv[s][rb];
            const int kq = (KD % 32 == 0) ? kg * 8 : ((kg * 8 < KD) ? kg * 8 : 0);
            const uint16_t* src = (ib >= 0 ? feat + (int64_t)ib * KD : g_zero_row) + kq;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
              const int off = (KD % 32 == 0 || ks * 32 + kg * 8 < KD) ? ks * 32 : 0;
              x[s][rb][ks] = *(const bf16x8*)(src + off);
            }
          }
        }
      }
      // Next batch's indices: issued behind this batch's gathers so both are in flight together
      // (one memory latency per batch).
#pragma unroll
      for (int s = 0; s < G; ++s) {
        knxt[s] = -1;
        if (um) { knxt[s] = __builtin_ctz(um); um &= um - 1; }
        int32_t v = -1;
        if (knxt[s] >= 0 && row_ok) v = table[(int64_t)knxt[s] * n_out + row_l];
        inxt[s] = v;
      }
      // Pass 2: weights from LDS, zero-select, MFMA.
#pragma unroll
      for (int s = 0; s < G; ++s) {
        if (!((onmask >> (s * RB)) & ((1u << RB) - 1u))) continue;
        const int k = kcur[s];
        bf16x8 w[KSTEPS][NB];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const int koff = ks * 32 + kg * 8;
#pragma unroll
          for (int cb = 0; cb < NB; ++cb) {
            if (KD % 32 == 0) {
              w[ks][cb] = *(const bf16x8*)(wl + (k * CS + cb * 16 + lrow) * LDW + koff);
            } else {
              const bf16x8 t = *(const bf16x8*)(wl + (k * CS + cb * 16 + lrow) * LDW + (koff < KD ? koff : 0));
              w[ks][cb] = koff < KD ? t : zero_bf16x8();
            }
          }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if (onmask & (1u << (s * RB + rb))) {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
              // k positions beyond KD (KD = 16 only) multiply zero weights: no select on x needed
              const bf16x8 xv = x[s][rb][ks];
#pragma unroll
              for (int cb = 0; cb < NB; ++cb)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks][cb], xv, acc[rb][cb], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int s = 0; s < G; ++s) {
        kcur[s] = knxt[s];
        icur[s] = inxt[s];
      }
    }

    // ---- epilogue: lane holds channels cs0 + cb*16 + 4*kg + {0..3} of voxel lrow ----
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int64_t r = (tile * RB + rb) * 16 + lrow;
      if (r >= n_out) continue;
      float rs = 0.f;
#pragma unroll
      for (int cb = 0; cb < NB; ++cb) {
        const int ch = cs0 + cb * 16 + kg * 4;
        f32x4 v = acc[rb][cb];
        if (bias) {
          const f32x4 b = *(const f32x4*)(bias + ch);
          v += b;
        }
        if (OUT_BF16) {
          u32x2 p;
          p.x = (uint32_t)ococc_f32_to_bf16(v.x) | ((uint32_t)ococc_f32_to_bf16(v.y) << 16);
          p.y = (uint32_t)ococc_f32_to_bf16(v.z) | ((uint32_t)ococc_f32_to_bf16(v.w) << 16);
          *(u32x2*)((uint16_t*)out_ + r * ncols + ch) = p;
        } else {
          *(f32x4*)((float*)out_ + r * ncols + ch) = v;
        }
        if (LN) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = round_bf16(v[j]);
            rs += v[j];
          }
          acc[rb][cb] = v;
        }
      }
      if (LN) {  // single column slice (CS == ncols): the 4 kg lanes of a row hold all of it
        const float mean = row_sum4(rs) * (1.f / CS);
        float sq = 0.f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = acc[rb][cb][j] - mean;
            sq += d * d;
          }
        const float rstd = rsqrtf(row_sum4(sq) * (1.f / CS) + ln.eps);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
          const int ch = cb * 16 + kg * 4;
          const f32x4 gm = *(const f32x4*)(ln.gamma + ch), bt = *(const f32x4*)(ln.beta + ch);
          f32x4 z;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float t = (acc[rb][cb][j] - mean) * rstd * gm[j] + bt[j];
            z[j] = ln.act == 1 ? conv_gelu(t) : t;
          }
          u32x2 p;
          p.x = (uint32_t)ococc_f32_to_bf16(z.x) | ((uint32_t)ococc_f32_to_bf16(z.y) << 16);
          p.y = (uint32_t)ococc_f32_to_bf16(z.z) | ((uint32_t)ococc_f32_to_bf16(z.w) << 16);
          *(u32x2*)(ln.y + r * ncols + ch) = p;
        }
        if (kg == 0) {
          ln.mean_rstd[r * 2] = mean;
          ln.mean_rstd[r * 2 + 1] = rstd;
        }
      }
    }
  }
}

template <int KD, int CS, int RB>
int launch_gather_gemm(const uint16_t* feat, const uint16_t* wn, int kvol, int ncols,
                       const int32_t* table, const uint32_t* blockmask, int64_t n_out,
                       const float* bias, void* out, int out_dtype, hipStream_t stream,
                       const LnArgs* ln = nullptr) {
  const int lds = kvol * CS * (KD + 8) * 2;
  const int n_slices = ncols / CS;
  const int64_t n_tiles = ococc_cdiv(ococc_cdiv(n_out, 16), RB);
  int occ = (160 * 1024) / (lds + 512);
  if (occ < 1) occ = 1;
  if (occ > 2) occ = 2;  // 512-thread blocks: at most 2 fit the 32-wave CU comfortably
  int64_t gx = (int64_t)(256 * occ) / n_slices;
  if (gx < 1) gx = 1;
  const int64_t need = ococc_cdiv(n_tiles, kConvWaves);
  if (gx > need) gx = need;
  if (gx < 1) gx = 1;
  dim3 grid((unsigned)gx, (unsigned)n_slices);
  if (ln) {
    if (n_slices != 1 || out_dtype != OCOCC_BF16) return -1;  // no fused instantiation
    auto kern = gather_gemm_kernel<KD, CS, RB, (KD >= 128 ? 2 : 4), true, true>;
    OCOCC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, grid, dim3(kConvThreads), lds, stream, feat, wn, kvol, ncols, table,
                       blockmask, n_out, bias, out, *ln);
    OCOCC_CHECK_LAUNCH();
    return OCOCC_OK;
  }
  if (out_dtype == OCOCC_BF16) {
    auto kern = gather_gemm_kernel<KD, CS, RB, (KD >= 128 ? 2 : 4), true>;
    OCOCC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, grid, dim3(kConvThreads), lds, stream, feat, wn, kvol, ncols, table,
                       blockmask, n_out, bias, out, LnArgs{});
  } else {
    auto kern = gather_gemm_kernel<KD, CS, RB, (KD >= 128 ? 2 : 4), false>;
    OCOCC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, grid, dim3(kConvThreads), lds, stream, feat, wn, kvol, ncols, table,
                       blockmask, n_out, bias, out, LnArgs{});
  }
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

// ---------------------------------------------------------------- full-width variant
// When the kvol weight matrices of ALL output columns do not fit LDS the kernel above has to cut
// the columns into slices, and every slice re-reads the gather indices and re-gathers the input
// rows (4x on the 64->128 layer).  Here a workgroup of 4 waves owns 256 consecutive output rows at
// full width and the weights stream through LDS one kernel offset at a time (double buffered,
// global_load_lds so they never occupy registers, one barrier per offset): rows are gathered once,
// indices are read once, and every gathered 16-byte fragment feeds NC/16 MFMAs.
// Per offset k a wave issues the gathers of k+1, the weight DMA of k+1 and the index load of k+2,
// then runs the MFMAs of k out of registers / LDS filled one iteration earlier.
// Measured anatomy on the 64->128 layer (tools/prof_conv.py, 126 k rows, 27 offsets): ~20 us are
// prologue + the 32 MB output store, the rest is the per-offset loop, which is bound by MFMA issue
// and scalar bookkeeping and NOT by memory latency -- hence (i) the MFMAs of a 16-row block are
// skipped with ONE scalar branch per 8 MFMAs when the block has no neighbour at the offset, and
// (ii) gathers are unconditional buffer loads (out-of-range offset -> zeros, no memory access).
constexpr int kStreamThreads = 256;
constexpr int kStreamRB = 4;  // 16-row blocks per wave (2 was measured slower: 55 vs 46 us)

// (the LayerNorm epilogue needs a few more registers: at two workgroups per CU the 64 -> 128 and 128 -> 64 variants
// park two values (lane row / k-group) in scratch AROUND the loop -- stored before it, reloaded in its exit block
// ahead of the explicit vmcnt(0); nothing inside the loop, which tools/check_stream_isa.py verifies.  Loads return in
// order among themselves, so the two stores only make the hand-counted waits more conservative.  Fused at two
// workgroups per CU the 64 -> 128 layer takes ~6 us MORE than conv + separate LayerNorm launch (0.298 against 0.292 ms
// per step): the epilogue's 128 GELUs per lane run at 8 waves per CU; it stays opt-in, OCOCC_FUSE_CONV_LN=1)
template <int KD, int NC, bool OUT_BF16, bool LN = false>
__global__ void __launch_bounds__(kStreamThreads, 2)
gather_gemm_stream_kernel(const uint16_t* __restrict__ feat, uint32_t feat_bytes,
                          const uint16_t* __restrict__ wn, int kvol, const int32_t* __restrict__ table,
                          const uint32_t* __restrict__ blockmask, int64_t n_out,
                          const float* __restrict__ bias, void* __restrict__ out_, LnArgs ln) {
  constexpr int RB = kStreamRB;
  constexpr int KSTEPS = KD / 32;
  constexpr int NB = NC / 16;
  constexpr int PPR = KD / 8;             // 16-byte pieces per weight row
  constexpr int RPB = 256 / (KD * 2);     // weight rows per 256-byte LDS bank row
  constexpr int PIECES = NC * PPR;        // per offset
  constexpr int CHUNKS = PIECES / 64;     // 1 KB chunks, one global_load_lds_dwordx4 each
  constexpr int NWAVES = kStreamThreads / 64;
  constexpr int CPW = (CHUNKS + NWAVES - 1) / NWAVES;
  constexpr int NG = RB * KSTEPS;         // gathers per offset (the loads younger than the weight DMA)
  static_assert(KD % 32 == 0 && PIECES % 64 == 0 && NB % 2 == 0, "stream kernel shape");
  // weights of one offset, row-major [NC][KD] in 16-byte pieces; piece j of row r sits in slot
  // j ^ swz(r) of its row so that the 16 rows an MFMA operand read touches hit 16 different bank
  // groups without padding (global_load_lds writes lane i at base + 16 i: no room for pad bytes)
  __shared__ __attribute__((aligned(16))) u32x4 wl[2][PIECES];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lrow = lane & 15;
  const int kg = lane >> 4;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Give XCD x the x-th
  // contiguous eighth of the rows: a voxel's neighbours are a few thousand rows away at most, so
  // the rows one XCD gathers are (almost) the rows it owns and stay in its 4 MB L2.
  // (gridDim.x is a multiple of 8, see the launcher.)
  const int wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if ((int64_t)wg * (NWAVES * RB * 16) >= n_out) return;  // padding workgroup: leaves as a whole
  const int64_t row0 = (int64_t)wg * (NWAVES * RB * 16) + wave * (RB * 16);
  const int64_t n_blocks = (n_out + 15) >> 4;
  const bool nomask = blockmask == nullptr;

  uint32_t m[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int64_t blk = (row0 >> 4) + rb;
    uint32_t v = 0;
    if (blk < n_blocks) v = nomask ? 0xffffffffu : blockmask[blk];
    m[rb] = __builtin_amdgcn_readfirstlane(v);
  }
  const int64_t row_l = row0 + lane;
  const int64_t row_c = row_l < n_out ? row_l : (n_out - 1);

  f32x4 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto swz = [](int row) -> int { return (row / RPB) & (PPR - 1); };
  // MFMA block cb, row i (= 4*kg' + r of the accumulator layout) carries output channel
  // chan(cb, i): blocks 2p and 2p+1 interleave in groups of 4 so that a lane's two accumulators
  // hold 8 CONSECUTIVE channels 32p + 8kg .. +7 and the epilogue stores 16 bytes per lane.
  auto chan = [](int cb, int i) -> int { return (cb >> 1) * 32 + (i >> 2) * 8 + (cb & 1) * 4 + (i & 3); };
  // global -> LDS without a register hop; CPW loads per wave (waves past CHUNKS re-load a chunk
  // with identical data, to keep the per-wave load count fixed)
  auto stage_w = [&](int k, int buf) {
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
      int c = u * NWAVES + wave;
      if (CHUNKS % NWAVES != 0 && c >= CHUNKS) c = c % CHUNKS;
      const int q = c * 64 + lane;
      const int row = q / PPR, slot = q % PPR;
      const uint16_t* src = wn + (int64_t)k * NC * KD + (row * PPR + (slot ^ swz(row))) * 8;
      const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(&wl[buf][c * 64]);
      stream_dma_b128(src, __builtin_amdgcn_readfirstlane(dst));
    }
  };
  // Unconditional buffer loads: a row without a neighbour at this offset gets an out-of-range
  // offset, for which the buffer unit returns zeros WITHOUT touching memory.
  i32x4 frs;  // raw buffer descriptor: base, stride 0, size in bytes, 32-bit data format
  frs.x = (int)(uint32_t)(uintptr_t)feat;
  frs.y = (int)(uint32_t)((uintptr_t)feat >> 32);
  frs.z = (int)feat_bytes;
  frs.w = 0x00020000;
  static_assert(KSTEPS <= 4, "gather spells the k-steps out");
  auto gather = [&](u32x4 (&x)[RB][KSTEPS], int32_t idx) {
    int32_t ibv[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) ibv[rb] = __shfl(idx, rb * 16 + lrow, 64);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const uint32_t off = ibv[rb] >= 0 ? (uint32_t)ibv[rb] * (KD * 2) + kg * 16 : 0xffffff00u;
      x[rb][0] = stream_buffer_load<0>(frs, off);
      if constexpr (KSTEPS > 1) x[rb][1] = stream_buffer_load<64>(frs, off);
      if constexpr (KSTEPS > 2) x[rb][2] = stream_buffer_load<128>(frs, off);
      if constexpr (KSTEPS > 3) x[rb][3] = stream_buffer_load<192>(frs, off);
    }
  };
  auto tie_rows = [&](u32x4 (&x)[RB][KSTEPS]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) stream_tie(x[rb][ks]);
  };
  auto keep_rows = [&](const u32x4 (&x)[RB][KSTEPS]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) stream_keep(x[rb][ks]);
  };
  // indices of offset k for this wave's 64 rows (lane -> row); offsets past the end re-read the last
  // one (the result is never used) so that the load count per iteration stays fixed
  const int32_t* tcol = table + row_c;  // rows past the end copy the last row; never stored
  auto load_idx = [&](int k) -> int32_t {
    const int kk = k < kvol ? k : kvol - 1;
    return stream_load_i32(tcol + (int64_t)kk * n_out);
  };
  // LDS byte offsets of this lane's weight fragments, split so that the k-step enters with one XOR:
  // slot = (ks*4 + kg) ^ swz = ((ks*4) ^ (swz & ~3)) + (kg ^ (swz & 3)); swz has period 16 in the row
  uint32_t wb[2], wh[2];
#pragma unroll
  for (int c1 = 0; c1 < 2; ++c1) {
    const int row = chan(c1, lrow);
    const int sw = swz(row);
    wb[c1] = (uint32_t)(row * PPR + (kg ^ (sw & 3))) * 16u;
    wh[c1] = (uint32_t)(sw & ~3) * 16u;
  }
  auto mma = [&](const u32x4 (&x)[RB][KSTEPS], int buf, int k) {
    if (k >= kvol) return;  // the padding half-iteration of an odd kernel volume
    // (opaque to the optimiser: otherwise all NB*KSTEPS fragment addresses are hoisted out of the loop,
    // kept in registers and spilled)
    stream_tie(wb[0]);
    stream_tie(wb[1]);
    stream_tie(wh[0]);
    stream_tie(wh[1]);
    // 4-bit activity of the wave's 16-row blocks at this offset
    uint32_t act = 0xfu;
    if (!nomask) {
      act = 0;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) act |= ((m[rb] >> (k & 31)) & 1u) << rb;
    }
    if (!act) return;
    // weight fragments are fetched 8 at a time (FG k-steps x NB column blocks), then each active
    // 16-row block runs its 8 MFMAs behind one scalar branch
    constexpr int FG = (NB >= 8) ? 1 : ((8 / NB) < KSTEPS ? (8 / NB) : KSTEPS);
#pragma unroll
    for (int k0 = 0; k0 < KSTEPS; k0 += FG) {
      bf16x8 w[FG][NB];
#pragma unroll
      for (int f = 0; f < FG; ++f)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
          const uint32_t a = wb[cb & 1] + (uint32_t)((cb >> 1) * 32 * PPR * 16) + ((uint32_t)((k0 + f) * 64) ^ wh[cb & 1]);
          w[f][cb] = __builtin_bit_cast(bf16x8, *(const u32x4*)((const char*)&wl[buf][0] + a));
        }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if ((act >> rb) & 1u) {
#pragma unroll
          for (int f = 0; f < FG; ++f)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb)
              acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[f][cb], __builtin_bit_cast(bf16x8, x[rb][k0 + f]),
                                                                    acc[rb][cb], 0, 0, 0);
        }
      }
    }
  };

  u32x4 xa[RB][KSTEPS], xb[RB][KSTEPS];
  constexpr int NI = 1 + CPW + NG;  // vector-memory operations one offset issues: index, weight DMA, gathers
  // prologue: weights of offset 0 into buffer 0, rows of offset 0 into xa, indices of offset 1;
  // everything has landed before the loop (no load in flight across the loop entry)
  int32_t i_nxt = load_idx(0);
  int32_t i_one = load_idx(1);
  stage_w(0, 0);
  stream_wait_vm<CPW + 1>();
  stream_tie(i_nxt);
  gather(xa, i_nxt);
  wait_vmcnt_barrier<0>();
  stream_tie(i_one);
  tie_rows(xa);
  i_nxt = i_one;

  const int klast = kvol - 1;
  // kvol rounded up to even: ONE loop exit; the padding half-iteration still issues its (clamped) loads
  for (int k = 0; k < kvol; k += 2) {
    // ---- offset k: operands xa / buffer 0; prefetch k+1 into xb / buffer 1 ----
    {
      int32_t i_nn = load_idx(k + 2);
      stage_w(k + 1 < kvol ? k + 1 : klast, 1);
      gather(xb, i_nxt);
      stream_wait_vm<NI>();  // everything older than this offset's own loads: the rows of offset k
      tie_rows(xa);
      mma(xa, 0, k);
      wait_vmcnt_barrier<NG>();  // index of k+2 and weights of k+1 landed, buffer 0 is free again
      stream_tie(i_nn);
      i_nxt = i_nn;
    }
    // ---- offset k+1: operands xb / buffer 1; prefetch k+2 into xa / buffer 0 ----
    {
      int32_t i_nn = load_idx(k + 3);
      stage_w(k + 2 < kvol ? k + 2 : klast, 0);
      gather(xa, i_nxt);
      stream_wait_vm<NI>();
      tie_rows(xb);
      mma(xb, 1, k + 1);
      wait_vmcnt_barrier<NG>();
      stream_tie(i_nn);
      i_nxt = i_nn;
    }
  }
  // the surplus prefetch of the last half-iteration: its destination registers stay allocated until it is in
  stream_wait_vm<0>();
  keep_rows(xa);
  keep_rows(xb);
  stream_keep(i_nxt);

  // ---- epilogue: lane holds channels 32p + 8kg .. +7 of voxel lrow in acc[rb][2p], acc[rb][2p+1] ----
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int64_t r = row0 + rb * 16 + lrow;
    if (r >= n_out) continue;
    float rs = 0.f;
#pragma unroll
    for (int p = 0; p < NB / 2; ++p) {
      const int ch = p * 32 + kg * 8;
      f32x4 v0 = acc[rb][2 * p], v1 = acc[rb][2 * p + 1];
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
        *(u32x4*)((uint16_t*)out_ + r * NC + ch) = q;
      } else {
        *(f32x4*)((float*)out_ + r * NC + ch) = v0;
        *(f32x4*)((float*)out_ + r * NC + ch + 4) = v1;
      }
      if (LN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v0[j] = round_bf16(v0[j]);
          v1[j] = round_bf16(v1[j]);
          rs += v0[j] + v1[j];
        }
        acc[rb][2 * p] = v0;
        acc[rb][2 * p + 1] = v1;
      }
    }
    if (LN) {
      const float mean = row_sum4(rs) * (1.f / NC);
      float sq = 0.f;
#pragma unroll
      for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = acc[rb][cb][j] - mean;
          sq += d * d;
        }
      const float rstd = rsqrtf(row_sum4(sq) * (1.f / NC) + ln.eps);
#pragma unroll
      for (int p = 0; p < NB / 2; ++p) {
        const int ch = p * 32 + kg * 8;
        float z[8];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x4 gm = *(const f32x4*)(ln.gamma + ch + 4 * h2), bt = *(const f32x4*)(ln.beta + ch + 4 * h2);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float t = (acc[rb][2 * p + h2][j] - mean) * rstd * gm[j] + bt[j];
            z[4 * h2 + j] = ln.act == 1 ? conv_gelu(t) : t;
          }
        }
        u32x4 q;
        q.x = (uint32_t)ococc_f32_to_bf16(z[0]) | ((uint32_t)ococc_f32_to_bf16(z[1]) << 16);
        q.y = (uint32_t)ococc_f32_to_bf16(z[2]) | ((uint32_t)ococc_f32_to_bf16(z[3]) << 16);
        q.z = (uint32_t)ococc_f32_to_bf16(z[4]) | ((uint32_t)ococc_f32_to_bf16(z[5]) << 16);
        q.w = (uint32_t)ococc_f32_to_bf16(z[6]) | ((uint32_t)ococc_f32_to_bf16(z[7]) << 16);
        *(u32x4*)(ln.y + r * NC + ch) = q;
      }
      if (kg == 0) {
        ln.mean_rstd[r * 2] = mean;
        ln.mean_rstd[r * 2 + 1] = rstd;
      }
    }
  }
}

template <int KD, int NC>
int launch_gather_gemm_stream(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, const int32_t* table,
                              const uint32_t* blockmask, int64_t n_out, const float* bias, void* out,
                              int out_dtype, hipStream_t stream, const LnArgs* ln = nullptr) {
  // (grid rounded up to a multiple of 8 so that the XCD permutation inside the kernel is a bijection)
  const dim3 grid((unsigned)ococc_align_up(ococc_cdiv(n_out, kStreamThreads / 64 * kStreamRB * 16), 8));
  if (ln) {
    if (out_dtype != OCOCC_BF16) return -1;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gather_gemm_stream_kernel<KD, NC, true, true>), grid, dim3(kStreamThreads), 0,
                       stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, table, blockmask, n_out, bias, out, *ln);
  } else if (out_dtype == OCOCC_BF16)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gather_gemm_stream_kernel<KD, NC, true>), grid, dim3(kStreamThreads), 0,
                       stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, table, blockmask, n_out, bias, out, LnArgs{});
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gather_gemm_stream_kernel<KD, NC, false>), grid, dim3(kStreamThreads), 0,
                       stream, feat, (uint32_t)(n_in * KD * 2), wn, kvol, table, blockmask, n_out, bias, out, LnArgs{});
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

// Returns -1 when the shape has no full-width instantiation.
template <int KD>
int dispatch_stream(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, int ncols, const int32_t* table,
                    const uint32_t* blockmask, int64_t n_out, const float* bias, void* out, int out_dtype,
                    hipStream_t stream, const LnArgs* ln = nullptr) {
  if constexpr (KD % 32 != 0) {
    return -1;
  } else {
    if (n_in * KD * 2 >= 0xffffff00ll) return -1;  // the gathers address feat through a 32-bit buffer offset
    switch (ncols) {
      case 32: return launch_gather_gemm_stream<KD, 32>(feat, n_in, wn, kvol, table, blockmask, n_out, bias, out, out_dtype, stream, ln);
      case 64: return launch_gather_gemm_stream<KD, 64>(feat, n_in, wn, kvol, table, blockmask, n_out, bias, out, out_dtype, stream, ln);
      case 128:
        if constexpr (KD <= 64)
          return launch_gather_gemm_stream<KD, 128>(feat, n_in, wn, kvol, table, blockmask, n_out, bias, out, out_dtype, stream, ln);
        else
          return -1;
      default: return -1;
    }
  }
}

template <int KD>
int dispatch_cs(const uint16_t* feat, int64_t n_in, const uint16_t* wn, int kvol, int ncols, const int32_t* table,
                const uint32_t* blockmask, int64_t n_out, const float* bias, void* out,
                int out_dtype, hipStream_t stream, const LnArgs* ln = nullptr) {
  // widest column slice whose kvol weight slices fit the LDS budget
  const int per_col = kvol * (KD + 8) * 2;
  int cs = 0;
  const int cands[3] = {64, 32, 16};
  for (int i = 0; i < 3; ++i)
    if (ncols % cands[i] == 0 && per_col * cands[i] <= kLdsBudget) {
      cs = cands[i];
      break;
    }
  // (the streamed-weights kernel was also tried on the small layers: 21.6 / 25.3 us against 21.1 / 21.2 us
  // for the resident-weights kernel, so it is only used where slicing would be needed)
  if (cs != 0 && (ncols / cs > 1 || (ln && cs != ncols))) {
    const int rc = dispatch_stream<KD>(feat, n_in, wn, kvol, ncols, table, blockmask, n_out, bias, out, out_dtype,
                                       stream, ln);
    if (rc >= 0) return rc;
  }
  int rc = -1;
  if (cs == 64)
    rc = launch_gather_gemm<KD, 64, (KD >= 128 ? 2 : 4)>(feat, wn, kvol, ncols, table, blockmask, n_out, bias, out,
                                                        out_dtype, stream, ln);
  else if (cs == 32)
    rc = launch_gather_gemm<KD, 32, (KD <= 16 ? 2 : 4)>(feat, wn, kvol, ncols, table, blockmask, n_out, bias, out, out_dtype, stream, ln);
  else if (cs == 16)
    rc = launch_gather_gemm<KD, 16, 4>(feat, wn, kvol, ncols, table, blockmask, n_out, bias, out, out_dtype, stream, ln);
  if (rc >= 0) return rc;
  if (ln && cs != 0)
    return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "no fused LayerNorm epilogue for this shape (use the separate calls)");
  return ococc_fail(OCOCC_EUNSUPPORTED, __func__,
                    "kernel volume x channels does not fit the LDS-resident weight plan");
}

// ---------------------------------------------------------------- weights
// (index maps of the operand layouts: weight_layout.hpp)
__device__ __forceinline__ int64_t prep_dest(int mode, int64_t i, int k, int r, int c, int ncols, int kd) {
  return ococc_prep_dest(mode, i, k, r, c, ncols, kd);
}
template <typename T>
__global__ void __launch_bounds__(256)
weight_prepare_kernel(const T* __restrict__ w, int kvol, int cin, int cout, int mode,
                      uint16_t* __restrict__ wn) {
  const int64_t total = (int64_t)kvol * cin * cout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    // i indexes the destination
    int64_t src, dst;
    const int base = mode & 3;
    if (base == 0) {  // wn[k][co][ci] = W[k][ci][co]
      const int ci = (int)(i % cin);
      const int co = (int)((i / cin) % cout);
      const int k = (int)(i / ((int64_t)cin * cout));
      src = ((int64_t)k * cin + ci) * cout + co;
      dst = prep_dest(mode, i, k, co, ci, cout, cin);
    } else {  // wn[k][ci][co] = W[k' ][ci][co]
      const int64_t rem = i % ((int64_t)cin * cout);
      const int k = (int)(i / ((int64_t)cin * cout));
      const int ks = base == 1 ? kvol - 1 - k : k;
      src = (int64_t)ks * cin * cout + rem;
      dst = prep_dest(mode, i, k, (int)(rem / cout), (int)(rem % cout), cin, cout);
    }
    float v;
    if (sizeof(T) == 4)
      v = ((const float*)w)[src];
    else
      v = ococc_bf16_to_f32(((const uint16_t*)w)[src]);
    wn[dst] = ococc_f32_to_bf16(v);
  }
}

// several (weight, mode) pairs in one launch: one graph node instead of one per conv call
constexpr int kMaxPrep = 16;
struct PrepPack {
  const float* w[kMaxPrep];
  uint16_t* wn[kMaxPrep];
  int32_t kvol[kMaxPrep], cin[kMaxPrep], cout[kMaxPrep], mode[kMaxPrep];
  int32_t first_block[kMaxPrep + 1];
  int32_t count;
};

__global__ void __launch_bounds__(256) weight_prepare_multi_kernel(PrepPack pk) {
  int t = 0;
  while (t + 1 < pk.count && (int)blockIdx.x >= pk.first_block[t + 1]) ++t;
  const int cin = pk.cin[t], cout = pk.cout[t], kvol = pk.kvol[t], mode = pk.mode[t];
  // 32-bit index arithmetic (a weight tensor has far fewer than 2^31 elements; the launcher checks): the 64-bit
  // divisions of the general kernel were most of this launch
  const uint32_t total = (uint32_t)kvol * cin * cout, kc = (uint32_t)cin * cout;
  const uint32_t nblk = pk.first_block[t + 1] - pk.first_block[t];
  const int base = mode & 3;
  for (uint32_t i = (blockIdx.x - pk.first_block[t]) * 256u + threadIdx.x; i < total; i += nblk * 256u) {
    const uint32_t k = i / kc, rem = i - k * kc;
    uint32_t src;
    int64_t dst;
    if (base == 0) {
      const uint32_t co = rem / cin, ci = rem - co * cin;
      src = (k * cin + ci) * cout + co;
      dst = prep_dest(mode, i, (int)k, (int)co, (int)ci, cout, cin);
    } else {
      const uint32_t ks = base == 1 ? kvol - 1 - k : k;
      src = ks * kc + rem;
      const uint32_t r = rem / cout;
      dst = prep_dest(mode, i, (int)k, (int)r, (int)(rem - r * cout), cin, cout);
    }
    pk.wn[t][dst] = ococc_f32_to_bf16(pk.w[t][src]);
  }
}

// ---------------------------------------------------------------- wgrad
constexpr int kWgThreads = 256;
constexpr int kWgGrid = 2048;    // workgroups of the weight-gradient launch (8 per CU); items beyond that are looped over
constexpr int kWgSteps = 16;  // 32-pair MFMA k-steps per work item (one slab per item)
constexpr int kWgDepth = 4;   // steps whose rows are in flight (registers) ahead of the one being multiplied

// Flattened (offset, part) work list: offset k owns ceil(ceil(num[k]/32) / kWgSteps) consecutive
// items.  Every wave locates its item with one coalesced load of num[] and a wave scan, so the
// thousands of surplus workgroups of the worst-case grid leave after ~30 instructions.
__device__ __forceinline__ int wg_items(int32_t n) { return (((n + 31) >> 5) + kWgSteps - 1) / kWgSteps; }

__device__ __forceinline__ bool wg_locate(const int32_t* __restrict__ num, int kvol, int g, int* k_out,
                                          int* part_out, int* base_out) {
  const int lane = threadIdx.x & 63;
  int base = 0;
  for (int k0 = 0; k0 < kvol; k0 += 64) {
    const int k = k0 + lane;
    const int cnt = k < kvol ? wg_items(num[k]) : 0;
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    const int tot = __shfl(inc, 63, 64);
    if (g < base + tot) {
      const int excl = base + inc - cnt;
      const unsigned long long hit = __ballot(cnt > 0 && g >= excl && g < excl + cnt);
      const int src = __ffsll((long long)hit) - 1;
      const int b = __shfl(excl, src, 64);
      *k_out = k0 + src;
      *base_out = b;
      *part_out = g - b;
      return true;
    }
    base += tot;
  }
  return false;
}

// number of items before offset k, and of offset k itself
__device__ __forceinline__ void wg_span(const int32_t* __restrict__ num, int k, int* base_out, int* count_out) {
  const int lane = threadIdx.x & 63;
  int s = 0;
  for (int j0 = 0; j0 < k; j0 += 64) {
    const int j = j0 + lane;
    int c = j < k ? wg_items(num[j]) : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    s += c;
  }
  *base_out = s;
  *count_out = wg_items(num[k]);
}

template <int CIN, int COUT>
__device__ __forceinline__ void wgrad_body(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                           const int32_t* __restrict__ pairs, const int32_t* __restrict__ num, int kvol,
                                           int64_t cap, float* __restrict__ slabs, int first_item, int item_stride,
                                           uint16_t* lds_base, int32_t* sidx_base) {
  constexpr int MB = CIN / 16, NB = COUT / 16;
  constexpr int WN = NB >= 4 ? 4 : NB;  // waves along the cout blocks
  constexpr int WM = 4 / WN;            // waves along the cin blocks
  constexpr int MBW = (MB + WM - 1) / WM, NBW = (NB + WN - 1) / WN;
  constexpr int LDX = CIN + 8, LDY = COUT + 8;  // LDS row strides (elements)
  constexpr int PX = 32 * CIN / 8, PY = 32 * COUT / 8;  // 16-byte pieces per step
  constexpr int PT = (PX + PY + kWgThreads - 1) / kWgThreads;
  // (LDS handed in by the kernel: several bodies in one launch share the space of the largest)
  uint16_t (*lds)[32 * LDX + 32 * LDY] = (uint16_t (*)[32 * LDX + 32 * LDY])lds_base;   // [2]: the step buffers
  int32_t (*sidx)[kWgSteps * 32] = (int32_t (*)[kWgSteps * 32])sidx_base;               // [2]: the item's pair list: input rows, output rows

  // Work items (slabs) are dealt round-robin to a grid that no longer has to cover the worst case
  // kvol * cap / 256 of them: on sparse grids most of those workgroups only found out that there was
  // nothing for them (13.5 k launches for 0.9 k items, ~10 us per call).
  for (int item = first_item;; item += item_stride) {
  int k, part, base;
  if (!wg_locate(num, kvol, item, &k, &part, &base)) return;
  if (item != first_item) __syncthreads();  // the step buffers of the previous item are free
  const int nk = num[k];
  const int ksteps = (nk + 31) >> 5;
  const int first = part * kWgSteps;
  const int last = (first + kWgSteps < ksteps) ? first + kWgSteps : ksteps;
  const int32_t* pin = pairs + ((int64_t)k * 2 + 0) * cap;
  const int32_t* pout = pairs + ((int64_t)k * 2 + 1) * cap;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int grp = lane >> 4, li = lane & 15;

  f32x4 acc[MBW][NBW];
#pragma unroll
  for (int i = 0; i < MBW; ++i)
#pragma unroll
    for (int j = 0; j < NBW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The item's pair indices go to LDS in one coalesced pass; the rows of kWgDepth steps are then in flight
  // in registers ahead of the step being multiplied (with one step ahead every step cost a full global-load
  // latency, which is why items had to stay short and the slab count high).
  for (int p = threadIdx.x; p < (last - first) * 32; p += kWgThreads) {
    const int gp = first * 32 + p;
    sidx[0][p] = gp < nk ? pin[gp] : -1;
    sidx[1][p] = gp < nk ? pout[gp] : -1;
  }
  __syncthreads();
  u32x4 stage[kWgDepth][PT];
  auto load_step = [&](u32x4 (&st)[PT], int step) {  // rows of `step` (item-relative) -> registers
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int pc = threadIdx.x + t * kWgThreads;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (step < last - first) {
        if (pc < PX) {
          const int32_t r = sidx[0][step * 32 + pc / (CIN / 8)];
          if (r >= 0) v = *(const u32x4*)(x + (int64_t)r * CIN + (pc % (CIN / 8)) * 8);
        } else if (pc < PX + PY) {
          const int32_t r = sidx[1][step * 32 + (pc - PX) / (COUT / 8)];
          if (r >= 0) v = *(const u32x4*)(dy + (int64_t)r * COUT + ((pc - PX) % (COUT / 8)) * 8);
        }
      }
      st[t] = v;
    }
  };
  auto store_step = [&](int buf, const u32x4 (&st)[PT]) {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int pc = threadIdx.x + t * kWgThreads;
      if (pc < PX) {
        const int row = pc / (CIN / 8), piece = pc % (CIN / 8);
        *(u32x4*)(&lds[buf][row * LDX + piece * 8]) = st[t];
      } else if (pc < PX + PY) {
        const int pc2 = pc - PX;
        const int row = pc2 / (COUT / 8), piece = pc2 % (COUT / 8);
        *(u32x4*)(&lds[buf][32 * LDX + row * LDY + piece * 8]) = st[t];
      }
    }
  };

#pragma unroll
  for (int d = 0; d < kWgDepth; ++d) load_step(stage[d], d);
  const int nsteps = last - first;
  // groups of kWgDepth steps so that the register ring has compile-time positions
  for (int s0 = 0; s0 < nsteps; s0 += kWgDepth) {
#pragma unroll
  for (int d = 0; d < kWgDepth; ++d) {
    const int step = s0 + d;
    if (step >= nsteps) break;
    const int buf = d & 1;  // (kWgDepth is even: consecutive steps alternate buffers across groups too)
    store_step(buf, stage[d]);
    __syncthreads();
    load_step(stage[d], step + kWgDepth);  // its registers are free again: the rows kWgDepth steps ahead
    // transposing reads: lane (grp, li) with q_=li>>2, p_=li&3 addresses row 8*grp+4h+q_,
    // columns 16*blk+4p_ .. +3 and receives column li of rows 8*grp+4h .. +3.
    const int q_ = li >> 2, p_ = li & 3;
    const uint16_t* xs = &lds[buf][0];
    const uint16_t* ys = &lds[buf][32 * LDX];
    bf16x8 bfrag[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int nb = wn + j * WN;
      bf16x8 f = zero_bf16x8();
      if (nb < NB) {
        const uint16_t* a0 = ys + (8 * grp + q_) * LDY + nb * 16 + 4 * p_;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(a0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(a0 + 4 * LDY));
        f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      bfrag[j] = f;
    }
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
      const int mb = wm + i * WM;
      if (mb < MB) {
        const uint16_t* a0 = xs + (8 * grp + q_) * LDX + mb * 16 + 4 * p_;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(a0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)(a0 + 4 * LDX));
        const bf16x8 afrag = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
          if (wn + j * WN < NB)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  }

  // D[m = cin][n = cout]: lane holds rows 4*grp + {0..3}, column li of each 16x16 tile
  float* slab = slabs + (int64_t)item * CIN * COUT;
#pragma unroll
  for (int i = 0; i < MBW; ++i) {
    const int mb = wm + i * WM;
    if (mb >= MB) continue;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int nb = wn + j * WN;
      if (nb >= NB) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        slab[(int64_t)(mb * 16 + grp * 4 + r) * COUT + nb * 16 + li] = acc[i][j][r];
    }
  }
  }  // next item
}

template <int CIN, int COUT>
__global__ void __launch_bounds__(kWgThreads)
wgrad_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
             const int32_t* __restrict__ pairs, const int32_t* __restrict__ num, int kvol,
             int64_t cap, float* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * (32 * (CIN + 8) + 32 * (COUT + 8))];
  __shared__ int32_t sidx[2 * kWgSteps * 32];
  wgrad_body<CIN, COUT>(x, dy, pairs, num, kvol, cap, slabs, (int)blockIdx.x, (int)gridDim.x, lds, sidx);
}

// The weight gradients of TWO small layers in one launch (ococc_sparse_conv_wgrad_pair_bf16): the 16 -> 32 and 32 -> 64
// layers of the encoder are a few hundred latency-bound work items each, 10 + 13 us one after the other on a chip they
// fill to a fifth; queued to the end of the backward pass (spconv/ops.py) they run side by side.  (All three shapes in
// one launch was tried in round 3: the 64 x 128 body pushes the joint register allocation to 210.)
struct WgradJob {
  const uint16_t* x;
  const uint16_t* dy;
  const int32_t* pairs;
  const int32_t* num;
  float* slabs;
  int64_t cap;
  int32_t kvol, shape, first_block, blocks;
};
constexpr int kWgradMulti = 3;
struct WgradPairPack {
  WgradJob job[kWgradMulti];
  int32_t count;
};
template <bool BIG>
__global__ void __launch_bounds__(kWgThreads) wgrad_pair_kernel(WgradPairPack pk) {
  constexpr int CI = BIG ? 64 : 32, CO = BIG ? 128 : 64;   // the largest body of the launch sizes the shared LDS
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * (32 * (CI + 8) + 32 * (CO + 8))];
  __shared__ int32_t sidx[2 * kWgSteps * 32];
  int j = 0;
  while (j + 1 < pk.count && (int)blockIdx.x >= pk.job[j + 1].first_block) ++j;
  const WgradJob& w = pk.job[j];
  const int local = (int)blockIdx.x - w.first_block;
  if (w.shape == 0) wgrad_body<16, 32>(w.x, w.dy, w.pairs, w.num, w.kvol, w.cap, w.slabs, local, w.blocks, lds, sidx);
  else if (w.shape == 1 || !BIG) wgrad_body<32, 64>(w.x, w.dy, w.pairs, w.num, w.kvol, w.cap, w.slabs, local, w.blocks, lds, sidx);
  else wgrad_body<64, 128>(w.x, w.dy, w.pairs, w.num, w.kvol, w.cap, w.slabs, local, w.blocks, lds, sidx);
}

// dw[k][i] = sum of the slabs of offset k, fixed order: 16 elements x 16 slab lanes per
// block; lane p adds slabs p, p+16, ... into eight interleaved sums; fixed trees join the sums and the lanes.
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ slabs, const int32_t* __restrict__ num,
                                                   int k, int chunk, int64_t elems, float* __restrict__ dw) {
  __shared__ float red[4][16];
  int base, nslabs;
  wg_span(num, k, &base, &nslabs);
  const int e = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int64_t i = (int64_t)chunk * 16 + e;
  // sixteen running sums per lane: the centre offset has hundreds of slabs, and one sum per lane made its
  // blocks a chain of dependent L2 round trips (11-16 us whatever the layer size); with sixteen the ~250 slabs of
  // the benchmark's centre offset are ONE round of independent loads per lane
  constexpr int U = 16;
  float a[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = 0.f;
  if (i < elems) {
    const float* src = slabs + (int64_t)base * elems + i;
    int j = part;
    for (; j + 16 * (U - 1) < nslabs; j += 16 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] += src[(int64_t)(j + 16 * u) * elems];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j + 16 * u < nslabs) a[u] += src[(int64_t)(j + 16 * u) * elems];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] += a[u + 8];
  float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if ((threadIdx.x & 63) < 16) red[threadIdx.x >> 6][e] = s;
  __syncthreads();
  if (threadIdx.x < 16 && i < elems)
    dw[(int64_t)k * elems + i] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slabs, const int32_t* __restrict__ num, int kvol,
                    int64_t elems, float* __restrict__ dw) {
  wgrad_reduce_block(slabs, num, blockIdx.y, blockIdx.x, elems, dw);
}

// the slab reductions of several layers in one launch (the reductions of a backward pass feed nothing before
// the optimizer: spconv/ops.py queues them to the end of the pass)
constexpr int kMaxReduce = 8;
struct ReducePack {
  const float* slabs[kMaxReduce];
  const int32_t* num[kMaxReduce];
  float* dw[kMaxReduce];
  int64_t elems[kMaxReduce];
  int32_t kvol[kMaxReduce], first_block[kMaxReduce + 1], count;
};
// Persistent: a fixed grid walks the work units of every layer.  An offset with MANY slabs (the centre: hundreds) is
// cut into units of 16 elements x 16 slab lanes (wgrad_reduce_block); an offset with at most 16 slabs (every other
// offset of a sparse grid) into units of 256 elements, one thread each -- 27 x fewer, fatter units than one block per
// 16 elements of every offset (18 k blocks of which 26/27 summed 8 values: 23 us for three layers).
// Round 6: the units of ALL layers form one flat list (layer after layer) that the grid walks once.  Before, every
// workgroup went through the layers one after the other -- up to two units of the widest layer, then one of the next, ...:
// four dependent unit lifetimes (each a chain of scans, two rounds of loads and a store) for ~1 800 units on 1 024
// workgroups, 17.1 us in the configs[1] step.  With one list and 2 048 workgroups (all resident) a workgroup has ONE unit.
constexpr int kReduceGrid = 2048;
__device__ __forceinline__ void wgrad_reduce_multi_body(const ReducePack& pk, int first_unit, int unit_stride) {
  // per layer, lane = offset: units of the offset (inclusive scan + own count), its slab count, its first slab
  __shared__ int s_inc[kMaxReduce][64], s_cnt[kMaxReduce][64], s_slabs[kMaxReduce][64], s_base[kMaxReduce][64];
  __shared__ int s_total[kMaxReduce];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = wave; t < pk.count; t += 4) {
    const int kvol = pk.kvol[t];
    const int64_t elems = pk.elems[t];
    const int deep_units = (int)((elems + 15) / 16), wide_units = (int)((elems + 255) / 256);
    if (kvol <= 64) {
      const int my_slabs = lane < kvol ? wg_items(pk.num[t][lane]) : 0;
      int sinc = my_slabs;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(sinc, d, 64);
        if (lane >= d) sinc += v;
      }
      const int cnt = lane < kvol ? (my_slabs > 16 ? deep_units : wide_units) : 0;
      int inc = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(inc, d, 64);
        if (lane >= d) inc += v;
      }
      s_inc[t][lane] = inc;
      s_cnt[t][lane] = cnt;
      s_slabs[t][lane] = my_slabs;
      s_base[t][lane] = sinc - my_slabs;
      if (lane == 63) s_total[t] = inc;
    } else if (lane == 0) {   // (larger kernels: deep units for every offset, located by division)
      s_total[t] = deep_units * kvol;
    }
  }
  __syncthreads();
  int first[kMaxReduce + 1];
  first[0] = 0;
#pragma unroll
  for (int t = 0; t < kMaxReduce; ++t) first[t + 1] = first[t] + (t < pk.count ? s_total[t] : 0);
  bool deep_before = false;
  for (int unit = first_unit; unit < first[kMaxReduce]; unit += unit_stride) {
    int t = 0;
#pragma unroll
    for (int q = 1; q < kMaxReduce; ++q) t += (unit >= first[q]) ? 1 : 0;
    int lu = unit;
#pragma unroll
    for (int q = 1; q < kMaxReduce; ++q) lu -= (unit >= first[q]) ? (first[q] - first[q - 1]) : 0;
    const int kvol = pk.kvol[t];
    const int64_t elems = pk.elems[t];
    int k, local, nsl, sbase = 0;
    if (kvol <= 64) {
      const int inc = s_inc[t][lane], cnt = s_cnt[t][lane];
      const unsigned long long hit = __ballot(cnt > 0 && lu >= inc - cnt && lu < inc);
      const int src = __ffsll((long long)hit) - 1;
      k = src;
      local = lu - (s_inc[t][src] - s_cnt[t][src]);
      nsl = s_slabs[t][src];
      sbase = s_base[t][src];
    } else {
      const int deep_units = (int)((elems + 15) / 16);
      k = lu / deep_units;
      local = lu % deep_units;
      nsl = 17;  // force the deep path
    }
    if (kvol <= 64 && nsl <= 16) {
      const int64_t i = (int64_t)local * 256 + threadIdx.x;
      if (i < elems) {
        const float* src = pk.slabs[t] + (int64_t)sbase * elems + i;
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (u < nsl) a[u] = src[(int64_t)u * elems];
          if (u + 8 < nsl) a[u] += src[(int64_t)(u + 8) * elems];
        }
        pk.dw[t][(int64_t)k * elems + i] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      }
    } else {
      if (deep_before) __syncthreads();  // red[] of a previous deep unit
      deep_before = true;
      wgrad_reduce_block(pk.slabs[t], pk.num[t], k, local, elems, pk.dw[t]);
    }
  }
}

__global__ void __launch_bounds__(256) wgrad_reduce_multi_kernel(ReducePack pk) {
  wgrad_reduce_multi_body(pk, (int)blockIdx.x, (int)gridDim.x);
}
// both kinds of end-of-backward sums in one launch (the two launches were 15 + 8 us of mostly latency, one after the
// other).  The FIRST ln_blocks workgroups sum the LayerNorm partial rows, the wgrad_grid workgroups behind them walk the
// slab units (round 6: the other way round the few LayerNorm workgroups -- an 8 us chain of strided loads -- were
// dispatched only when the first of the 2 048 slab workgroups, which fill every slot of the chip, had left: 16.0 us for
// 10.6 + 8.3 us of separate launches).
__global__ void __launch_bounds__(256) backward_reduce_multi_kernel(ReducePack pk, LnReducePack lp, int ln_count,
                                                                    int ln_blocks, int wgrad_grid) {
  if ((int)blockIdx.x < ln_blocks) {
    ln_param_reduce_multi_body(lp, ln_count, (int)blockIdx.x);
    return;
  }
  wgrad_reduce_multi_body(pk, (int)blockIdx.x - ln_blocks, wgrad_grid);
}

inline int64_t wgrad_max_groups(int kvol, int64_t cap) {
  // sum_k ceil(ceil(num[k]/32)/kWgSteps) <= kvol*cap/(32*kWgSteps) + kvol (+ slack)
  return ococc_cdiv((int64_t)kvol * ococc_cdiv(cap, 32), kWgSteps) + kvol;
}

template <int CIN>
int dispatch_wgrad_cout(const uint16_t* x, const uint16_t* dy, int cout, const int32_t* pairs,
                        const int32_t* num, int kvol, int64_t cap, float* slabs, hipStream_t stream) {
  const int64_t maxg = wgrad_max_groups(kvol, cap);
  dim3 grid((unsigned)(maxg < kWgGrid ? maxg : kWgGrid)), block(kWgThreads);
  switch (cout) {
    case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<CIN, 16>), grid, block, 0, stream, x, dy, pairs, num, kvol, cap, slabs); break;
    case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<CIN, 32>), grid, block, 0, stream, x, dy, pairs, num, kvol, cap, slabs); break;
    case 64: hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<CIN, 64>), grid, block, 0, stream, x, dy, pairs, num, kvol, cap, slabs); break;
    case 128: hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<CIN, 128>), grid, block, 0, stream, x, dy, pairs, num, kvol, cap, slabs); break;
    default: return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "cout must be 16/32/64/128");
  }
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

}  // namespace

extern "C" int ococc_sparse_conv_gather_gemm_bf16(const uint16_t* feat, int64_t n_in, int32_t kd,
                                                  const uint16_t* wn, int32_t kvol, int32_t ncols,
                                                  const int32_t* table, const uint32_t* blockmask,
                                                  int64_t n_out, const float* bias, void* out,
                                                  int32_t out_dtype, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n_in >= 0 && n_out >= 0, "negative row count");
  OCOCC_REQUIRE(kvol >= 1 && kvol <= 32, "kernel volume must be 1..32");
  OCOCC_REQUIRE(ncols >= 16 && ncols % 16 == 0, "ncols must be a multiple of 16");
  OCOCC_REQUIRE(out_dtype == OCOCC_BF16 || out_dtype == OCOCC_F32, "out_dtype must be f32/bf16");
  if (n_out == 0) return OCOCC_OK;
  OCOCC_REQUIRE(wn && table && out, "null pointer");
  OCOCC_REQUIRE(feat || n_in == 0, "null feat");
  switch (kd) {
    case 16: return dispatch_cs<16>(feat, n_in, wn, kvol, ncols, table, blockmask, n_out, bias, out, out_dtype, stream);
    case 32: return dispatch_cs<32>(feat, n_in, wn, kvol, ncols, table, blockmask, n_out, bias, out, out_dtype, stream);
    case 64: return dispatch_cs<64>(feat, n_in, wn, kvol, ncols, table, blockmask, n_out, bias, out, out_dtype, stream);
    case 128: return dispatch_cs<128>(feat, n_in, wn, kvol, ncols, table, blockmask, n_out, bias, out, out_dtype, stream);
    default: return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "kd must be 16/32/64/128");
  }
}

extern "C" int ococc_sparse_conv_gather_gemm_ln_bf16(const uint16_t* feat, int64_t n_in, int32_t kd,
                                                     const uint16_t* wn, int32_t kvol, int32_t ncols,
                                                     const int32_t* table, const uint32_t* blockmask,
                                                     int64_t n_out, const float* gamma, const float* beta,
                                                     float eps, int32_t act, uint16_t* conv_out, uint16_t* y,
                                                     float* mean_rstd, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n_in >= 0 && n_out >= 0, "negative row count");
  OCOCC_REQUIRE(kvol >= 1 && kvol <= 32, "kernel volume must be 1..32");
  OCOCC_REQUIRE(ncols >= 16 && ncols % 16 == 0, "ncols must be a multiple of 16");
  OCOCC_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (gelu)");
  if (n_out == 0) return OCOCC_OK;
  OCOCC_REQUIRE(wn && table && conv_out && y && mean_rstd && gamma && beta, "null pointer");
  OCOCC_REQUIRE(feat || n_in == 0, "null feat");
  const LnArgs ln{gamma, beta, eps, (int)act, y, mean_rstd};
  switch (kd) {
    case 16: return dispatch_cs<16>(feat, n_in, wn, kvol, ncols, table, blockmask, n_out, nullptr, conv_out, OCOCC_BF16, stream, &ln);
    case 32: return dispatch_cs<32>(feat, n_in, wn, kvol, ncols, table, blockmask, n_out, nullptr, conv_out, OCOCC_BF16, stream, &ln);
    case 64: return dispatch_cs<64>(feat, n_in, wn, kvol, ncols, table, blockmask, n_out, nullptr, conv_out, OCOCC_BF16, stream, &ln);
    case 128: return dispatch_cs<128>(feat, n_in, wn, kvol, ncols, table, blockmask, n_out, nullptr, conv_out, OCOCC_BF16, stream, &ln);
    default: return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "kd must be 16/32/64/128");
  }
}

extern "C" int ococc_weight_prepare_bf16(const void* w, int32_t w_dtype, int32_t kvol, int32_t cin,
                                         int32_t cout, int32_t mode, uint16_t* wn,
                                         ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(kvol >= 1 && cin >= 1 && cout >= 1, "bad sizes");
  OCOCC_REQUIRE(mode >= 0 && mode < 8 && (mode & 3) <= 2, "mode must be 0/1/2, +4 for the fragment-major order");
  OCOCC_REQUIRE(!(mode & 4) || (cin % 32 == 0 && cout % 32 == 0), "fragment-major order needs channels in multiples of 32");
  OCOCC_REQUIRE(w_dtype == OCOCC_F32 || w_dtype == OCOCC_BF16, "w_dtype must be f32/bf16");
  OCOCC_REQUIRE(w && wn, "null pointer");
  const int64_t total = (int64_t)kvol * cin * cout;
  const int grid = ococc_grid_1d(total, 256);
  if (w_dtype == OCOCC_F32)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(weight_prepare_kernel<float>), dim3(grid), dim3(256), 0,
                       stream, (const float*)w, (int)kvol, (int)cin, (int)cout, (int)mode, wn);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(weight_prepare_kernel<uint16_t>), dim3(grid), dim3(256), 0,
                       stream, (const uint16_t*)w, (int)kvol, (int)cin, (int)cout, (int)mode, wn);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_weight_prepare_multi_bf16(int32_t count, const void* const* w, const int32_t* kvol,
                                               const int32_t* cin, const int32_t* cout, const int32_t* mode,
                                               void* const* wn, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(count >= 0 && count <= kMaxPrep, "at most 16 weight tensors per call");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(w && kvol && cin && cout && mode && wn, "null pointer table");
  PrepPack pk;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    OCOCC_REQUIRE(w[i] && wn[i] && kvol[i] >= 1 && cin[i] >= 1 && cout[i] >= 1 && mode[i] >= 0 && (mode[i] & 3) <= 2 && mode[i] < 8 &&
                      (!(mode[i] & 4) || (cin[i] % 32 == 0 && cout[i] % 32 == 0)),
                  "bad weight descriptor");
    pk.w[i] = (const float*)w[i];
    pk.wn[i] = (uint16_t*)wn[i];
    pk.kvol[i] = kvol[i];
    pk.cin[i] = cin[i];
    pk.cout[i] = cout[i];
    pk.mode[i] = mode[i];
    pk.first_block[i] = blocks;
    const int64_t total = (int64_t)kvol[i] * cin[i] * cout[i];
    OCOCC_REQUIRE(total < 0x7fffffffLL, "weight tensor too large");
    blocks += (int)(ococc_cdiv(total, 256) < 256 ? ococc_cdiv(total, 256) : 256);
  }
  pk.first_block[count] = blocks;
  pk.count = count;
  hipLaunchKernelGGL(weight_prepare_multi_kernel, dim3(blocks), dim3(256), 0, stream, pk);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int64_t ococc_sparse_conv_wgrad_workspace_bytes(int32_t kvol, int64_t pair_capacity,
                                                           int32_t cin, int32_t cout) {
  if (kvol < 1 || cin < 1 || cout < 1 || pair_capacity < 0) return -1;
  return wgrad_max_groups(kvol, pair_capacity) * cin * cout * (int64_t)sizeof(float);
}

extern "C" int ococc_sparse_conv_wgrad_bf16(const uint16_t* x, int64_t n_in, int32_t cin,
                                            const uint16_t* dy, int64_t n_out, int32_t cout,
                                            const int32_t* indice_pairs, const int32_t* indice_num,
                                            int32_t kvol, int64_t pair_capacity, float* dw,
                                            void* workspace, int64_t workspace_bytes,
                                            ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n_in >= 0 && n_out >= 0 && pair_capacity >= 0, "negative size");
  OCOCC_REQUIRE(kvol >= 1, "kvol < 1");
  const int64_t need = ococc_sparse_conv_wgrad_workspace_bytes(kvol, pair_capacity, cin, cout);
  OCOCC_REQUIRE(workspace && workspace_bytes >= need, "workspace too small");
  const int64_t elems = (int64_t)cin * cout;
  if (pair_capacity == 0 || n_in == 0 || n_out == 0) {
    OCOCC_REQUIRE(dw, "an empty problem has no slabs to reduce later: dw required");
    OCOCC_HIP(hipMemsetAsync(dw, 0, (int64_t)kvol * elems * sizeof(float), stream));
    return OCOCC_OK;
  }
  OCOCC_REQUIRE(x && dy && indice_pairs && indice_num, "null pointer");
  int rc;
  float* slabs = (float*)workspace;
  switch (cin) {
    case 16: rc = dispatch_wgrad_cout<16>(x, dy, cout, indice_pairs, indice_num, kvol, pair_capacity, slabs, stream); break;
    case 32: rc = dispatch_wgrad_cout<32>(x, dy, cout, indice_pairs, indice_num, kvol, pair_capacity, slabs, stream); break;
    case 64: rc = dispatch_wgrad_cout<64>(x, dy, cout, indice_pairs, indice_num, kvol, pair_capacity, slabs, stream); break;
    case 128: rc = dispatch_wgrad_cout<128>(x, dy, cout, indice_pairs, indice_num, kvol, pair_capacity, slabs, stream); break;
    default: return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "cin must be 16/32/64/128");
  }
  if (rc != OCOCC_OK) return rc;
  if (!dw) return OCOCC_OK;  // slabs only: the caller reduces later (ococc_sparse_conv_wgrad_reduce_multi)
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ococc_cdiv(elems, 16), kvol), dim3(256), 0,
                     stream, slabs, indice_num, (int)kvol, elems, dw);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_backward_param_reduce_multi(int32_t wcount, const void* const* workspaces,
                                                const int32_t* const* indice_nums, const int32_t* kvols,
                                                const int64_t* elems, float* const* dws, int32_t lcount,
                                                const void* const* partials, const int32_t* rows, const int32_t* c,
                                                void* const* dgamma, void* const* dbeta, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(wcount >= 1 && wcount <= kMaxReduce, "1..8 weight-gradient reductions per call");
  OCOCC_REQUIRE(lcount >= 1 && lcount <= kLnMultiMax, "1..16 LayerNorm reductions per call");
  OCOCC_REQUIRE(workspaces && indice_nums && kvols && elems && dws && partials && rows && c && dgamma && dbeta,
                "null pointer table");
  ReducePack pk;
  int blocks = 0;
  for (int i = 0; i < wcount; ++i) {
    OCOCC_REQUIRE(workspaces[i] && indice_nums[i] && dws[i] && kvols[i] >= 1 && elems[i] >= 1, "bad reduction descriptor");
    pk.slabs[i] = (const float*)workspaces[i];
    pk.num[i] = indice_nums[i];
    pk.dw[i] = dws[i];
    pk.elems[i] = elems[i];
    pk.kvol[i] = kvols[i];
    pk.first_block[i] = blocks;
    blocks += (int)ococc_cdiv(elems[i], 16) * kvols[i];
  }
  pk.first_block[wcount] = blocks;
  pk.count = wcount;
  const int wgrad_grid = blocks < kReduceGrid ? blocks : kReduceGrid;
  LnReducePack lp;
  int lblocks = 0;
  for (int j = 0; j < lcount; ++j) {
    OCOCC_REQUIRE(partials[j] && rows[j] >= 1 && c[j] >= 1, "bad layer");
    lp.partials[j] = (const float*)partials[j];
    lp.dgamma[j] = (float*)dgamma[j];
    lp.dbeta[j] = (float*)dbeta[j];
    lp.rows[j] = rows[j];
    lp.c[j] = c[j];
    lp.first[j] = lblocks;
    lblocks += (2 * c[j] + 7) / 8;
  }
  lp.first[lcount] = lblocks;
  hipLaunchKernelGGL(backward_reduce_multi_kernel, dim3(wgrad_grid + lblocks), dim3(256), 0, stream, pk, lp, (int)lcount,
                     lblocks, wgrad_grid);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_sparse_conv_wgrad_multi_bf16(int32_t count, const uint16_t* const* x, const uint16_t* const* dy,
                                                  const int32_t* cin, const int32_t* cout,
                                                  const int32_t* const* indice_pairs, const int32_t* const* indice_num,
                                                  const int32_t* kvol, const int64_t* pair_capacity,
                                                  void* const* workspaces, const int64_t* workspace_bytes,
                                                  ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(count >= 2 && count <= kWgradMulti, "2 or 3 layers per call");
  OCOCC_REQUIRE(x && dy && cin && cout && indice_pairs && indice_num && kvol && pair_capacity && workspaces && workspace_bytes,
                "null pointer table");
  WgradPairPack pk;
  int blocks = 0;
  bool big = false;
  for (int j = 0; j < count; ++j) {
    const int shape = (cin[j] == 16 && cout[j] == 32) ? 0 : (cin[j] == 32 && cout[j] == 64) ? 1
                      : (cin[j] == 64 && cout[j] == 128) ? 2 : -1;
    if (shape < 0) return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "shapes served together: 16x32, 32x64, 64x128");
    big = big || shape == 2;
    OCOCC_REQUIRE(kvol[j] >= 1 && pair_capacity[j] >= 1, "bad sizes");
    OCOCC_REQUIRE(x[j] && dy[j] && indice_pairs[j] && indice_num[j] && workspaces[j], "null pointer");
    OCOCC_REQUIRE(workspace_bytes[j] >= ococc_sparse_conv_wgrad_workspace_bytes(kvol[j], pair_capacity[j], cin[j], cout[j]),
                  "workspace too small");
    const int64_t maxg = wgrad_max_groups(kvol[j], pair_capacity[j]);
    WgradJob& w = pk.job[j];
    w.x = x[j]; w.dy = dy[j]; w.pairs = indice_pairs[j]; w.num = indice_num[j]; w.slabs = (float*)workspaces[j];
    w.cap = pair_capacity[j]; w.kvol = kvol[j]; w.shape = shape; w.first_block = blocks;
    // (a workgroup without an item still costs its slot ~2 us to find that out: the benchmark's layers have ~460 items each)
    w.blocks = (int)(maxg < kWgGrid / 4 ? maxg : kWgGrid / 4);
    blocks += w.blocks;
  }
  pk.count = count;
  if (big) hipLaunchKernelGGL(wgrad_pair_kernel<true>, dim3(blocks), dim3(kWgThreads), 0, stream, pk);
  else hipLaunchKernelGGL(wgrad_pair_kernel<false>, dim3(blocks), dim3(kWgThreads), 0, stream, pk);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_sparse_conv_wgrad_reduce_multi(int32_t count, const void* const* workspaces,
                                                    const int32_t* const* indice_nums, const int32_t* kvols,
                                                    const int64_t* elems, float* const* dws, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(count >= 0 && count <= kMaxReduce, "at most 8 reductions per call");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(workspaces && indice_nums && kvols && elems && dws, "null pointer table");
  ReducePack pk;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    OCOCC_REQUIRE(workspaces[i] && indice_nums[i] && dws[i] && kvols[i] >= 1 && elems[i] >= 1, "bad reduction descriptor");
    pk.slabs[i] = (const float*)workspaces[i];
    pk.num[i] = indice_nums[i];
    pk.dw[i] = dws[i];
    pk.elems[i] = elems[i];
    pk.kvol[i] = kvols[i];
    pk.first_block[i] = blocks;
    blocks += (int)ococc_cdiv(elems[i], 16) * kvols[i];
  }
  pk.first_block[count] = blocks;
  pk.count = count;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks < kReduceGrid ? blocks : kReduceGrid), dim3(256), 0, stream, pk);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
