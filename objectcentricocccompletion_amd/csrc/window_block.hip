// B7, fused: one SST encoder layer (mmdet3d/models/sst/sst_basic_block_v2.py:41-75 WindowAttention.forward,
// :105-127 EncoderLayer.forward, post-norm) as two tile kernels per direction instead of ~50 library / ATen launches:
//
//   window_attn_block:  y1 = LN1(x + out_proj(MHA(q = k = x + pos, v = x)))        (sst_basic_block_v2.py:58-71,113-115)
//   token_ffn_block:    y2 = LN2(y1 + linear2(act(linear1(y1))))                   (sst_basic_block_v2.py:116-118)
//
// Shape of the work on MI355X.  A window of configs[4] holds ~10 tokens (max 30 / 60 per drop level), so per-window
// workgroups leave the matrix cores 90 % padding.  Here a workgroup owns a TILE of 64 token slots that holds several
// whole windows (ococc_window_tile_plan packs them greedily); the projections and the FFN are dense 64-token GEMMs and
// attention is block diagonal inside the tile (a slot attends the slots of its own window, given as a [lo, hi) slot span).
// All GEMMs run transposed, out^T = W x^T: the weight is the MFMA A operand (16 output channels x 32 k), the tokens are
// the B operand (16 tokens) read from a row-major bf16 LDS tile with one ds_read_b128 per lane, and a lane ends with 4
// consecutive channels of one token -> 8-byte LDS stores, no transposes.  Weights never touch LDS: they are
// pre-arranged in MFMA-fragment order (ococc_linear_fragments_bf16: one wave load = 1 KB of consecutive bytes) and
// stream from L2 into registers, each fragment reused for the 4 token blocks of the tile.  A wave owns a slice of the
// output channels of every GEMM, so a weight matrix is read once per tile.
// Rounding points (mirrored by oracle/sst_ref.py): x, pos, x + pos, q, k, v, P, attention output, y1, act(h), y2 and
// every gradient that leaves a kernel are bf16; all sums (MFMA accumulators, softmax, LayerNorm, residuals) are f32.
// Backward: nothing but x is saved by the forward; each backward kernel recomputes its block from its input
// (flash-style), then runs the chain rule inside the tile, and writes the operands of the weight gradients
// (dqkv, attention output, dz1 | act(h), dh, dz2) for ococc_token_wgrad_bf16.
// No atomics anywhere: results are bit-reproducible.
#include "common.hpp"
#include "ln_math.hpp"

namespace {

constexpr int TM = 64;           // token slots per tile
constexpr int E = 128;           // d_model
constexpr int NH = 8, HD = 16;   // heads, head dim
constexpr int FF = 256;          // feed-forward width
constexpr int LDX = E + 8;       // LDS row strides (elements): an ODD number of 16-byte pieces, so that the ds_read_b128 of 16 consecutive rows (a GEMM operand) start in 16 different bank quads (+32 B, rounds 2-4: rows r and r + 8 shared theirs -- SQ_LDS_BANK_CONFLICT 33 % of the LDS cycles)
constexpr int LDQ = 3 * E + 8;
constexpr int LDH = FF + 8;
constexpr int kThreads = 256;

// Workgroup barrier of the block kernels: everything the waves exchange goes through LDS, so the barrier waits for this
// wave's LDS operations only.  __syncthreads() also waits for every global load in flight (s_waitcnt vmcnt(0)) -- i.e. the
// first barrier behind a prefetch of the next tile's rows waited for that prefetch: 2-5 us per tile (stamps,
// tools/probe/sst_bwd_stamps.py).  Global results reach their consumers in registers (the compiler's own waits).
#ifndef OCOCC_WB_FULL_BARRIERS
#define TILE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define TILE_BARRIER() __syncthreads()
#endif

#ifdef OCOCC_WB_STAMPS
// diagnostic build only (tools/probe/sst_bwd_stamps.py): wall-clock stamps (100 MHz) per workgroup and phase of its SECOND tile
__device__ long long* wb_stamps = nullptr;
#define WSTAMP(it, slot) do { if ((it) == 1 && threadIdx.x == 0 && wb_stamps) wb_stamps[(int64_t)blockIdx.x * 16 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define WSTAMP_F(slot) do { if (NW == 8 && threadIdx.x == 0 && wb_stamps) wb_stamps[(int64_t)blockIdx.x * 16 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WSTAMP(it, slot) do { } while (0)
#define WSTAMP_F(slot) do { } while (0)
#endif

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ bf16x8 tr_pair(const uint16_t* lo_rows, const uint16_t* hi_rows) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)lo_rows);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)hi_rows);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 pack_tiles(const f32x4 a, const f32x4 b) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[j] = (__bf16)a[j];
    r[4 + j] = (__bf16)b[j];
  }
  return r;
}
__device__ __forceinline__ u32x2 pack4(const f32x4 o) {
  return u32x2{ococc_pack_bf16x2(o[0], o[1]), ococc_pack_bf16x2(o[2], o[3])};
}
__device__ __forceinline__ f32x4 unpack4(const u32x2 v) {
  return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
               __uint_as_float(v.y & 0xffff0000u)};
}
__device__ __forceinline__ s16x4 ld4(const uint16_t* p) { return *(const s16x4*)p; }


// ---------------------------------------------------------------------------------------------------------------
// GEMM pieces.  out^T[n][m] += sum_k W[n][k] X[m][k] for NB blocks of 16 output channels and the 4 token blocks of the
// tile.  The weight fragments of a wave's channel slice go L2 -> registers (load_frags), requested one GEMM ahead of
// their use; xs: LDS tile [64][ldb] bf16.
template <int NB, int KS, bool STREAM = true>
__device__ __forceinline__ void load_frags(const int tid_, const uint16_t* __restrict__ wf, int nb0, bf16x8 (&a)[NB][KS]) {
  // STREAM: the pointer passes through an empty asm -- the loads are loop invariant, and hoisted out of the persistent
  // tile loop all fragments of a kernel would have to live in registers at once (which is what the backward kernels,
  // one workgroup per CU with 512 registers per lane, do on purpose with STREAM = false)
  if (STREAM) asm volatile("" : "+s"(wf));
  const bf16x8* wp = (const bf16x8*)wf + (size_t)nb0 * KS * 64 + (tid_ & 63);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[nb][ks] = wp[(nb * KS + ks) * 64];
}
// FRESH: acc is write-only (the first k-step multiplies into a zero literal: no register zeroing in front of the GEMM)
template <int NB, int KS, bool FRESH = false>
__device__ __forceinline__ void tile_gemm(const int tid_, const bf16x8 (&a)[NB][KS], const uint16_t* xs, int ldb, f32x4 (&acc)[NB][4]) {
  const int lane = tid_ & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 b[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) b[mb] = *(const bf16x8*)(xs + (mb * 16 + c) * ldb + ks * 32 + 8 * g);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nb][ks], b[mb], (FRESH && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nb][mb], 0, 0, 0);
  }
}

template <int NB>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NB][4]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Sum over the 128 channels of each of the lane's 4 tokens (token mb*16 + c): the lane's own values, its three
// partner lanes (same c, other g) and the other waves through `red` ([NW waves][64 tokens] floats).  One barrier.
template <int NW>
__device__ __forceinline__ void token_sums(const int tid_, float (&part)[4], float* red) {
  const int lane = tid_ & 63, wave = tid_ >> 6, c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    float p = part[mb];
    p += __shfl_xor(p, 16, 64);
    p += __shfl_xor(p, 32, 64);
    if (g == 0) red[wave * TM + mb * 16 + c] = p;
  }
  TILE_BARRIER();
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int t = mb * 16 + c;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * TM + t];
    part[mb] = s;
  }
}

// LayerNorm statistics of z (the wave's NB channel blocks x 4 token blocks; a token's 128 channels are spread over the
// waves): two passes (mean, then centred squares).  z <- xhat = (z - mean) * rstd; rstd returned per token block.
template <int NW, int NB>
__device__ __forceinline__ void tile_layernorm(const int tid_, f32x4 (&z)[NB][4], float eps, float* red0, float* red1, float (&rstd)[4]) {
  float s[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    s[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[mb] += z[nb][mb][r];
  }
  token_sums<NW>(tid_, s, red0);
  float q[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const float mean = s[mb] * (1.f / E);
    q[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        z[nb][mb][r] -= mean;
        q[mb] += z[nb][mb][r] * z[nb][mb][r];
      }
  }
  token_sums<NW>(tid_, q, red1);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    rstd[mb] = rsqrtf(q[mb] * (1.f / E) + eps);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[nb][mb][r] *= rstd[mb];
  }
}

// LayerNorm backward on the wave's slice: xh = xhat, dy = gradient of the LN output -> dy <- gradient of the LN
// input; adds the tile's terms to dgam / dbet (per lane: channels 16(nb0+nb)+4g+r, summed over the lane's tokens).
template <int NW, int NB>
__device__ __forceinline__ void tile_layernorm_bwd(const int tid_, const f32x4 (&xh)[NB][4], f32x4 (&dy)[NB][4], const f32x4 (&gam)[NB],
                                                   const float (&rstd)[4], float* red0, float* red1,
                                                   f32x4 (&dgam)[NB], f32x4 (&dbet)[NB]) {
  float s1[4], s2[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    s1[mb] = s2[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = dy[nb][mb][r];
        dgam[nb][r] += d * xh[nb][mb][r];
        dbet[nb][r] += d;
        const float dg = d * gam[nb][r];
        dy[nb][mb][r] = dg;
        s1[mb] += dg;
        s2[mb] += dg * xh[nb][mb][r];
      }
  }
  token_sums<NW>(tid_, s1, red0);
  token_sums<NW>(tid_, s2, red1);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const float m1 = s1[mb] * (1.f / E), m2 = s2[mb] * (1.f / E);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dy[nb][mb][r] = ((dy[nb][mb][r] - m1) - xh[nb][mb][r] * m2) * rstd[mb];
  }
}

// per-channel sums of the wave's slice over all the tokens the workgroup has seen -> its row of partial sums
// ([dgamma(128) | dbeta(128)] floats)
template <int NB>
__device__ __forceinline__ void store_param_partials(const int tid_, const f32x4 (&dgam)[NB], const f32x4 (&dbet)[NB], int nb0,
                                                     float* __restrict__ dst) {
  const int lane = tid_ & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = dgam[nb][r], b = dbet[nb][r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        a += __shfl_xor(a, m, 64);
        b += __shfl_xor(b, m, 64);
      }
      if (c == 0) {
        dst[16 * (nb0 + nb) + 4 * g + r] = a;
        dst[E + 16 * (nb0 + nb) + 4 * g + r] = b;
      }
    }
}

// rows of a [64][ld] bf16 LDS tile -> rows of a global [*, WIDTH] bf16 tensor, 16 B per lane, whole rows coalesced
template <int NW, int WIDTH>
__device__ __forceinline__ void tile_store_rows(const uint16_t* ts, int ld, uint16_t* __restrict__ dst, const int* rows_s,
                                                int64_t row0, int64_t nrows) {
  constexpr int PP = WIDTH / 8;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // (keeps the per-piece address arithmetic out of the registers between calls)
#pragma unroll
  for (int i0 = 0; i0 < TM * PP; i0 += NW * 64) {
    const int i = i0 + tid, s = i / PP, p = i % PP;
    const int64_t r = rows_s ? (int64_t)rows_s[s] : (row0 + s < nrows ? row0 + s : -1);
    if (r >= 0) *(u32x4*)(dst + r * WIDTH + p * 8) = *(const u32x4*)(ts + s * ld + p * 8);
  }
}

// the thread's 16-byte pieces of a [64][128] bf16 tile: registers <- global rows (zeros for missing rows), -> LDS
template <int NW>
struct TilePieces {
  static constexpr int P = TM * (E / 8) / (NW * 64);   // 4 (256 threads) or 2 (512)
  u32x4 v[P];
  __device__ __forceinline__ void fetch(const uint16_t* __restrict__ src, const int* rows_s, int64_t row0, int64_t nrows) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int i = tid + j * NW * 64, s = i >> 4, p = i & 15;
      const int64_t r = rows_s ? (int64_t)rows_s[s] : (row0 + s < nrows ? row0 + s : -1);
      v[j] = u32x4{0u, 0u, 0u, 0u};
      if (r >= 0 && src) v[j] = *(const u32x4*)(src + r * E + p * 8);
    }
  }
  __device__ __forceinline__ void stash(uint16_t* ts, int ld) const {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const int i = tid + j * NW * 64, s = i >> 4, p = i & 15;
      *(u32x4*)(ts + s * ld + p * 8) = v[j];
    }
  }
};

__device__ __forceinline__ uint32_t add_bf16x2(uint32_t a, uint32_t b) {
  const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
  const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
  return ococc_pack_bf16x2(lo, hi);
}

template <int ACT>  // 0 gelu (erf), 1 relu
__device__ __forceinline__ f32x4 act_fwd4(const f32x4 h) {
  if (ACT == 0) {
    const ln_f32x2 a = ln_gelu2(ln_f32x2{h[0], h[1]}), b = ln_gelu2(ln_f32x2{h[2], h[3]});
    return f32x4{a.x, a.y, b.x, b.y};
  }
  return f32x4{fmaxf(h[0], 0.f), fmaxf(h[1], 0.f), fmaxf(h[2], 0.f), fmaxf(h[3], 0.f)};
}
template <int ACT>
__device__ __forceinline__ f32x4 act_grad4(const f32x4 h) {
  if (ACT == 0) {
    const ln_f32x2 a = ln_gelu_grad2(ln_f32x2{h[0], h[1]}), b = ln_gelu_grad2(ln_f32x2{h[2], h[3]});
    return f32x4{a.x, a.y, b.x, b.y};
  }
  return f32x4{h[0] > 0.f ? 1.f : 0.f, h[1] > 0.f ? 1.f : 0.f, h[2] > 0.f ? 1.f : 0.f, h[3] > 0.f ? 1.f : 0.f};
}

// ---------------------------------------------------------------------------------------------------------------
struct TileMeta {
  int rows[TM];   // flat token row of each slot, -1 = empty
  int span[TM];   // lo | hi << 8: the slots of the slot's window; 0 for an empty slot
};
__device__ __forceinline__ void load_meta(const int tid_, TileMeta* tm, const int32_t* tile_rows, const int32_t* tile_span, int64_t tile) {
  if (tid_ < TM) {
    tm->rows[tid_] = tile_rows[tile * TM + tid_];
    tm->span[tid_] = tile_span[tile * TM + tid_];
  }
}

// Front of the attention block (forward, and the recompute of the backward), given the x / pos pieces of the tile in
// registers and the V fragments of the in-projection already requested (fv):
//   xs <- x;  V = Wv x + bv;  xs <- bf16(x + pos);  Q | K = Wqk (x + pos) + bqk   -> qs [64][Q | K | V].
// Weight fragments are always requested one phase ahead of the GEMM that uses them, so that their L2 latency runs
// under the previous phase: Wqk during the V GEMM, and the caller's `after_qk` (the out-projection's) during Q | K.
// `first` runs behind the first barrier (the caller loads the next tile's meta there), `between` after the pos pieces
// have been consumed (the caller fetches the next tile's rows there: its meta is visible by then).
template <int NW, bool RES, typename F0, typename F, typename F2>
__device__ __forceinline__ void attn_front(const int tid_, const uint16_t* __restrict__ wqkv, const bf16x8 (&fv)[8 / NW][4],
                                           const bf16x8 (&fqk_res)[16 / NW][4], const float* __restrict__ bqkv,
                                           TilePieces<NW>& xv, TilePieces<NW>& pv, bool has_pos, uint16_t* xs,
                                           uint16_t* qs, F0 first, F between, F2 after_qk) {
  const int lane = tid_ & 63, wave = tid_ >> 6, c = lane & 15, g = lane >> 4;
  constexpr int NQK = 16 / NW, NV = 8 / NW;
  WSTAMP_F(11);
  xv.stash(xs, LDX);
  TILE_BARRIER();
  WSTAMP_F(12);
  first();   // every thread has left the previous tile behind
  bf16x8 fqk[NQK][4];
  {
    f32x4 acc[NV][4];
    tile_gemm<NV, 4, true>(tid_, fv, xs, LDX, acc);
    if (!RES) load_frags<NQK, 4>(tid_, wqkv, NQK * wave, fqk);
#pragma unroll
    for (int nb = 0; nb < NV; ++nb) {
      const int n = 16 * (16 + NV * wave + nb) + 4 * g;
      const f32x4 b = *(const f32x4*)(bqkv + n);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) *(u32x2*)(qs + (mb * 16 + c) * LDQ + n) = pack4(acc[nb][mb] + b);
    }
  }
  TILE_BARRIER();
  WSTAMP_F(13);
  if (has_pos) {
#pragma unroll
    for (int j = 0; j < TilePieces<NW>::P; ++j) {   // xs <- bf16(x + pos), every thread its own pieces
      xv.v[j].x = add_bf16x2(xv.v[j].x, pv.v[j].x);
      xv.v[j].y = add_bf16x2(xv.v[j].y, pv.v[j].y);
      xv.v[j].z = add_bf16x2(xv.v[j].z, pv.v[j].z);
      xv.v[j].w = add_bf16x2(xv.v[j].w, pv.v[j].w);
    }
    xv.stash(xs, LDX);
  }
  between();
  TILE_BARRIER();
  WSTAMP_F(14);
  {
    f32x4 acc[NQK][4];
    if (RES) tile_gemm<NQK, 4, true>(tid_, fqk_res, xs, LDX, acc);
    else tile_gemm<NQK, 4, true>(tid_, fqk, xs, LDX, acc);
    after_qk();
#pragma unroll
    for (int nb = 0; nb < NQK; ++nb) {
      const int n = 16 * (NQK * wave + nb) + 4 * g;
      const f32x4 b = *(const f32x4*)(bqkv + n);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) *(u32x2*)(qs + (mb * 16 + c) * LDQ + n) = pack4(acc[nb][mb] + b);
    }
  }
  TILE_BARRIER();
}

// key-tile range [lo, hi] that the 16 queries of query tile qt can see (union of their windows); hi < lo: none
__device__ __forceinline__ void tile_range(const int tid_, const TileMeta* tm, int qt, int& klo, int& khi) {
  const int c = tid_ & 15;
  const int sp = tm->span[qt * 16 + c];
  int lo = sp & 255, hi = sp >> 8;
  int a = hi > lo ? (lo >> 4) : 99, b = hi > lo ? ((hi - 1) >> 4) : -1;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    a = min(a, __shfl_xor(a, m, 64));
    b = max(b, __shfl_xor(b, m, 64));
  }
  klo = __builtin_amdgcn_readfirstlane(a);   // (every lane holds the same pair: scalar branches on the tile range)
  khi = __builtin_amdgcn_readfirstlane(b);
}

// Attention of head h for the 16 queries of tile qt (lane: query c): S^T = K Q^T (16x16x16 MFMA, keys on the rows),
// softmax over the keys of the query's window in registers, O^T = V^T P^T (16x16x32 MFMA, V^T by transposing reads).
// Returns o (d = 4g + r of query c) and the log-sum-exp of the query.
// madd: the lane's additive key mask (0 for a key of the query's window, -inf otherwise; slot [kt][r] = key 16 kt + 4 g + r),
// the same for every head: the caller works it out once per query tile.  Scores are kept in base 2 (the 1 / sqrt(16)
// scale times log2 e goes into one fused multiply-add with the mask), so that an exponential is v_sub + v_exp.
__device__ __forceinline__ f32x4 attn_head_fwd(const int tid_, const uint16_t* qs, int h, int qt, int klo, int khi,
                                               const f32x4 (&madd)[4], float& lse) {
  const int lane = tid_ & 63, c = lane & 15, g = lane >> 4;
  const int q_ = c >> 2, p_ = c & 3;
  const uint16_t* qh = qs + h * HD;
  const uint16_t* kh = qs + E + h * HD;
  const uint16_t* vh = qs + 2 * E + h * HD;
  constexpr float kScale2 = 0.25f * 1.44269504088896340736f;   // 1 / sqrt(16) * log2(e)
  const s16x4 bq = ld4(qh + (qt * 16 + c) * LDQ + 4 * g);
  f32x4 s[4];
  float m = -1e30f;   // (finite: a query without a key -- an empty slot -- ends with weights 0, not NaN)
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (kt >= klo && kt <= khi) {
      const s16x4 ak = ld4(kh + (kt * 16 + c) * LDQ + 4 * g);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ak, bq, acc, 0, 0, 0);   // rows: keys 4g+r, col: query c
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kt][r] = fmaf(acc[r], kScale2, madd[kt][r]);
      m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
    }
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt >= klo && kt <= khi) {   // (a window of ~10 tokens touches one or two of the four key tiles)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[kt][r] - m);   // (-inf - m -> 0)
        s[kt][r] = e;
        sum += e;
      }
    } else {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  lse = sum > 0.f ? (m + __log2f(sum)) * 0.69314718055994530942f : 0.f;   // natural units, as the backward reads it
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (2 * u + 1 >= klo && 2 * u <= khi) {
      const bf16x8 pb = pack_tiles(s[2 * u] * inv, s[2 * u + 1] * inv);
      const uint16_t* a0 = vh + (32 * u + 4 * g + q_) * LDQ + 4 * p_;
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(a0, a0 + 16 * LDQ), pb, o, 0, 0, 0);   // rows: d, col: query
    }
  }
  return o;
}

// all heads of the tile: wave w runs heads [w * 8 / NW, (w + 1) * 8 / NW); o -> os (row stride LDX), lse -> lse_s or null
template <int NW>
__device__ __forceinline__ void attn_tile_fwd(const int tid_, const TileMeta* tm, const uint16_t* qs, uint16_t* os, float* lse_s) {
  const int lane = tid_ & 63, wave = tid_ >> 6, c = lane & 15, g = lane >> 4;
  for (int qt = 0; qt < 4; ++qt) {
    int klo, khi;
    tile_range(tid_, tm, qt, klo, khi);
    const int sp = tm->span[qt * 16 + c];
    const int lo = sp & 255, hi = sp >> 8;
    f32x4 madd[4];   // 0 / -inf per score slot of the lane: keys of the query's window only
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        madd[kt][r] = (key >= lo && key < hi) ? 0.f : -INFINITY;
      }
#pragma unroll
    for (int hh = 0; hh < NH / NW; ++hh) {
      const int h = (NH / NW) * wave + hh;
      float lse;
      const f32x4 o = attn_head_fwd(tid_, qs, h, qt, klo, khi, madd, lse);
      *(u32x2*)(os + (qt * 16 + c) * LDX + h * HD + 4 * g) = pack4(o);
      if (lse_s && g == 0) lse_s[h * TM + qt * 16 + c] = lse;
    }
  }
}

template <int NW_>
constexpr int attn_lds() {
  return (TM * LDX + TM * LDQ) * 2 + 2 * NW_ * TM * 4 + 2 * (int)sizeof(TileMeta) + NH * TM * 4;
}

// ---------------------------------------------------------------------------------------------------------------
// All four block kernels: persistent workgroups of 4 waves, two per CU, looping over tiles.  The x (/ pos) rows of the
// next tile are fetched into registers while this one computes; weight fragments are requested one GEMM ahead.
constexpr int NW = 4;

// Forward of the attention block.
__global__ void __launch_bounds__(256, 2)
window_attn_block_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ pos,
                             const int32_t* __restrict__ tile_rows, const int32_t* __restrict__ tile_span,
                             int64_t num_tiles, const uint16_t* __restrict__ wqkv, const float* __restrict__ bqkv,
                             const uint16_t* __restrict__ wo, const float* __restrict__ bo,
                             const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                             uint16_t* __restrict__ y, uint16_t* __restrict__ o_save, float* __restrict__ lse_save) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* xs = (uint16_t*)smem;                 // x | x + pos | attention output
  uint16_t* qs = xs + TM * LDX;                   // Q | K | V, then y staging
  float* red0 = (float*)(qs + TM * LDQ);
  float* red1 = red0 + NW * TM;
  TileMeta* tms = (TileMeta*)(red1 + NW * TM);    // [2]: this tile's and the next one's
  float* lse_s = (float*)(tms + 2);               // [8 heads][64] (training: the backward reads it back)
  const int tid_ = threadIdx.x, lane = tid_ & 63, wave = tid_ >> 6, g = lane >> 4;
  (void)g;
  bf16x8 fv[2][4], fo[2][4], fqk_none[4][4] = {};   // (fqk_none: unused, the forward streams Wqk inside attn_front)
  load_frags<2, 4>(tid_, wqkv, 16 + 2 * wave, fv);
  int64_t tile = blockIdx.x;
  load_meta(tid_, &tms[0], tile_rows, tile_span, tile);
  TILE_BARRIER();
  TilePieces<NW> xv, pv;
  xv.fetch(x, tms[0].rows, 0, 0);
  pv.fetch(pos, tms[0].rows, 0, 0);
#pragma unroll 1
  for (int it = 0; tile < num_tiles; tile += gridDim.x, ++it) {
    // (thread coordinates re-derived per tile from an opaque copy of threadIdx.x: hoisted out of the loop, the address
    // arithmetic that hangs on them would sit in ~100 registers for the whole kernel)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6, c = lane & 15, g = lane >> 4;
    const TileMeta* tm = &tms[it & 1];
    const int64_t next = tile + gridDim.x;
    // the residual (x at the lanes' output positions) is asked for now and used after the attention
    u32x2 res[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int r = tm->rows[mb * 16 + c];
        res[nb][mb] = u32x2{0u, 0u};
        if (r >= 0) res[nb][mb] = *(const u32x2*)(x + (int64_t)r * E + 16 * (2 * wave + nb) + 4 * g);
      }
    attn_front<NW, false>(tid_, 
        wqkv, fv, fqk_none, bqkv, xv, pv, pos != nullptr, xs, qs,
        [&]() {
          if (next < num_tiles) load_meta(tid_, &tms[(it & 1) ^ 1], tile_rows, tile_span, next);
        },
        [&]() {},
        [&]() {
          load_frags<2, 4>(tid_, wo, 2 * wave, fo);
          // the next tile's rows are asked for HERE, behind the last wait in front of the attention (LDS and matrix work
          // only, ~8 us): the vector-memory counter retires in order, and asked for in front of the Q | K GEMM (rounds 3-4)
          // they were waited for by its weight fragments (the next tile's meta: written behind the first barrier)
          if (next < num_tiles) {
            xv.fetch(x, tms[(it & 1) ^ 1].rows, 0, 0);
            pv.fetch(pos, tms[(it & 1) ^ 1].rows, 0, 0);
          }
        });
    attn_tile_fwd<NW>(tid_, tm, qs, xs, o_save ? lse_s : nullptr);   // o goes where x + pos was (its last reader was the Q | K GEMM)
    TILE_BARRIER();
    if (o_save) {
      // training: the attention output and the softmax's log-sum-exp of every (token, head) leave for the backward kernel,
      // which then starts behind the attention instead of running it again (288 B per token against 28 % of its time)
      tile_store_rows<NW, E>(xs, LDX, o_save, tm->rows, 0, 0);
      for (int i = tid_; i < TM * 2; i += NW * 64) {
        const int s_ = i >> 1, hq = (i & 1) * 4;
        const int r = tm->rows[s_];
        if (r >= 0)
          *(f32x4*)(lse_save + (int64_t)r * NH + hq) = f32x4{lse_s[hq * TM + s_], lse_s[(hq + 1) * TM + s_],
                                                           lse_s[(hq + 2) * TM + s_], lse_s[(hq + 3) * TM + s_]};
      }
    }
    f32x4 z[2][4];
    tile_gemm<2, 4, true>(tid_, fo, xs, LDX, z);
    load_frags<2, 4>(tid_, wqkv, 16 + 2 * wave, fv);    // the next tile's first GEMM
    f32x4 gam[2], bet[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = 16 * (2 * wave + nb) + 4 * g;
      const f32x4 b = *(const f32x4*)(bo + n);
      gam[nb] = *(const f32x4*)(ln_w + n);
      bet[nb] = *(const f32x4*)(ln_b + n);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) z[nb][mb] = z[nb][mb] + b + unpack4(res[nb][mb]);
    }
    float rstd[4];
    tile_layernorm<NW, 2>(tid_, z, eps, red0, red1, rstd);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {   // y staged over Q | K | V (last read by the attention, two barriers ago)
      const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) *(u32x2*)(qs + (mb * 16 + c) * LDQ + n) = pack4(z[nb][mb] * gam[nb] + bet[nb]);
    }
    TILE_BARRIER();
    tile_store_rows<NW, E>(qs, LDQ, y, tm->rows, 0, 0);
    // (the next iteration writes xs before its first barrier: xs was last read by the out-projection GEMM, before the
    // LayerNorm barriers; it writes qs and this tile's meta slot only behind that barrier)
  }
}

// ---------------------------------------------------------------------------------------------------------------
template <int NW_>
constexpr int ffn_lds() {
  return (TM * LDX + TM * LDH) * 2 + 2 * NW_ * TM * 4;
}

template <int ACT>
__global__ void __launch_bounds__(256, 2)
token_ffn_block_fwd_kernel(const uint16_t* __restrict__ x, int64_t num_tokens, const uint16_t* __restrict__ w1,
                           const float* __restrict__ b1, const uint16_t* __restrict__ w2, const float* __restrict__ b2,
                           const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                           uint16_t* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* xs = (uint16_t*)smem;
  uint16_t* hs = xs + TM * LDX;
  float* red0 = (float*)(hs + TM * LDH);
  float* red1 = red0 + NW * TM;
  const int tid_ = threadIdx.x, lane = tid_ & 63, wave = tid_ >> 6, g = lane >> 4;
  (void)g;
  bf16x8 f1[4][4], f2[2][8];
  load_frags<4, 4>(tid_, w1, 4 * wave, f1);
  const int64_t tiles = (num_tokens + TM - 1) / TM;
  int64_t tile = blockIdx.x;
  TilePieces<NW> xv;
  xv.fetch(x, nullptr, tile * TM, num_tokens);
#pragma unroll 1
  for (; tile < tiles; tile += gridDim.x) {
    // (thread coordinates re-derived per tile from an opaque copy of threadIdx.x: hoisted out of the loop, the address
    // arithmetic that hangs on them would sit in ~100 registers for the whole kernel)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6, c = lane & 15, g = lane >> 4;
    const int64_t row0 = tile * TM;
    xv.stash(xs, LDX);
    if (tile + gridDim.x < tiles) xv.fetch(x, nullptr, (tile + gridDim.x) * TM, num_tokens);
    TILE_BARRIER();
    {
      f32x4 acc[4][4];
      tile_gemm<4, 4, true>(tid_, f1, xs, LDX, acc);
      load_frags<2, 8>(tid_, w2, 2 * wave, f2);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int n = 16 * (4 * wave + nb) + 4 * g;
        const f32x4 b = *(const f32x4*)(b1 + n);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) *(u32x2*)(hs + (mb * 16 + c) * LDH + n) = pack4(act_fwd4<ACT>(acc[nb][mb] + b));
      }
    }
    TILE_BARRIER();
    f32x4 z[2][4];
    tile_gemm<2, 8, true>(tid_, f2, hs, LDH, z);
    load_frags<4, 4>(tid_, w1, 4 * wave, f1);           // the next tile's first GEMM
    f32x4 gam[2], bet[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = 16 * (2 * wave + nb) + 4 * g;
      const f32x4 b = *(const f32x4*)(b2 + n);
      gam[nb] = *(const f32x4*)(ln_w + n);
      bet[nb] = *(const f32x4*)(ln_b + n);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        z[nb][mb] = z[nb][mb] + b + unpack4(*(const u32x2*)(xs + (mb * 16 + c) * LDX + n));
    }
    float rstd[4];
    tile_layernorm<NW, 2>(tid_, z, eps, red0, red1, rstd);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) *(u32x2*)(hs + (mb * 16 + c) * LDH + n) = pack4(z[nb][mb] * gam[nb] + bet[nb]);
    }
    TILE_BARRIER();
    tile_store_rows<NW, E>(hs, LDH, y, nullptr, row0, num_tokens);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The two backward kernels carry more live values per lane than the forward ones (act'(h), the LayerNorm gradient, the
// three gradients of a head): with 4 waves per tile they spilled ~200 registers.  They run 8 waves per tile, one
// workgroup per CU -- every wave owns half as many channel blocks (and one head) -- and request a GEMM's fragments
// when it starts rather than one GEMM ahead.
//
// Backward of the FFN block.  In: x (= y1), dy (gradient of y2).  Out: dx (gradient of y1, residual path included),
// and the weight-gradient operands a = act(h) [*,256], dh [*,256], dz [*,128] (gradient at the LN input), plus one
// row of LN parameter-gradient partial sums per workgroup.  act'(h) stays (as bf16) in the registers of the lanes
// that computed h: the gradient of act(h) arrives in the same lanes, because the two GEMMs have the same shape.
template <int ACT>
__global__ void __launch_bounds__(512, 2)
token_ffn_block_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, int64_t num_tokens,
                           const uint16_t* __restrict__ w1, const float* __restrict__ b1,
                           const uint16_t* __restrict__ w2, const float* __restrict__ b2,
                           const float* __restrict__ ln_w, float eps, const uint16_t* __restrict__ w2t,
                           const uint16_t* __restrict__ w1t, uint16_t* __restrict__ dx, uint16_t* __restrict__ a_out,
                           uint16_t* __restrict__ dh_out, uint16_t* __restrict__ dz_out,
                           float* __restrict__ ln_partial) {
  constexpr int NW = 8;                // (shadows the file-wide 4: see the comment on the backward kernels above)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* xs = (uint16_t*)smem;      // y1 | dz2 | dx staging
  uint16_t* hs = xs + TM * LDX;        // act(h) | dh
  float* red0 = (float*)(hs + TM * LDH);
  float* red1 = red0 + NW * TM;
  const int tid_ = threadIdx.x, lane = tid_ & 63, wave = tid_ >> 6, g = lane >> 4;
  // (fragments are requested when a GEMM starts, not one GEMM ahead: see above)
  const f32x4 gam[1] = {*(const f32x4*)(ln_w + 16 * wave + 4 * g)};
  f32x4 dgam[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, dbet[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
  const int64_t tiles = (num_tokens + TM - 1) / TM;
  int64_t tile = blockIdx.x;
  TilePieces<NW> xv;
  xv.fetch(x, nullptr, tile * TM, num_tokens);
  // dy at the lanes' output positions, one tile ahead like x (rows past the end read the last row: their gradient is zeroed
  // where it is used) -- straight-line loads, see window_attn_block_bwd_kernel
  u32x2 dy_n[4];
  auto side = [&](int64_t t0) {
    int t_ = threadIdx.x;
    asm volatile("" : "+v"(t_));
    const int c_ = t_ & 15, n_ = 16 * (t_ >> 6) + 4 * ((t_ & 63) >> 4);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int64_t r = t0 + mb * 16 + c_;
      dy_n[mb] = *(const u32x2*)(dy + (r < num_tokens ? r : num_tokens - 1) * E + n_);
    }
  };
  side(tile * TM);
#pragma unroll 1
  for (; tile < tiles; tile += gridDim.x) {
    // (thread coordinates re-derived per tile from an opaque copy of threadIdx.x: hoisted out of the loop, the address
    // arithmetic that hangs on them would sit in ~100 registers for the whole kernel)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6, c = lane & 15, g = lane >> 4;
    const int64_t row0 = tile * TM;
    xv.stash(xs, LDX);
    TILE_BARRIER();
    const int n1 = 16 * wave + 4 * g;   // the lane's 4 channels of the 128-wide tensors
    u32x2 gact[2][4];
    {
      bf16x8 f[2][4];
      load_frags<2, 4>(tid_, w1, 2 * wave, f);
      f32x4 hpre[2][4];
      tile_gemm<2, 4, true>(tid_, f, xs, LDX, hpre);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int n = 16 * (2 * wave + nb) + 4 * g;
        const f32x4 b = *(const f32x4*)(b1 + n);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const f32x4 hp = hpre[nb][mb] + b;
          gact[nb][mb] = pack4(act_grad4<ACT>(hp));
          *(u32x2*)(hs + (mb * 16 + c) * LDH + n) = pack4(act_fwd4<ACT>(hp));
        }
      }
    }
    TILE_BARRIER();
    tile_store_rows<NW, FF>(hs, LDH, a_out, nullptr, row0, num_tokens);
    f32x4 z[1][4], dz[1][4];
    {
      bf16x8 f[1][8];
      load_frags<1, 8>(tid_, w2, wave, f);
      tile_gemm<1, 8, true>(tid_, f, hs, LDH, z);
    }
    {
      const f32x4 b = *(const f32x4*)(b2 + n1);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        z[0][mb] = z[0][mb] + b + unpack4(*(const u32x2*)(xs + (mb * 16 + c) * LDX + n1));
        const int64_t r = row0 + mb * 16 + c;
        dz[0][mb] = r < num_tokens ? unpack4(dy_n[mb]) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // the next tile's rows are asked for HERE, in front of the LayerNorm (LDS and arithmetic only, ~3 us): the vector-memory
    // counter retires in order, and asked for at the top of the tile (rounds 3-4) they were waited for by the first GEMM's
    // weight fragments
    if (tile + gridDim.x < tiles) {
      xv.fetch(x, nullptr, (tile + gridDim.x) * TM, num_tokens);
      side((tile + gridDim.x) * TM);
    }
    float rstd[4];
    tile_layernorm<NW, 1>(tid_, z, eps, red0, red1, rstd);                       // z = xhat2
    tile_layernorm_bwd<NW, 1>(tid_, z, dz, gam, rstd, red0, red1, dgam, dbet);   // dz = gradient at the LN input (f32)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)   // (every wave is past its reads of y1: the barriers of the LN sums)
      *(u32x2*)(xs + (mb * 16 + c) * LDX + n1) = pack4(dz[0][mb]);
    TILE_BARRIER();
    tile_store_rows<NW, E>(xs, LDX, dz_out, nullptr, row0, num_tokens);
    {
      bf16x8 f[2][4];
      load_frags<2, 4>(tid_, w2t, 2 * wave, f);
      f32x4 da[2][4];
      tile_gemm<2, 4, true>(tid_, f, xs, LDX, da);           // d act(h) = W2^T dz
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)           // over act(h): its last reader was the W2 GEMM
          *(u32x2*)(hs + (mb * 16 + c) * LDH + n) = pack4(da[nb][mb] * unpack4(gact[nb][mb]));
      }
    }
    TILE_BARRIER();
    tile_store_rows<NW, FF>(hs, LDH, dh_out, nullptr, row0, num_tokens);
    f32x4 gx[1][4];
    {
      bf16x8 f[1][8];
      load_frags<1, 8>(tid_, w1t, wave, f);
      tile_gemm<1, 8, true>(tid_, f, hs, LDH, gx);           // W1^T dh
    }
    // (the dz tile in xs -- B operand of the d act GEMM, source of dz_out -- was last read before the barrier above)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) *(u32x2*)(xs + (mb * 16 + c) * LDX + n1) = pack4(gx[0][mb] + dz[0][mb]);
    TILE_BARRIER();
    tile_store_rows<NW, E>(xs, LDX, dx, nullptr, row0, num_tokens);
    TILE_BARRIER();   // the next tile's y1 goes into xs
  }
  store_param_partials<1>(tid_, dgam, dbet, wave, ln_partial + (int64_t)blockIdx.x * 2 * E);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the attention block.  In: x, pos, dy (gradient of y1).  Out: dx, and for the weight gradients dqkv
// [*,384], o (attention output) [*,128], dz (gradient at the LN input) [*,128]; LN partial sums per workgroup.
__global__ void __launch_bounds__(512, 2)
window_attn_block_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ pos,
                             const uint16_t* __restrict__ dy, const int32_t* __restrict__ tile_rows,
                             const int32_t* __restrict__ tile_span, int64_t num_tiles,
                             const uint16_t* __restrict__ wqkv, const float* __restrict__ bqkv,
                             const uint16_t* __restrict__ wo, const float* __restrict__ bo,
                             const float* __restrict__ ln_w, float eps, const uint16_t* __restrict__ wot,
                             const uint16_t* __restrict__ wqkvt, uint16_t* __restrict__ dx,
                             uint16_t* __restrict__ dqkv_out, uint16_t* __restrict__ dz_out,
                             uint16_t* __restrict__ o_out, float* __restrict__ ln_partial,
                             const uint16_t* __restrict__ o_in, const float* __restrict__ lse_in) {
  constexpr int NW = 8;                           // one head per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* xs = (uint16_t*)smem;                 // x | x + pos | o | dz1 | dO | dx staging
  uint16_t* qs = xs + TM * LDX;                   // Q | K | V, then dQ | dK | dV in place
  float* red0 = (float*)(qs + TM * LDQ);
  float* red1 = red0 + NW * TM;
  TileMeta* tms = (TileMeta*)(red1 + NW * TM);
  float* lse_s = (float*)(tms + 2);               // [8 heads][64]
  const int tid_ = threadIdx.x, lane = tid_ & 63, wave = tid_ >> 6, g = lane >> 4;
  (void)g;
  bf16x8 fv[1][4], fqk_none[2][4] = {};
  load_frags<1, 4>(tid_, wqkv, 16 + wave, fv);
  const f32x4 gam[1] = {*(const f32x4*)(ln_w + 16 * wave + 4 * g)};
  f32x4 dgam[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, dbet[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
  int64_t tile = blockIdx.x;
  load_meta(tid_, &tms[0], tile_rows, tile_span, tile);
  TILE_BARRIER();
  TilePieces<NW> xv, pv;
  xv.fetch(x, tms[0].rows, 0, 0);
  pv.fetch(pos, tms[0].rows, 0, 0);
  TilePieces<NW> ov_n;
  float lse_n = 0.f;
  u32x2 res_n[4], dy_n[4];
  auto side = [&](const TileMeta* m) {   // the side inputs of the tile described by m (see the top of the loop)
    // Straight-line code: rows of empty slots read row 0 (their values are never used: dy is zeroed at the top of the
    // loop, nothing of such a row is stored).  With the loads behind `if (row >= 0)` the wait for the Q | K weight fragments
    // that follows became a wait for everything in flight (the counter is tracked per basic block): +3 us per tile.
    int t_ = threadIdx.x;
    asm volatile("" : "+v"(t_));
    const int c_ = t_ & 15, n_ = 16 * (t_ >> 6) + 4 * ((t_ & 63) >> 4);
    if (o_in) {
#pragma unroll
      for (int j = 0; j < TilePieces<NW>::P; ++j) {
        const int i = t_ + j * NW * 64, s_ = i >> 4, p = i & 15;
        const int r = m->rows[s_];
        ov_n.v[j] = *(const u32x4*)(o_in + (int64_t)(r < 0 ? 0 : r) * E + p * 8);
      }
      const int r = m->rows[t_ >> 3];
      lse_n = lse_in[(int64_t)(r < 0 ? 0 : r) * NH + (t_ & 7)];
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int r = m->rows[mb * 16 + c_];
      const int64_t rc = r < 0 ? 0 : r;
      res_n[mb] = *(const u32x2*)(x + rc * E + n_);
      dy_n[mb] = *(const u32x2*)(dy + rc * E + n_);
    }
  };
  side(&tms[0]);
#pragma unroll 1
  for (int it = 0; tile < num_tiles; tile += gridDim.x, ++it) {
    // (thread coordinates re-derived per tile from an opaque copy of threadIdx.x: hoisted out of the loop, the address
    // arithmetic that hangs on them would sit in ~100 registers for the whole kernel)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6, c = lane & 15, g = lane >> 4;
    const int q_ = c >> 2, p_ = c & 3;
    const TileMeta* tm = &tms[it & 1];
    const int64_t next = tile + gridDim.x;
    WSTAMP(it, 0);
    // the tile's side inputs -- the kept attention output and log-sum-exp rows (o_in), the residual and dy at the lanes'
    // output positions -- were asked for one tile ahead (`side` below): every wait of this kernel is counted in order,
    // so loads issued in front of a tile's first GEMM would be waited for there
    const int n1 = 16 * wave + 4 * g;   // the lane's 4 channels of the 128-wide tensors
    TilePieces<NW> ov = ov_n;
    const float lse_mine = lse_n;
    u32x2 res[4];
    f32x4 dz[1][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      res[mb] = res_n[mb];
      dz[0][mb] = tm->rows[mb * 16 + c] >= 0 ? unpack4(dy_n[mb]) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (tm->rows[mb * 16 + c] < 0) res[mb] = u32x2{0u, 0u};
    }
    attn_front<NW, false>(tid_, 
        wqkv, fv, fqk_none, bqkv, xv, pv, pos != nullptr, xs, qs,
        [&]() {
          if (next < num_tiles) load_meta(tid_, &tms[(it & 1) ^ 1], tile_rows, tile_span, next);
        },
        [&]() {},
        [&]() {});
    WSTAMP(it, 1);   // front: x + pos, Q | K | V
    if (o_in) {   // (o goes where x + pos was: its last reader was the Q | K GEMM, in front of attn_front's last barrier)
      ov.stash(xs, LDX);
      lse_s[(tid_ & 7) * TM + (tid_ >> 3)] = lse_mine;
    } else {
      attn_tile_fwd<NW>(tid_, tm, qs, xs, lse_s);
    }
    TILE_BARRIER();
    WSTAMP(it, 2);   // attention forward
    if (!o_in) tile_store_rows<NW, E>(xs, LDX, o_out, tm->rows, 0, 0);
    f32x4 z[1][4];
    {
      bf16x8 f[1][4];
      load_frags<1, 4>(tid_, wo, wave, f);
      tile_gemm<1, 4, true>(tid_, f, xs, LDX, z);
    }
    WSTAMP(it, 3);   // o stored, out-projection
    {
      const f32x4 b = *(const f32x4*)(bo + n1);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) z[0][mb] = z[0][mb] + b + unpack4(res[mb]);
    }
    float rstd[4];
    tile_layernorm<NW, 1>(tid_, z, eps, red0, red1, rstd);
    tile_layernorm_bwd<NW, 1>(tid_, z, dz, gam, rstd, red0, red1, dgam, dbet);   // dz = dz1, kept for the residual
    WSTAMP(it, 4);   // LayerNorm forward + backward
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)   // over o: every wave is past the out-projection GEMM and the o_out copy
      *(u32x2*)(xs + (mb * 16 + c) * LDX + n1) = pack4(dz[0][mb]);
    TILE_BARRIER();
    tile_store_rows<NW, E>(xs, LDX, dz_out, tm->rows, 0, 0);
    {
      f32x4 go[1][4];
      bf16x8 f[1][4];
      load_frags<1, 4>(tid_, wot, wave, f);
      tile_gemm<1, 4, true>(tid_, f, xs, LDX, go);    // dO = Wo^T dz1
      TILE_BARRIER();                     // dz1 tile: read by every wave's GEMM and by the copy above
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) *(u32x2*)(xs + (mb * 16 + c) * LDX + n1) = pack4(go[0][mb]);
    }
    TILE_BARRIER();
    WSTAMP(it, 5);   // dz1 stored, dO = Wo^T dz1
    // The next tile's rows (x, pos, the kept attention output and log-sum-exp, the residual and dy pieces) are asked for
    // HERE: the two attention passes below are LDS and matrix work only, ~7 us without a wait on the vector-memory
    // counter.  That counter retires in order: asked for inside the front (rounds 3-4), the prefetch was waited for by
    // the Q | K GEMM's weight fragments right behind it -- 3 us per tile (tools/probe/sst_bwd_stamps.py).
    if (next < num_tiles) {
      xv.fetch(x, tms[(it & 1) ^ 1].rows, 0, 0);
      pv.fetch(pos, tms[(it & 1) ^ 1].rows, 0, 0);
      side(&tms[(it & 1) ^ 1]);
    }
    // attention backward, one head per wave.  Pass 1 (lanes own queries): dQ and delta; pass 2 (lanes own
    // keys): dK, dV.  Both recompute the probabilities with 16x16x16 MFMAs; the gradients of a head replace its Q, K, V
    // in place once both passes have read them (only this wave touches the head's columns).
    {
      const int h = wave;
      const uint16_t* qh = qs + h * HD;
      const uint16_t* kh = qs + E + h * HD;
      const uint16_t* vh = qs + 2 * E + h * HD;
      const uint16_t* dh_ = xs + h * HD;            // dO of the head, row stride LDX
      float* lq = lse_s + h * TM;
      float* dl = red0 + wave * TM;                 // delta of the head's queries (red0 is free here)
      f32x4 gq[4], gk[4], gv[4];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        int klo, khi;
        tile_range(tid_, tm, qt, klo, khi);
        const int qi = qt * 16 + c;
        const int sp = tm->span[qi];
        const int lo = sp & 255, hi = sp >> 8;
        const s16x4 bq = ld4(qh + qi * LDQ + 4 * g);
        const s16x4 bdo = ld4(dh_ + qi * LDX + 4 * g);
        const float lse_q = lq[qi];
        f32x4 pT[4], dpT[4];
        float delta = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          pT[kt] = dpT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (kt >= klo && kt <= khi) {
            const s16x4 ak = ld4(kh + (kt * 16 + c) * LDQ + 4 * g);
            const s16x4 av = ld4(vh + (kt * 16 + c) * LDQ + 4 * g);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 sc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ak, bq, zero, 0, 0, 0);    // S^T[key][query]
            const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bdo, zero, 0, 0, 0);   // dP^T[key][query]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt * 16 + 4 * g + r;
              const float p = (key >= lo && key < hi) ? __expf(sc[r] * 0.25f - lse_q) : 0.f;
              pT[kt][r] = p;
              dpT[kt][r] = dp[r];
              delta += p * dp[r];
            }
          }
        }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        if (g == 0) dl[qi] = delta;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * u + 1 >= klo && 2 * u <= khi) {
            f32x4 d0, d1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              d0[r] = pT[2 * u][r] * (dpT[2 * u][r] - delta) * 0.25f;
              d1[r] = pT[2 * u + 1][r] * (dpT[2 * u + 1][r] - delta) * 0.25f;
            }
            const uint16_t* a0 = kh + (32 * u + 4 * g + q_) * LDQ + 4 * p_;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(a0, a0 + 16 * LDQ), pack_tiles(d0, d1), acc, 0, 0, 0);
          }
        }
        gq[qt] = acc;   // dQ^T: d = 4g + r of query c
      }
      WSTAMP(it, 6);   // attention backward pass 1 (dQ, delta)
      // (delta written by this wave's g == 0 lanes, read below by all its lanes: LDS ops of a wave complete in order)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        int qlo, qhi;
        tile_range(tid_, tm, kt, qlo, qhi);   // the windows are equivalence classes: the same range read as query tiles
        const int kj = kt * 16 + c;
        const int spk = tm->span[kj];
        const int klo_ = spk & 255, khi_ = spk >> 8;
        const s16x4 bk = ld4(kh + kj * LDQ + 4 * g);
        const s16x4 bv = ld4(vh + kj * LDQ + 4 * g);
        f32x4 dsv[4], pvv[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          dsv[qt] = pvv[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (qt >= qlo && qt <= qhi) {
            const s16x4 aq = ld4(qh + (qt * 16 + c) * LDQ + 4 * g);
            const s16x4 ado = ld4(dh_ + (qt * 16 + c) * LDX + 4 * g);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 sc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aq, bk, zero, 0, 0, 0);    // S[query][key]
            const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ado, bv, zero, 0, 0, 0);   // dP[query][key]
            const f32x4 l4 = *(const f32x4*)(lq + qt * 16 + 4 * g);
            const f32x4 d4 = *(const f32x4*)(dl + qt * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int qi = qt * 16 + 4 * g + r;   // a query sees this key iff both sit in the same (non-empty) window
              const float p = (qi >= klo_ && qi < khi_) ? __expf(sc[r] * 0.25f - l4[r]) : 0.f;
              pvv[qt][r] = p;
              dsv[qt][r] = p * (dp[r] - d4[r]) * 0.25f;
            }
          }
        }
        f32x4 acck = {0.f, 0.f, 0.f, 0.f}, accv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * u + 1 >= qlo && 2 * u <= qhi) {
            const uint16_t* aq0 = qh + (32 * u + 4 * g + q_) * LDQ + 4 * p_;
            const uint16_t* ad0 = dh_ + (32 * u + 4 * g + q_) * LDX + 4 * p_;
            acck = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(aq0, aq0 + 16 * LDQ),
                                                           pack_tiles(dsv[2 * u], dsv[2 * u + 1]), acck, 0, 0, 0);
            accv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(ad0, ad0 + 16 * LDX),
                                                           pack_tiles(pvv[2 * u], pvv[2 * u + 1]), accv, 0, 0, 0);
          }
        }
        gk[kt] = acck;
        gv[kt] = accv;
      }
      WSTAMP(it, 7);   // pass 2 (dK, dV)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        uint16_t* row = qs + (t * 16 + c) * LDQ + h * HD + 4 * g;
        *(u32x2*)(row) = pack4(gq[t]);
        *(u32x2*)(row + E) = pack4(gk[t]);
        *(u32x2*)(row + 2 * E) = pack4(gv[t]);
      }
    }
    TILE_BARRIER();
    tile_store_rows<NW, 3 * E>(qs, LDQ, dqkv_out, tm->rows, 0, 0);
    WSTAMP(it, 8);   // dqkv stored
    f32x4 gx[1][4];
    {
      bf16x8 f[1][12];
      load_frags<1, 12>(tid_, wqkvt, wave, f);
      tile_gemm<1, 12, true>(tid_, f, qs, LDQ, gx);   // dx = Wqkv^T dqkv (+ dz1: the residual)
    }
    WSTAMP(it, 9);   // dx GEMM
    load_frags<1, 4>(tid_, wqkv, 16 + wave, fv);   // the next tile's first GEMM
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)   // over dO: its last readers were the attention passes, before the barrier above
      *(u32x2*)(xs + (mb * 16 + c) * LDX + n1) = pack4(gx[0][mb] + dz[0][mb]);
    TILE_BARRIER();
    tile_store_rows<NW, E>(xs, LDX, dx, tm->rows, 0, 0);
    TILE_BARRIER();   // the next tile's x goes into xs, its Q | K | V into qs, its meta into this tile's slot
    WSTAMP(it, 10);
  }
  store_param_partials<1>(tid_, dgam, dbet, wave, ln_partial + (int64_t)blockIdx.x * 2 * E);
}

#ifdef OCOCC_WB_STAMPS
}  // namespace
extern "C" int ococc_wb_set_stamps(long long* dev_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(wb_stamps), &dev_buffer, sizeof(dev_buffer)) == hipSuccess ? 0 : -1;
}
namespace {
#endif
// ---------------------------------------------------------------------------------------------------------------
// f32 matrices (any strides) -> bf16 MFMA A-operand fragments: dst[rb][cs][lane = 16 g + r][j] = S[16 rb + r][32 cs + 8 g + j]
constexpr int kMaxFrag = 16;
struct FragPack {
  const float* src[kMaxFrag];
  uint16_t* dst[kMaxFrag];
  int32_t rows[kMaxFrag], cols[kMaxFrag];
  int64_t rs[kMaxFrag], cs[kMaxFrag];
  int32_t first_block[kMaxFrag + 1];
  int32_t count;
};
__global__ void __launch_bounds__(256) linear_fragments_kernel(FragPack pk) {
  int t = 0;
  while (t + 1 < pk.count && (int)blockIdx.x >= pk.first_block[t + 1]) ++t;
  const int cols = pk.cols[t], total = pk.rows[t] * cols, ksteps = cols >> 5;
  const int nblk = pk.first_block[t + 1] - pk.first_block[t];
  for (int i = ((int)blockIdx.x - pk.first_block[t]) * 256 + (int)threadIdx.x; i < total; i += nblk * 256) {
    const int j = i & 7, lane = (i >> 3) & 63, blk = i >> 9;   // destination index = ((rb * ksteps + cs) * 64 + lane) * 8 + j
    const int cs = blk % ksteps, rb = blk / ksteps;
    const int r = 16 * rb + (lane & 15), cidx = 32 * cs + 8 * (lane >> 4) + j;
    pk.dst[t][i] = ococc_f32_to_bf16(pk.src[t][r * pk.rs[t] + cidx * pk.cs[t]]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Tile plan: windows (their token counts) -> tiles of 64 slots, every window whole inside one tile.  Greedy first
// fit over the windows in their given order, run by one workgroup: each thread packs a chunk of >= 128 consecutive
// windows on its own (a tile never spans two chunks: ~1 half-empty tile per chunk), a block scan numbers the tiles.
constexpr int kPlanThreads = 1024;
constexpr int kPlanLdsWindows = 96 * 1024;   // window populations staged in LDS as bytes (a window has <= 64 tokens)
__global__ void __launch_bounds__(kPlanThreads)
window_tile_plan_kernel(const int32_t* __restrict__ win_len, int64_t num_windows, int32_t* __restrict__ win_tile,
                        int32_t* __restrict__ win_base, int32_t* __restrict__ num_tiles) {
  __shared__ int cnt[kPlanThreads];
  extern __shared__ uint8_t lens[];
  // one coalesced pass brings the populations on chip: the greedy loops below are chains of dependent reads
  const bool staged = num_windows <= kPlanLdsWindows;
  if (staged) {
    for (int64_t i = threadIdx.x; i < num_windows; i += kPlanThreads) {
      const int n = win_len[i];
      lens[i] = (uint8_t)(n < 0 ? 0 : (n > 255 ? 255 : n));
    }
    __syncthreads();
  }
  auto len_of = [&](int64_t w) -> int { return staged ? (int)lens[w] : win_len[w]; };
  const int64_t chunk = max((int64_t)128, (num_windows + kPlanThreads - 1) / kPlanThreads);
  const int64_t w0 = (int64_t)threadIdx.x * chunk, w1 = min(num_windows, w0 + chunk);
  int tiles = 0, fill = 0;
  for (int64_t w = w0; w < w1; ++w) {
    const int n = len_of(w);
    if (n <= 0) continue;
    if (fill + n > TM) {
      ++tiles;
      fill = 0;
    }
    fill += n;
  }
  if (fill > 0) ++tiles;
  cnt[threadIdx.x] = tiles;
  __syncthreads();
  for (int off = 1; off < kPlanThreads; off <<= 1) {   // inclusive scan
    const int v = threadIdx.x >= off ? cnt[threadIdx.x - off] : 0;
    __syncthreads();
    cnt[threadIdx.x] += v;
    __syncthreads();
  }
  int tile = cnt[threadIdx.x] - tiles;
  if (threadIdx.x == kPlanThreads - 1) *num_tiles = cnt[threadIdx.x];
  fill = 0;
  for (int64_t w = w0; w < w1; ++w) {
    const int n = len_of(w);
    if (n <= 0) {
      win_tile[w] = -1;
      win_base[w] = 0;
      continue;
    }
    if (fill + n > TM) {
      ++tile;
      fill = 0;
    }
    win_tile[w] = tile;
    win_base[w] = fill;
    fill += n;
  }
}

__global__ void __launch_bounds__(256)
window_tile_fill_kernel(const int32_t* __restrict__ win_len, const int64_t* __restrict__ win_off,
                        const int32_t* __restrict__ tok, int64_t num_windows, const int32_t* __restrict__ win_tile,
                        const int32_t* __restrict__ win_base, int32_t* __restrict__ tile_rows,
                        int32_t* __restrict__ tile_span) {
  // one 64-lane wave per window (a window holds at most 64 tokens)
  const int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int t = threadIdx.x & 63;
  if (w >= num_windows) return;
  const int n = win_len[w];
  if (t >= n) return;
  const int base = win_base[w];
  const int64_t slot = (int64_t)win_tile[w] * TM + base + t;
  tile_rows[slot] = tok[win_off[w] + t];
  tile_span[slot] = base | ((base + n) << 8);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients of the token linears: dW[n][k] = sum_t G[t][n] X[t][k] (+ db[n] = sum_t G[t][n]) for up to 8
// (G, X) pairs in one launch.  The contraction runs over tokens, so both MFMA operands need the token index innermost:
// 32-token pieces of G (a 64-column slice) and X (all K columns) are staged row-major in LDS and read with the
// transposing ds_read_b64_tr_b16.  A workgroup owns (slab of tokens, 64-row slice of dW) and leaves an f32 partial
// [64, K]; ococc_partial_rows_sum_f32 adds the slabs in a fixed order.  Workgroups of one slab sit on one XCD
// (blockIdx % 8) next to each other in dispatch order: the X rows they share are read from HBM once.
constexpr int kWgMax = 8;
constexpr int kWgSlice = 64;
struct WgradPack {
  const uint16_t* g[kWgMax];      // [tokens, ldg] bf16, the matrix's columns start at g
  const uint16_t* x[kWgMax];      // [tokens, K] bf16
  const uint16_t* xadd[kWgMax];   // optional second operand added to x (bf16 sum) for dW rows < add_rows
  float* dw[kWgMax];              // partials [slabs][N][K]
  float* db[kWgMax];              // partials [slabs][N]
  int32_t ldg[kWgMax], n[kWgMax], k[kWgMax], add_rows[kWgMax];
  int32_t first_slice[kWgMax + 1];
  int32_t count, slabs;
  int64_t tokens, chunk;          // tokens per slab (a multiple of 32)
};

template <int K>
__device__ __forceinline__ void wgrad_body(const WgradPack& pk, int t, int slice, int slab, char* smem) {
  constexpr int LDG = kWgSlice + 16, LDK = K + 16;
  constexpr int XP = K / 64;      // 16-byte pieces of X per thread and step
  uint16_t* gs = (uint16_t*)smem;                 // [2][32][LDG]
  uint16_t* xs = gs + 2 * 32 * LDG;               // [2][32][LDK]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int q_ = c >> 2, p_ = c & 3;
  const int n0 = slice * kWgSlice;
  const uint16_t* gp = pk.g[t] + n0;
  const uint16_t* xp = pk.x[t];
  const uint16_t* ap = (pk.xadd[t] && n0 < pk.add_rows[t]) ? pk.xadd[t] : nullptr;
  const int ldg = pk.ldg[t];
  const int64_t t0 = (int64_t)slab * pk.chunk, t1 = min(pk.tokens, t0 + pk.chunk);
  const int steps = t1 > t0 ? (int)((t1 - t0 + 31) >> 5) : 0;
  const int grow = threadIdx.x >> 3, gpc = threadIdx.x & 7;
  // The rows of FOUR steps are in flight in registers ahead of the one being multiplied: with one step ahead every step
  // (32 tokens, 8-16 matrix instructions per wave) waited a full trip to memory, ~1.4 us: 127 steps = the kernel's 180 us.
  constexpr int RD = 4;
  u32x4 gv[RD], xv[RD][XP];
  float dbacc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbacc[j] = 0.f;
  constexpr bool ADD = K == 128;   // (only the attention block's q | k rows add a second operand, x + pos: K = 128)
  u32x4 av[RD][ADD ? XP : 1];   // added when the step is stashed: no arithmetic on registers in flight
  auto fetch = [&](int st, int s) {
    const int64_t r = t0 + (int64_t)s * 32 + grow;
    gv[st] = u32x4{0u, 0u, 0u, 0u};
    if (r < t1) gv[st] = *(const u32x4*)(gp + r * ldg + gpc * 8);
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int i = threadIdx.x + j * kThreads, row = i / (K / 8), pc = i % (K / 8);
      const int64_t rr = t0 + (int64_t)s * 32 + row;
      xv[st][j] = u32x4{0u, 0u, 0u, 0u};
      if constexpr (ADD) av[st][j] = u32x4{0u, 0u, 0u, 0u};
      if (rr < t1) {
        xv[st][j] = *(const u32x4*)(xp + rr * K + pc * 8);
        if constexpr (ADD) {
          if (ap) av[st][j] = *(const u32x4*)(ap + rr * K + pc * 8);
        }
      }
    }
  };
  auto stash = [&](int buf, int st) {
    *(u32x4*)(gs + (buf * 32 + grow) * LDG + gpc * 8) = gv[st];
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int i = threadIdx.x + j * kThreads, row = i / (K / 8), pc = i % (K / 8);
      u32x4 v = xv[st][j];
      if constexpr (ADD) if (ap) {
        v.x = add_bf16x2(v.x, av[st][j].x);
        v.y = add_bf16x2(v.y, av[st][j].y);
        v.z = add_bf16x2(v.z, av[st][j].z);
        v.w = add_bf16x2(v.w, av[st][j].w);
      }
      *(u32x4*)(xs + (buf * 32 + row) * LDK + pc * 8) = v;
    }
    dbacc[0] += __uint_as_float(gv[st].x << 16);
    dbacc[1] += __uint_as_float(gv[st].x & 0xffff0000u);
    dbacc[2] += __uint_as_float(gv[st].y << 16);
    dbacc[3] += __uint_as_float(gv[st].y & 0xffff0000u);
    dbacc[4] += __uint_as_float(gv[st].z << 16);
    dbacc[5] += __uint_as_float(gv[st].z & 0xffff0000u);
    dbacc[6] += __uint_as_float(gv[st].w << 16);
    dbacc[7] += __uint_as_float(gv[st].w & 0xffff0000u);
  };
  constexpr int KB = K / 64;      // 16-column blocks of X per wave
  f32x4 acc[4][KB];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc[nb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < RD; ++st) fetch(st, st);   // (steps past the slab's end fetch nothing: zero rows)
  if (steps > 0) stash(0, 0);
  __syncthreads();
  for (int s0 = 0; s0 < steps; s0 += RD) {
#pragma unroll
    for (int d = 0; d < RD; ++d) {   // ring positions are compile-time
      const int s = s0 + d;
      if (s >= steps) break;
      const int buf = s & 1;
      fetch(d, s + RD);              // the stage of step s was stashed before this step: free again
      const uint16_t* gb = gs + (buf * 32 + 4 * g + q_) * LDG + 4 * p_;
      const uint16_t* xb = xs + (buf * 32 + 4 * g + q_) * LDK + 4 * p_ + wave * (16 * KB);
      bf16x8 a[4], b[KB];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) a[nb] = tr_pair(gb + 16 * nb, gb + 16 * nb + 16 * LDG);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) b[kb] = tr_pair(xb + 16 * kb, xb + 16 * kb + 16 * LDK);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc[nb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nb], b[kb], acc[nb][kb], 0, 0, 0);
      if (s + 1 < steps) stash(buf ^ 1, (d + 1) % RD);
      __syncthreads();
    }
  }
  const int N = pk.n[t];
  float* dw = pk.dw[t] + ((int64_t)slab * N + n0) * K;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dw[(int64_t)(nb * 16 + 4 * g + r) * K + wave * (16 * KB) + kb * 16 + c] = acc[nb][kb][r];
  // bias: the 32 threads that staged the same 8 columns (all rows) combine through LDS
  float* red = (float*)smem;   // [32 rows][64 columns]; the staging buffers are dead (barrier at the end of the loop)
#pragma unroll
  for (int j = 0; j < 8; ++j) red[grow * kWgSlice + gpc * 8 + j] = dbacc[j];
  __syncthreads();
  if (threadIdx.x < kWgSlice) {
    float sum = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) sum += red[r * kWgSlice + threadIdx.x];
    pk.db[t][(int64_t)slab * N + n0 + threadIdx.x] = sum;
  }
}

constexpr int kWgradLds = 2 * 32 * (kWgSlice + 16) * 2 + 2 * 32 * (256 + 16) * 2;

__global__ void __launch_bounds__(kThreads, 2) token_wgrad_kernel(WgradPack pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int total = pk.first_slice[pk.count];
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  const int slab = xcd + 8 * (j / total), sl = j % total;
  int t = 0;
  while (t + 1 < pk.count && sl >= pk.first_slice[t + 1]) ++t;
  const int slice = sl - pk.first_slice[t];
  if (pk.k[t] == 128) wgrad_body<128>(pk, t, slice, slab, smem);
  else wgrad_body<256>(pk, t, slice, slab, smem);
}

// dst[c] = sum over rows of src[r][c], rows added in a fixed order (eight running sums per thread), up to 16 tensors
constexpr int kSumMax = 16;
struct RowSumPack {
  const float* src[kSumMax];
  float* dst[kSumMax];
  int32_t rows[kSumMax];
  int64_t cols[kSumMax];
  int32_t first_block[kSumMax + 1];
  int32_t count;
};
__global__ void __launch_bounds__(256) partial_rows_sum_kernel(RowSumPack pk) {
  int t = 0;
  while (t + 1 < pk.count && (int)blockIdx.x >= pk.first_block[t + 1]) ++t;
  const int64_t col = ((int64_t)blockIdx.x - pk.first_block[t]) * 256 + threadIdx.x, cols = pk.cols[t];
  if (col >= cols) return;
  const float* src = pk.src[t] + col;
  const int rows = pk.rows[t];
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.f;
  int r = 0;
  for (; r + 8 <= rows; r += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += src[(int64_t)(r + u) * cols];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (r + u < rows) a[u] += src[(int64_t)(r + u) * cols];
  pk.dst[t][col] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// persistent grids: `per_cu` workgroups on every CU, never more workgroups than tiles
inline int block_grid(int64_t tiles, int per_cu) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  const int64_t cap = (int64_t)cus * per_cu;
  return (int)(tiles < cap ? (tiles < 1 ? 1 : tiles) : cap);
}

}  // namespace

extern "C" int ococc_linear_fragments_bf16(int32_t count, const void* const* src, const int64_t* rows,
                                           const int64_t* cols, const int64_t* row_stride,
                                           const int64_t* col_stride, void* const* dst, ococc_stream_t stream) {
  OCOCC_REQUIRE(count >= 0 && count <= kMaxFrag, "at most 16 matrices per call");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(src && rows && cols && row_stride && col_stride && dst, "null pointer table");
  FragPack pk;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    OCOCC_REQUIRE(src[i] && dst[i] && rows[i] > 0 && cols[i] > 0 && rows[i] % 16 == 0 && cols[i] % 32 == 0 &&
                      rows[i] * cols[i] < (1 << 30),
                  "a matrix needs rows in multiples of 16 and columns in multiples of 32");
    pk.src[i] = (const float*)src[i];
    pk.dst[i] = (uint16_t*)dst[i];
    pk.rows[i] = (int32_t)rows[i];
    pk.cols[i] = (int32_t)cols[i];
    pk.rs[i] = row_stride[i];
    pk.cs[i] = col_stride[i];
    pk.first_block[i] = blocks;
    blocks += (int)(ococc_cdiv(rows[i] * cols[i], 256) < 64 ? ococc_cdiv(rows[i] * cols[i], 256) : 64);
  }
  pk.first_block[count] = blocks;
  pk.count = count;
  hipLaunchKernelGGL(linear_fragments_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pk);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int64_t ococc_window_tile_plan_workspace_bytes(int64_t num_windows) {
  return num_windows < 0 ? -1 : 2 * ococc_align_up(num_windows * 4, 256) + 256;
}

extern "C" int ococc_window_tile_plan(const int32_t* win_len, const int64_t* win_off, const int32_t* tok,
                                      int64_t num_windows, int32_t tile_slots, int64_t cap_tiles, int32_t* tile_rows,
                                      int32_t* tile_span, int32_t* num_tiles, void* workspace, int64_t workspace_bytes,
                                      ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(tile_slots == TM, "tiles have 64 token slots");
  OCOCC_REQUIRE(num_windows >= 0 && cap_tiles >= num_windows, "cap_tiles must be at least num_windows (worst case)");
  OCOCC_REQUIRE(num_tiles, "null pointer");
  if (num_windows == 0) {
    OCOCC_HIP(hipMemsetAsync(num_tiles, 0, 4, stream));
    return OCOCC_OK;
  }
  OCOCC_REQUIRE(win_len && win_off && tok && tile_rows && tile_span && workspace, "null pointer");
  OCOCC_REQUIRE(workspace_bytes >= ococc_window_tile_plan_workspace_bytes(num_windows), "workspace too small");
  OCOCC_REQUIRE(num_windows < ((int64_t)1 << 31) / 64, "too many windows");
  int32_t* win_tile = (int32_t*)workspace;
  int32_t* win_base = (int32_t*)((char*)workspace + ococc_align_up(num_windows * 4, 256));
  OCOCC_HIP(hipMemsetAsync(tile_rows, 0xff, (size_t)cap_tiles * TM * 4, stream));
  OCOCC_HIP(hipMemsetAsync(tile_span, 0, (size_t)cap_tiles * TM * 4, stream));
  const int plan_lds = num_windows <= kPlanLdsWindows ? (int)ococc_align_up(num_windows, 16) : 0;
  OCOCC_HIP(hipFuncSetAttribute((const void*)window_tile_plan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, plan_lds));
  hipLaunchKernelGGL(window_tile_plan_kernel, dim3(1), dim3(kPlanThreads), plan_lds, stream, win_len, num_windows,
                     win_tile, win_base, num_tiles);
  hipLaunchKernelGGL(window_tile_fill_kernel, dim3((unsigned)ococc_cdiv(num_windows * 64, 256)), dim3(256), 0, stream,
                     win_len, win_off, tok, num_windows, win_tile, win_base, tile_rows, tile_span);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

#define OCOCC_BLOCK_DIMS_OK(d_model, heads, ffn) \
  OCOCC_REQUIRE((d_model) == E && (heads) == NH && (ffn) == FF, "the fused encoder-layer kernels are built for d_model 128, 8 heads, feed-forward 256")

namespace {
int attn_block_fwd(const uint16_t* x, const uint16_t* pos, const int32_t* tile_rows, const int32_t* tile_span,
                   int64_t num_tiles, int32_t d_model, int32_t num_heads, const uint16_t* wqkv_frag, const float* bqkv,
                   const uint16_t* wo_frag, const float* bo, const float* ln_weight, const float* ln_bias, float eps,
                   uint16_t* y, uint16_t* attn_save, float* lse_save, ococc_stream_t stream) {
  OCOCC_BLOCK_DIMS_OK(d_model, num_heads, FF);
  OCOCC_REQUIRE(num_tiles >= 0, "bad sizes");
  if (num_tiles == 0) return OCOCC_OK;
  OCOCC_REQUIRE(x && tile_rows && tile_span && wqkv_frag && bqkv && wo_frag && bo && ln_weight && ln_bias && y,
                "null pointer");
  OCOCC_REQUIRE(aligned16(x) && aligned16(pos) && aligned16(y) && aligned16(wqkv_frag) && aligned16(wo_frag) &&
                    aligned16(bqkv) && aligned16(bo) && aligned16(ln_weight) && aligned16(ln_bias),
                "buffers must be 16-byte aligned");
  OCOCC_HIP(hipFuncSetAttribute((const void*)window_attn_block_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                attn_lds<4>()));
  hipLaunchKernelGGL(window_attn_block_fwd_kernel, dim3((unsigned)block_grid(num_tiles, 2)), dim3(256), attn_lds<4>(),
                     (hipStream_t)stream, x, pos, tile_rows, tile_span, num_tiles, wqkv_frag, bqkv, wo_frag, bo,
                     ln_weight, ln_bias, eps, y, attn_save, lse_save);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
}  // namespace

extern "C" int ococc_window_attn_block_fwd_bf16(const uint16_t* x, const uint16_t* pos, const int32_t* tile_rows,
                                                const int32_t* tile_span, int64_t num_tiles, int32_t d_model,
                                                int32_t num_heads, const uint16_t* wqkv_frag, const float* bqkv,
                                                const uint16_t* wo_frag, const float* bo, const float* ln_weight,
                                                const float* ln_bias, float eps, uint16_t* y, ococc_stream_t stream) {
  return attn_block_fwd(x, pos, tile_rows, tile_span, num_tiles, d_model, num_heads, wqkv_frag, bqkv, wo_frag, bo, ln_weight,
                        ln_bias, eps, y, nullptr, nullptr, stream);
}

extern "C" int ococc_window_attn_block_train_fwd_bf16(const uint16_t* x, const uint16_t* pos, const int32_t* tile_rows,
                                                      const int32_t* tile_span, int64_t num_tiles, int32_t d_model,
                                                      int32_t num_heads, const uint16_t* wqkv_frag, const float* bqkv,
                                                      const uint16_t* wo_frag, const float* bo, const float* ln_weight,
                                                      const float* ln_bias, float eps, uint16_t* y, uint16_t* attn_save,
                                                      float* lse_save, ococc_stream_t stream) {
  OCOCC_REQUIRE(num_tiles == 0 || (attn_save && lse_save && aligned16(attn_save) && aligned16(lse_save)),
                "attn_save / lse_save: 16-byte aligned device buffers");
  return attn_block_fwd(x, pos, tile_rows, tile_span, num_tiles, d_model, num_heads, wqkv_frag, bqkv, wo_frag, bo, ln_weight,
                        ln_bias, eps, y, attn_save, lse_save, stream);
}

namespace {
int attn_block_bwd(const uint16_t* x, const uint16_t* pos, const uint16_t* dy, const int32_t* tile_rows,
                   const int32_t* tile_span, int64_t num_tiles, int32_t d_model, int32_t num_heads, const uint16_t* wqkv_frag,
                   const float* bqkv, const uint16_t* wo_frag, const float* bo, const float* ln_weight, float eps,
                   const uint16_t* wo_t_frag, const uint16_t* wqkv_t_frag, uint16_t* dx, uint16_t* dqkv, uint16_t* dz,
                   uint16_t* attn_out, float* ln_partial, const uint16_t* attn_saved, const float* lse_saved,
                   ococc_stream_t stream) {
  OCOCC_BLOCK_DIMS_OK(d_model, num_heads, FF);
  OCOCC_REQUIRE(num_tiles >= 0, "bad sizes");
  if (num_tiles == 0) return OCOCC_OK;
  OCOCC_REQUIRE(x && dy && tile_rows && tile_span && wqkv_frag && bqkv && wo_frag && bo && ln_weight && wo_t_frag &&
                    wqkv_t_frag && dx && dqkv && dz && (attn_out || attn_saved) && ln_partial,
                "null pointer");
  OCOCC_REQUIRE((attn_saved == nullptr) == (lse_saved == nullptr) && aligned16(attn_saved) && aligned16(lse_saved),
                "attn_saved and lse_saved go together, 16-byte aligned");
  OCOCC_REQUIRE(aligned16(x) && aligned16(pos) && aligned16(dy) && aligned16(dx) && aligned16(dqkv) && aligned16(dz) &&
                    aligned16(attn_out) && aligned16(wqkv_frag) && aligned16(wo_frag) && aligned16(wo_t_frag) &&
                    aligned16(wqkv_t_frag) && aligned16(bqkv) && aligned16(bo) && aligned16(ln_weight),
                "buffers must be 16-byte aligned");
  OCOCC_HIP(hipFuncSetAttribute((const void*)window_attn_block_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                attn_lds<8>()));
  hipLaunchKernelGGL(window_attn_block_bwd_kernel, dim3((unsigned)block_grid(num_tiles, 1)), dim3(512), attn_lds<8>(),
                     (hipStream_t)stream, x, pos, dy, tile_rows, tile_span, num_tiles, wqkv_frag, bqkv, wo_frag, bo,
                     ln_weight, eps, wo_t_frag, wqkv_t_frag, dx, dqkv, dz, attn_out, ln_partial, attn_saved, lse_saved);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
}  // namespace

extern "C" int ococc_window_attn_block_bwd_bf16(const uint16_t* x, const uint16_t* pos, const uint16_t* dy,
                                                const int32_t* tile_rows, const int32_t* tile_span,
                                                int64_t num_tiles, int32_t d_model, int32_t num_heads,
                                                const uint16_t* wqkv_frag, const float* bqkv, const uint16_t* wo_frag,
                                                const float* bo, const float* ln_weight, float eps,
                                                const uint16_t* wo_t_frag, const uint16_t* wqkv_t_frag, uint16_t* dx,
                                                uint16_t* dqkv, uint16_t* dz, uint16_t* attn_out, float* ln_partial,
                                                ococc_stream_t stream) {
  return attn_block_bwd(x, pos, dy, tile_rows, tile_span, num_tiles, d_model, num_heads, wqkv_frag, bqkv, wo_frag, bo,
                        ln_weight, eps, wo_t_frag, wqkv_t_frag, dx, dqkv, dz, attn_out, ln_partial, nullptr, nullptr, stream);
}

extern "C" int ococc_window_attn_block_bwd_saved_bf16(const uint16_t* x, const uint16_t* pos, const uint16_t* dy,
                                                      const int32_t* tile_rows, const int32_t* tile_span,
                                                      int64_t num_tiles, int32_t d_model, int32_t num_heads,
                                                      const uint16_t* wqkv_frag, const float* bqkv, const uint16_t* wo_frag,
                                                      const float* bo, const float* ln_weight, float eps,
                                                      const uint16_t* wo_t_frag, const uint16_t* wqkv_t_frag,
                                                      const uint16_t* attn_saved, const float* lse_saved, uint16_t* dx,
                                                      uint16_t* dqkv, uint16_t* dz, float* ln_partial,
                                                      ococc_stream_t stream) {
  OCOCC_REQUIRE(num_tiles == 0 || (attn_saved && lse_saved), "null pointer");
  return attn_block_bwd(x, pos, dy, tile_rows, tile_span, num_tiles, d_model, num_heads, wqkv_frag, bqkv, wo_frag, bo,
                        ln_weight, eps, wo_t_frag, wqkv_t_frag, dx, dqkv, dz, nullptr, ln_partial, attn_saved, lse_saved,
                        stream);
}

extern "C" int ococc_token_ffn_block_fwd_bf16(const uint16_t* x, int64_t num_tokens, int32_t d_model, int32_t d_ffn,
                                              const uint16_t* w1_frag, const float* b1, const uint16_t* w2_frag,
                                              const float* b2, const float* ln_weight, const float* ln_bias, float eps,
                                              int32_t act, uint16_t* y, ococc_stream_t stream) {
  OCOCC_BLOCK_DIMS_OK(d_model, NH, d_ffn);
  OCOCC_REQUIRE(num_tokens >= 0 && (act == 0 || act == 1), "bad sizes / act must be 0 (gelu) or 1 (relu)");
  if (num_tokens == 0) return OCOCC_OK;
  OCOCC_REQUIRE(x && w1_frag && b1 && w2_frag && b2 && ln_weight && ln_bias && y, "null pointer");
  OCOCC_REQUIRE(aligned16(x) && aligned16(y) && aligned16(w1_frag) && aligned16(w2_frag) && aligned16(b1) &&
                    aligned16(b2) && aligned16(ln_weight) && aligned16(ln_bias),
                "buffers must be 16-byte aligned");
  const unsigned grid = (unsigned)block_grid(ococc_cdiv(num_tokens, TM), 2);
#define OCOCC_FFN_FWD(A)                                                                                          \
  do {                                                                                                            \
    OCOCC_HIP(hipFuncSetAttribute((const void*)token_ffn_block_fwd_kernel<A>,                                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ffn_lds<4>()));                     \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(token_ffn_block_fwd_kernel<A>), dim3(grid), dim3(256), ffn_lds<4>(),       \
                       (hipStream_t)stream, x, num_tokens, w1_frag, b1, w2_frag, b2, ln_weight, ln_bias, eps, y); \
  } while (0)
  if (act == 0) OCOCC_FFN_FWD(0);
  else OCOCC_FFN_FWD(1);
#undef OCOCC_FFN_FWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_token_ffn_block_bwd_bf16(const uint16_t* x, const uint16_t* dy, int64_t num_tokens,
                                              int32_t d_model, int32_t d_ffn, const uint16_t* w1_frag, const float* b1,
                                              const uint16_t* w2_frag, const float* b2, const float* ln_weight,
                                              float eps, int32_t act, const uint16_t* w2_t_frag,
                                              const uint16_t* w1_t_frag, uint16_t* dx, uint16_t* act_out, uint16_t* dh,
                                              uint16_t* dz, float* ln_partial, ococc_stream_t stream) {
  OCOCC_BLOCK_DIMS_OK(d_model, NH, d_ffn);
  OCOCC_REQUIRE(num_tokens >= 0 && (act == 0 || act == 1), "bad sizes / act must be 0 (gelu) or 1 (relu)");
  if (num_tokens == 0) return OCOCC_OK;
  OCOCC_REQUIRE(x && dy && w1_frag && b1 && w2_frag && b2 && ln_weight && w2_t_frag && w1_t_frag && dx && act_out &&
                    dh && dz && ln_partial,
                "null pointer");
  OCOCC_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(act_out) && aligned16(dh) && aligned16(dz) &&
                    aligned16(w1_frag) && aligned16(w2_frag) && aligned16(w2_t_frag) && aligned16(w1_t_frag) &&
                    aligned16(b1) && aligned16(b2) && aligned16(ln_weight),
                "buffers must be 16-byte aligned");
  const unsigned grid = (unsigned)block_grid(ococc_cdiv(num_tokens, TM), 1);
#define OCOCC_FFN_BWD(A)                                                                                           \
  do {                                                                                                             \
    OCOCC_HIP(hipFuncSetAttribute((const void*)token_ffn_block_bwd_kernel<A>,                                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ffn_lds<8>()));                      \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(token_ffn_block_bwd_kernel<A>), dim3(grid), dim3(512), ffn_lds<8>(),        \
                       (hipStream_t)stream, x, dy, num_tokens, w1_frag, b1, w2_frag, b2, ln_weight, eps, w2_t_frag, \
                       w1_t_frag, dx, act_out, dh, dz, ln_partial);                                                \
  } while (0)
  if (act == 0) OCOCC_FFN_BWD(0);
  else OCOCC_FFN_BWD(1);
#undef OCOCC_FFN_BWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int64_t ococc_window_block_partial_rows(int64_t num_tiles) {
  // rows of ln_partial a backward block kernel writes for this many tiles (one per persistent workgroup)
  return num_tiles < 0 ? -1 : (num_tiles == 0 ? 0 : block_grid(num_tiles, 1));
}

extern "C" int64_t ococc_token_wgrad_slabs(int64_t num_tokens) {
  // 64 slabs of the token range once there is enough work for them (>= 256 tokens each), 8 at least: a multiple of 8,
  // so that the workgroups of a slab share an XCD
  if (num_tokens < 0) return -1;
  int64_t s = num_tokens / 256;
  s = s < 8 ? 8 : (s > 64 ? 64 : s / 8 * 8);
  return s;
}

extern "C" int ococc_token_wgrad_bf16(int32_t count, const void* const* g, const int64_t* ldg, const int64_t* n,
                                      const void* const* x, const void* const* xadd, const int64_t* add_rows,
                                      const int64_t* k, int64_t num_tokens, int64_t slabs, void* const* dw_partial,
                                      void* const* db_partial, ococc_stream_t stream) {
  OCOCC_REQUIRE(count >= 0 && count <= kWgMax, "at most 8 weight gradients per call");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(g && ldg && n && x && xadd && add_rows && k && dw_partial && db_partial, "null pointer table");
  OCOCC_REQUIRE(num_tokens >= 0 && slabs >= 8 && slabs % 8 == 0, "slabs must be a positive multiple of 8");
  WgradPack pk;
  int slices = 0;
  for (int i = 0; i < count; ++i) {
    OCOCC_REQUIRE(g[i] && x[i] && dw_partial[i] && db_partial[i], "null pointer");
    OCOCC_REQUIRE(n[i] > 0 && n[i] % kWgSlice == 0 && (k[i] == 128 || k[i] == 256) && ldg[i] >= n[i] && ldg[i] % 8 == 0,
                  "dW rows in multiples of 64, 128 or 256 columns, 16-byte aligned gradient rows");
    OCOCC_REQUIRE(aligned16(g[i]) && aligned16(x[i]) && aligned16(xadd[i]), "operands must be 16-byte aligned");
    OCOCC_REQUIRE(!xadd[i] || k[i] == 128, "a second x operand (xadd) only with 128 columns");
    pk.g[i] = (const uint16_t*)g[i];
    pk.x[i] = (const uint16_t*)x[i];
    pk.xadd[i] = (const uint16_t*)xadd[i];
    pk.dw[i] = (float*)dw_partial[i];
    pk.db[i] = (float*)db_partial[i];
    pk.ldg[i] = (int32_t)ldg[i];
    pk.n[i] = (int32_t)n[i];
    pk.k[i] = (int32_t)k[i];
    pk.add_rows[i] = (int32_t)add_rows[i];
    pk.first_slice[i] = slices;
    slices += (int)(n[i] / kWgSlice);
  }
  pk.first_slice[count] = slices;
  pk.count = count;
  pk.slabs = (int32_t)slabs;
  pk.tokens = num_tokens;
  pk.chunk = ococc_align_up(ococc_cdiv(num_tokens > 0 ? num_tokens : 1, slabs), 32);
  OCOCC_HIP(hipFuncSetAttribute((const void*)token_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWgradLds));
  hipLaunchKernelGGL(token_wgrad_kernel, dim3((unsigned)(slices * slabs)), dim3(kThreads), kWgradLds,
                     (hipStream_t)stream, pk);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_partial_rows_sum_f32(int32_t count, const void* const* src, const int64_t* rows,
                                          const int64_t* cols, void* const* dst, ococc_stream_t stream) {
  OCOCC_REQUIRE(count >= 0 && count <= kSumMax, "at most 16 tensors per call");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(src && rows && cols && dst, "null pointer table");
  RowSumPack pk;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    OCOCC_REQUIRE(src[i] && dst[i] && rows[i] >= 0 && rows[i] < (1 << 30) && cols[i] > 0, "bad tensor");
    pk.src[i] = (const float*)src[i];
    pk.dst[i] = (float*)dst[i];
    pk.rows[i] = (int32_t)rows[i];
    pk.cols[i] = cols[i];
    pk.first_block[i] = blocks;
    blocks += (int)ococc_cdiv(cols[i], 256);
  }
  pk.first_block[count] = blocks;
  pk.count = count;
  hipLaunchKernelGGL(partial_rows_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pk);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
