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
constexpr int LDX = E + 16;      // LDS row strides (elements): +32 B keeps ds_read_b128 of 16 rows conflict free
constexpr int LDQ = 3 * E + 16;
constexpr int LDH = FF + 16;
constexpr int kThreads = 256;

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
  u32x2 v;
  v.x = (uint32_t)ococc_f32_to_bf16(o[0]) | ((uint32_t)ococc_f32_to_bf16(o[1]) << 16);
  v.y = (uint32_t)ococc_f32_to_bf16(o[2]) | ((uint32_t)ococc_f32_to_bf16(o[3]) << 16);
  return v;
}
__device__ __forceinline__ f32x4 unpack4(const u32x2 v) {
  return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
               __uint_as_float(v.y & 0xffff0000u)};
}
__device__ __forceinline__ s16x4 ld4(const uint16_t* p) { return *(const s16x4*)p; }

// ---------------------------------------------------------------------------------------------------------------
// out^T[n][m] += sum_k W[n][k] X[m][k] for the wave's NB blocks of 16 output channels starting at block nb0 and the
// 4 token blocks of the tile.  wf: fragment-major weights [n block][k step][lane][8]; xs: LDS tile [64][ldb] bf16.
// All fragments of the wave's slice are requested before the first MFMA (NB*KS <= 24 loads of 16 B per lane in
// flight): the L2 latency of the weight stream is paid once per GEMM, not once per k step.
template <int NB, int KS>
__device__ __forceinline__ void tile_gemm(const uint16_t* __restrict__ wf, int nb0, const uint16_t* xs, int ldb,
                                          f32x4 (&acc)[NB][4]) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const bf16x8* wp = (const bf16x8*)wf + (size_t)nb0 * KS * 64 + lane;
  bf16x8 a[NB][KS];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[nb][ks] = wp[(nb * KS + ks) * 64];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 b[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) b[mb] = *(const bf16x8*)(xs + (mb * 16 + c) * ldb + ks * 32 + 8 * g);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nb][ks], b[mb], acc[nb][mb], 0, 0, 0);
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
// partner lanes (same c, other g) and the other three waves through `red` ([4 waves][64 tokens] floats).  One barrier.
__device__ __forceinline__ void token_sums(float (&part)[4], float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    float p = part[mb];
    p += __shfl_xor(p, 16, 64);
    p += __shfl_xor(p, 32, 64);
    if (g == 0) red[wave * TM + mb * 16 + c] = p;
  }
  __syncthreads();
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int t = mb * 16 + c;
    part[mb] = (red[t] + red[TM + t]) + (red[2 * TM + t] + red[3 * TM + t]);
  }
}

// LayerNorm statistics of z (the wave's 2 channel blocks x 4 token blocks; a token's 128 channels are spread over the
// 4 waves): two passes (mean, then centred squares).  z <- xhat = (z - mean) * rstd; rstd returned per token block.
__device__ __forceinline__ void tile_layernorm(f32x4 (&z)[2][4], float eps, float* red0, float* red1, float (&rstd)[4]) {
  float s[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    s[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[mb] += z[nb][mb][r];
  }
  token_sums(s, red0);
  float q[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const float mean = s[mb] * (1.f / E);
    q[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        z[nb][mb][r] -= mean;
        q[mb] += z[nb][mb][r] * z[nb][mb][r];
      }
  }
  token_sums(q, red1);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    rstd[mb] = rsqrtf(q[mb] * (1.f / E) + eps);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[nb][mb][r] *= rstd[mb];
  }
}

// LayerNorm backward on the wave's slice: xh = xhat, dy = gradient of the LN output -> dy <- gradient of the LN
// input; adds the tile's terms to dgam / dbet (per lane: channels 16(nb0+nb)+4g+r, summed over the lane's tokens).
__device__ __forceinline__ void tile_layernorm_bwd(const f32x4 (&xh)[2][4], f32x4 (&dy)[2][4], const f32x4 (&gam)[2],
                                                   const float (&rstd)[4], float* red0, float* red1, f32x4 (&dgam)[2],
                                                   f32x4 (&dbet)[2]) {
  float s1[4], s2[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    s1[mb] = s2[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
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
  token_sums(s1, red0);
  token_sums(s2, red1);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const float m1 = s1[mb] * (1.f / E), m2 = s2[mb] * (1.f / E);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dy[nb][mb][r] = ((dy[nb][mb][r] - m1) - xh[nb][mb][r] * m2) * rstd[mb];
  }
}

// per-channel sums of the wave's slice over the tile's tokens -> one row of partial sums per tile ([2][128] floats)
__device__ __forceinline__ void store_param_partials(const f32x4 (&dgam)[2], const f32x4 (&dbet)[2], int nb0,
                                                     float* __restrict__ dst) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
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

// rows of a [64][ld] bf16 LDS tile <-> rows of a global [*, width] bf16 tensor, 16 B per lane, whole rows coalesced
template <int WIDTH>
__device__ __forceinline__ void tile_store_rows(const uint16_t* ts, int ld, uint16_t* __restrict__ dst, const int* rows_s,
                                                int64_t row0, int64_t nrows) {
  constexpr int PP = WIDTH / 8;
  for (int i = threadIdx.x; i < TM * PP; i += kThreads) {
    const int s = i / PP, p = i % PP;
    const int64_t r = rows_s ? (int64_t)rows_s[s] : (row0 + s < nrows ? row0 + s : -1);
    if (r >= 0) *(u32x4*)(dst + r * WIDTH + p * 8) = *(const u32x4*)(ts + s * ld + p * 8);
  }
}
__device__ __forceinline__ void tile_load_rows(uint16_t* ts, int ld, const uint16_t* __restrict__ src, const int* rows_s,
                                               int64_t row0, int64_t nrows) {
  constexpr int PP = E / 8;
  for (int i = threadIdx.x; i < TM * PP; i += kThreads) {
    const int s = i / PP, p = i % PP;
    const int64_t r = rows_s ? (int64_t)rows_s[s] : (row0 + s < nrows ? row0 + s : -1);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r >= 0) v = *(const u32x4*)(src + r * E + p * 8);
    *(u32x4*)(ts + s * ld + p * 8) = v;
  }
}

__device__ __forceinline__ uint32_t add_bf16x2(uint32_t a, uint32_t b) {
  const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
  const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
  return (uint32_t)ococc_f32_to_bf16(lo) | ((uint32_t)ococc_f32_to_bf16(hi) << 16);
}

template <int ACT>  // 0 gelu (erf), 1 relu
__device__ __forceinline__ float act_fwd(float h) {
  return ACT == 0 ? ln_gelu1(h) : fmaxf(h, 0.f);
}
template <int ACT>
__device__ __forceinline__ float act_grad(float h) {
  return ACT == 0 ? ln_gelu_grad2(ln_f32x2{h, h}).x : (h > 0.f ? 1.f : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------
// Shared front of the attention block (forward and the recompute of the backward):
//   xs <- x rows of the tile;  V = Wv x + bv;  xs += pos;  Q | K = Wqk (x + pos) + bqk   -> qs [64][Q | K | V]
struct TileMeta {
  int rows[TM];   // flat token row of each slot, -1 = empty
  int span[TM];   // lo | hi << 8: the slots of the slot's window; 0 for an empty slot
};

__device__ __forceinline__ void attn_front(const uint16_t* __restrict__ x, const uint16_t* __restrict__ pos,
                                           const uint16_t* __restrict__ wqkv, const float* __restrict__ bqkv,
                                           const TileMeta* tm, uint16_t* xs, uint16_t* qs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  // x -> xs, the matching pieces of pos kept in registers until V has been computed
  u32x4 pv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = threadIdx.x + j * kThreads, s = i >> 4, p = i & 15;
    const int r = tm->rows[s];
    u32x4 v = {0u, 0u, 0u, 0u};
    pv[j] = u32x4{0u, 0u, 0u, 0u};
    if (r >= 0) {
      v = *(const u32x4*)(x + (int64_t)r * E + p * 8);
      if (pos) pv[j] = *(const u32x4*)(pos + (int64_t)r * E + p * 8);
    }
    *(u32x4*)(xs + s * LDX + p * 8) = v;
  }
  __syncthreads();
  {
    f32x4 acc[2][4];
    zero_acc(acc);
    tile_gemm<2, 4>(wqkv, 16 + 2 * wave, xs, LDX, acc);   // V: channel blocks 16..23 of the in-projection
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = 16 * (16 + 2 * wave + nb) + 4 * g;
      const f32x4 b = *(const f32x4*)(bqkv + n);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) *(u32x2*)(qs + (mb * 16 + c) * LDQ + n) = pack4(acc[nb][mb] + b);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {   // xs <- bf16(x + pos), in place, every thread its own pieces
    const int i = threadIdx.x + j * kThreads, s = i >> 4, p = i & 15;
    u32x4 v = *(const u32x4*)(xs + s * LDX + p * 8);
    v.x = add_bf16x2(v.x, pv[j].x);
    v.y = add_bf16x2(v.y, pv[j].y);
    v.z = add_bf16x2(v.z, pv[j].z);
    v.w = add_bf16x2(v.w, pv[j].w);
    if (pos) *(u32x4*)(xs + s * LDX + p * 8) = v;
  }
  __syncthreads();
  {
    f32x4 acc[4][4];
    zero_acc(acc);
    tile_gemm<4, 4>(wqkv, 4 * wave, xs, LDX, acc);        // Q | K: channel blocks 0..15
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = 16 * (4 * wave + nb) + 4 * g;
      const f32x4 b = *(const f32x4*)(bqkv + n);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) *(u32x2*)(qs + (mb * 16 + c) * LDQ + n) = pack4(acc[nb][mb] + b);
    }
  }
  __syncthreads();
}

// key-tile range [lo, hi] that the 16 queries of query tile qt can see (union of their windows); hi < lo: none
__device__ __forceinline__ void tile_range(const TileMeta* tm, int qt, int& klo, int& khi) {
  const int c = threadIdx.x & 15;
  const int sp = tm->span[qt * 16 + c];
  int lo = sp & 255, hi = sp >> 8;
  int a = hi > lo ? (lo >> 4) : 99, b = hi > lo ? ((hi - 1) >> 4) : -1;
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    a = min(a, __shfl_xor(a, m, 64));
    b = max(b, __shfl_xor(b, m, 64));
  }
  klo = a;
  khi = b;
}

// Attention of head h for the 16 queries of tile qt (lane: query c): S^T = K Q^T (16x16x16 MFMA, keys on the rows),
// softmax over the keys of the query's window in registers, O^T = V^T P^T (16x16x32 MFMA, V^T by transposing reads).
// Returns o (d = 4g + r of query c) and the log-sum-exp of the query.
__device__ __forceinline__ f32x4 attn_head_fwd(const uint16_t* qs, int h, int qt, int klo, int khi, int span_q,
                                               float& lse) {
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int q_ = c >> 2, p_ = c & 3;
  const uint16_t* qh = qs + h * HD;
  const uint16_t* kh = qs + E + h * HD;
  const uint16_t* vh = qs + 2 * E + h * HD;
  const int lo = span_q & 255, hi = span_q >> 8;
  const s16x4 bq = ld4(qh + (qt * 16 + c) * LDQ + 4 * g);
  f32x4 s[4];
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (kt >= klo && kt <= khi) {
      const s16x4 ak = ld4(kh + (kt * 16 + c) * LDQ + 4 * g);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ak, bq, acc, 0, 0, 0);   // rows: keys 4g+r, col: query c
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        const float val = (key >= lo && key < hi) ? acc[r] * 0.25f : -INFINITY;   // 1 / sqrt(16)
        s[kt][r] = val;
        m = fmaxf(m, val);
      }
    }
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = m > -INFINITY ? __expf(s[kt][r] - m) : 0.f;
      s[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  lse = sum > 0.f ? m + __logf(sum) : 0.f;
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

constexpr int kAttnLds = (TM * LDX + TM * LDQ) * 2 + 2 * 4 * TM * 4 + (int)sizeof(TileMeta) + NH * TM * 4;

// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads, 2)
window_attn_block_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ pos,
                             const int32_t* __restrict__ tile_rows, const int32_t* __restrict__ tile_span,
                             const uint16_t* __restrict__ wqkv, const float* __restrict__ bqkv,
                             const uint16_t* __restrict__ wo, const float* __restrict__ bo,
                             const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                             uint16_t* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* xs = (uint16_t*)smem;                 // x | x + pos | attention output | y staging
  uint16_t* qs = xs + TM * LDX;                   // Q | K | V
  float* red0 = (float*)(qs + TM * LDQ);
  float* red1 = red0 + 4 * TM;
  TileMeta* tm = (TileMeta*)(red1 + 4 * TM);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int64_t tile = blockIdx.x;
  if (threadIdx.x < TM) {
    tm->rows[threadIdx.x] = tile_rows[tile * TM + threadIdx.x];
    tm->span[threadIdx.x] = tile_span[tile * TM + threadIdx.x];
  }
  __syncthreads();
  attn_front(x, pos, wqkv, bqkv, tm, xs, qs);
  // attention: wave w runs heads 2w, 2w + 1; the output goes where x + pos was (its last reader was the Q | K GEMM)
  for (int qt = 0; qt < 4; ++qt) {
    int klo, khi;
    tile_range(tm, qt, klo, khi);
    const int sp = tm->span[qt * 16 + c];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = 2 * wave + hh;
      float lse;
      const f32x4 o = attn_head_fwd(qs, h, qt, klo, khi, sp, lse);
      *(u32x2*)(xs + (qt * 16 + c) * LDX + h * HD + 4 * g) = pack4(o);
    }
  }
  __syncthreads();
  // out-projection + residual + LayerNorm
  f32x4 z[2][4];
  zero_acc(z);
  tile_gemm<2, 4>(wo, 2 * wave, xs, LDX, z);
  f32x4 gam[2], bet[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = 16 * (2 * wave + nb) + 4 * g;
    const f32x4 b = *(const f32x4*)(bo + n);
    gam[nb] = *(const f32x4*)(ln_w + n);
    bet[nb] = *(const f32x4*)(ln_b + n);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int r = tm->rows[mb * 16 + c];
      f32x4 res = {0.f, 0.f, 0.f, 0.f};
      if (r >= 0) res = unpack4(*(const u32x2*)(x + (int64_t)r * E + n));
      z[nb][mb] = z[nb][mb] + b + res;
    }
  }
  float rstd[4];
  tile_layernorm(z, eps, red0, red1, rstd);   // (its first barrier also ends every wave's reads of xs)
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) *(u32x2*)(xs + (mb * 16 + c) * LDX + n) = pack4(z[nb][mb] * gam[nb] + bet[nb]);
  }
  __syncthreads();
  tile_store_rows<E>(xs, LDX, y, tm->rows, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int kFfnLds = (TM * LDX + TM * LDH) * 2 + 2 * 4 * TM * 4;

template <int ACT>
__global__ void __launch_bounds__(kThreads, 2)
token_ffn_block_fwd_kernel(const uint16_t* __restrict__ x, int64_t num_tokens, const uint16_t* __restrict__ w1,
                           const float* __restrict__ b1, const uint16_t* __restrict__ w2, const float* __restrict__ b2,
                           const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                           uint16_t* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* xs = (uint16_t*)smem;
  uint16_t* hs = xs + TM * LDX;
  float* red0 = (float*)(hs + TM * LDH);
  float* red1 = red0 + 4 * TM;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TM;
  tile_load_rows(xs, LDX, x, nullptr, row0, num_tokens);
  __syncthreads();
  {
    f32x4 acc[4][4];
    zero_acc(acc);
    tile_gemm<4, 4>(w1, 4 * wave, xs, LDX, acc);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = 16 * (4 * wave + nb) + 4 * g;
      const f32x4 b = *(const f32x4*)(b1 + n);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        f32x4 h = acc[nb][mb] + b;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = act_fwd<ACT>(h[r]);
        *(u32x2*)(hs + (mb * 16 + c) * LDH + n) = pack4(h);
      }
    }
  }
  __syncthreads();
  f32x4 z[2][4];
  zero_acc(z);
  tile_gemm<2, 8>(w2, 2 * wave, hs, LDH, z);
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
  tile_layernorm(z, eps, red0, red1, rstd);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) *(u32x2*)(hs + (mb * 16 + c) * LDH + n) = pack4(z[nb][mb] * gam[nb] + bet[nb]);
  }
  __syncthreads();
  tile_store_rows<E>(hs, LDH, y, nullptr, row0, num_tokens);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the FFN block.  In: x (= y1), dy (gradient of y2).  Out: dx (gradient of y1, residual path included),
// and the weight-gradient operands a = act(h) [*,256], dh [*,256], dz [*,128] (gradient at the LN input), plus one row
// of LN parameter-gradient partial sums per tile.  act'(h) stays (as bf16) in the registers of the lanes that computed h: the
// gradient of act(h) arrives in the same lanes, because the two GEMMs have the same shape.
template <int ACT>
__global__ void __launch_bounds__(kThreads, 2)
token_ffn_block_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, int64_t num_tokens,
                           const uint16_t* __restrict__ w1, const float* __restrict__ b1,
                           const uint16_t* __restrict__ w2, const float* __restrict__ b2,
                           const float* __restrict__ ln_w, float eps, const uint16_t* __restrict__ w2t,
                           const uint16_t* __restrict__ w1t, uint16_t* __restrict__ dx, uint16_t* __restrict__ a_out,
                           uint16_t* __restrict__ dh_out, uint16_t* __restrict__ dz_out,
                           float* __restrict__ ln_partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* xs = (uint16_t*)smem;      // y1 | dz2 | dx staging
  uint16_t* hs = xs + TM * LDX;        // act(h) | dh
  float* red0 = (float*)(hs + TM * LDH);
  float* red1 = red0 + 4 * TM;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TM;
  tile_load_rows(xs, LDX, x, nullptr, row0, num_tokens);
  __syncthreads();
  u32x2 gact[4][4];   // act'(h) of the lane's 64 pre-activations, packed bf16 (the d act GEMM below has the same shape)
  {
    f32x4 hpre[4][4];
    zero_acc(hpre);
    tile_gemm<4, 4>(w1, 4 * wave, xs, LDX, hpre);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = 16 * (4 * wave + nb) + 4 * g;
      const f32x4 b = *(const f32x4*)(b1 + n);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const f32x4 hp = hpre[nb][mb] + b;
        f32x4 h, dh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h[r] = act_fwd<ACT>(hp[r]);
          dh[r] = act_grad<ACT>(hp[r]);
        }
        gact[nb][mb] = pack4(dh);
        *(u32x2*)(hs + (mb * 16 + c) * LDH + n) = pack4(h);
      }
    }
  }
  __syncthreads();
  tile_store_rows<FF>(hs, LDH, a_out, nullptr, row0, num_tokens);
  f32x4 z[2][4];
  zero_acc(z);
  tile_gemm<2, 8>(w2, 2 * wave, hs, LDH, z);
  f32x4 gam[2], dgam[2], dbet[2];
  f32x4 dz[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = 16 * (2 * wave + nb) + 4 * g;
    const f32x4 b = *(const f32x4*)(b2 + n);
    gam[nb] = *(const f32x4*)(ln_w + n);
    dgam[nb] = dbet[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      z[nb][mb] = z[nb][mb] + b + unpack4(*(const u32x2*)(xs + (mb * 16 + c) * LDX + n));
      const int64_t r = row0 + mb * 16 + c;
      dz[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (r < num_tokens) dz[nb][mb] = unpack4(*(const u32x2*)(dy + r * E + n));
    }
  }
  float rstd[4];
  tile_layernorm(z, eps, red0, red1, rstd);                       // z = xhat2
  tile_layernorm_bwd(z, dz, gam, rstd, red0, red1, dgam, dbet);   // dz = gradient at the LN input (f32)
  store_param_partials(dgam, dbet, 2 * wave, ln_partial + (int64_t)blockIdx.x * 2 * E);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {   // (every wave is past its reads of y1: the barriers of the LN sums)
    const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) *(u32x2*)(xs + (mb * 16 + c) * LDX + n) = pack4(dz[nb][mb]);
  }
  __syncthreads();
  tile_store_rows<E>(xs, LDX, dz_out, nullptr, row0, num_tokens);
#pragma unroll
  for (int half = 0; half < 2; ++half) {   // two channel-block pairs at a time: h, d act and the fragments share 256 registers
    f32x4 da[2][4];
    zero_acc(da);
    tile_gemm<2, 4>(w2t, 4 * wave + 2 * half, xs, LDX, da);   // d act(h) = W2^T dz
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = 16 * (4 * wave + 2 * half + nb) + 4 * g;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const f32x4 d = da[nb][mb] * unpack4(gact[2 * half + nb][mb]);
        *(u32x2*)(hs + (mb * 16 + c) * LDH + n) = pack4(d);   // over act(h): its last reader was the W2 GEMM
      }
    }
  }
  __syncthreads();
  tile_store_rows<FF>(hs, LDH, dh_out, nullptr, row0, num_tokens);
  f32x4 gx[2][4];
  zero_acc(gx);
  tile_gemm<2, 8>(w1t, 2 * wave, hs, LDH, gx);     // W1^T dh
  // (the dz tile in xs -- B operand of the d act GEMM, source of dz_out -- was last read before the barrier above)
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) *(u32x2*)(xs + (mb * 16 + c) * LDX + n) = pack4(gx[nb][mb] + dz[nb][mb]);
  }
  __syncthreads();
  tile_store_rows<E>(xs, LDX, dx, nullptr, row0, num_tokens);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the attention block.  In: x, pos, dy (gradient of y1).  Out: dx, and for the weight gradients dqkv
// [*,384], o (attention output) [*,128], dz (gradient at the LN input) [*,128]; LN partial sums per tile.
__global__ void __launch_bounds__(kThreads, 2)
window_attn_block_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ pos,
                             const uint16_t* __restrict__ dy, const int32_t* __restrict__ tile_rows,
                             const int32_t* __restrict__ tile_span, const uint16_t* __restrict__ wqkv,
                             const float* __restrict__ bqkv, const uint16_t* __restrict__ wo,
                             const float* __restrict__ bo, const float* __restrict__ ln_w, float eps,
                             const uint16_t* __restrict__ wot, const uint16_t* __restrict__ wqkvt,
                             uint16_t* __restrict__ dx, uint16_t* __restrict__ dqkv_out,
                             uint16_t* __restrict__ dz_out, uint16_t* __restrict__ o_out,
                             float* __restrict__ ln_partial) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* xs = (uint16_t*)smem;                 // x | x + pos | o | dz1 | dO | dx staging
  uint16_t* qs = xs + TM * LDX;                   // Q | K | V, then dQ | dK | dV in place
  float* red0 = (float*)(qs + TM * LDQ);
  float* red1 = red0 + 4 * TM;
  TileMeta* tm = (TileMeta*)(red1 + 4 * TM);
  float* lse_s = (float*)(tm + 1);                // [8 heads][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int q_ = c >> 2, p_ = c & 3;
  const int64_t tile = blockIdx.x;
  if (threadIdx.x < TM) {
    tm->rows[threadIdx.x] = tile_rows[tile * TM + threadIdx.x];
    tm->span[threadIdx.x] = tile_span[tile * TM + threadIdx.x];
  }
  __syncthreads();
  attn_front(x, pos, wqkv, bqkv, tm, xs, qs);
  for (int qt = 0; qt < 4; ++qt) {
    int klo, khi;
    tile_range(tm, qt, klo, khi);
    const int sp = tm->span[qt * 16 + c];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int h = 2 * wave + hh;
      float lse;
      const f32x4 o = attn_head_fwd(qs, h, qt, klo, khi, sp, lse);
      *(u32x2*)(xs + (qt * 16 + c) * LDX + h * HD + 4 * g) = pack4(o);
      if (g == 0) lse_s[h * TM + qt * 16 + c] = lse;
    }
  }
  __syncthreads();
  tile_store_rows<E>(xs, LDX, o_out, tm->rows, 0, 0);
  f32x4 z[2][4];
  zero_acc(z);
  tile_gemm<2, 4>(wo, 2 * wave, xs, LDX, z);
  f32x4 gam[2], dgam[2], dbet[2];
  f32x4 dz[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = 16 * (2 * wave + nb) + 4 * g;
    const f32x4 b = *(const f32x4*)(bo + n);
    gam[nb] = *(const f32x4*)(ln_w + n);
    dgam[nb] = dbet[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int r = tm->rows[mb * 16 + c];
      f32x4 res = {0.f, 0.f, 0.f, 0.f};
      dz[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (r >= 0) {
        res = unpack4(*(const u32x2*)(x + (int64_t)r * E + n));
        dz[nb][mb] = unpack4(*(const u32x2*)(dy + (int64_t)r * E + n));
      }
      z[nb][mb] = z[nb][mb] + b + res;
    }
  }
  float rstd[4];
  tile_layernorm(z, eps, red0, red1, rstd);
  tile_layernorm_bwd(z, dz, gam, rstd, red0, red1, dgam, dbet);   // dz = dz1, kept in registers for the residual
  store_param_partials(dgam, dbet, 2 * wave, ln_partial + tile * 2 * E);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {   // over o: every wave is past the out-projection GEMM and the o_out copy
    const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) *(u32x2*)(xs + (mb * 16 + c) * LDX + n) = pack4(dz[nb][mb]);
  }
  __syncthreads();
  tile_store_rows<E>(xs, LDX, dz_out, tm->rows, 0, 0);
  {
    f32x4 go[2][4];
    zero_acc(go);
    tile_gemm<2, 4>(wot, 2 * wave, xs, LDX, go);   // dO = Wo^T dz1
    __syncthreads();                                // dz1 tile: read by every wave's GEMM and by the copy above
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) *(u32x2*)(xs + (mb * 16 + c) * LDX + n) = pack4(go[nb][mb]);
    }
  }
  __syncthreads();
  // attention backward, wave w: heads 2w, 2w + 1.  Pass 1 (lanes own queries): dQ and delta; pass 2 (lanes own
  // keys): dK, dV.  Both recompute the probabilities with 16x16x16 MFMAs; the gradients of a head replace its Q, K, V
  // in place once both passes have read them (only this wave touches the head's columns).
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = 2 * wave + hh;
    const uint16_t* qh = qs + h * HD;
    const uint16_t* kh = qs + E + h * HD;
    const uint16_t* vh = qs + 2 * E + h * HD;
    const uint16_t* dh_ = xs + h * HD;              // dO of the head, row stride LDX
    float* lq = lse_s + h * TM;
    float* dl = red0;                               // delta of the head's queries: [wave][64] (red0 is free here)
    f32x4 gq[4], gk[4], gv[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      int klo, khi;
      tile_range(tm, qt, klo, khi);
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
      if (g == 0) dl[wave * TM + qi] = delta;
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
    // (delta written by this wave's g == 0 lanes, read below by all its lanes: LDS ops of a wave complete in order)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      int qlo, qhi;
      tile_range(tm, kt, qlo, qhi);   // the windows are equivalence classes: the same range read as query tiles
      const int kj = kt * 16 + c;
      const int spk = tm->span[kj];
      const int klo_ = spk & 255, khi_ = spk >> 8;
      const s16x4 bk = ld4(kh + kj * LDQ + 4 * g);
      const s16x4 bv = ld4(vh + kj * LDQ + 4 * g);
      f32x4 dsv[4], pv[4];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        dsv[qt] = pv[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qt >= qlo && qt <= qhi) {
          const s16x4 aq = ld4(qh + (qt * 16 + c) * LDQ + 4 * g);
          const s16x4 ado = ld4(dh_ + (qt * 16 + c) * LDX + 4 * g);
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          const f32x4 sc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aq, bk, zero, 0, 0, 0);    // S[query][key]
          const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ado, bv, zero, 0, 0, 0);   // dP[query][key]
          const f32x4 l4 = *(const f32x4*)(lq + qt * 16 + 4 * g);
          const f32x4 d4 = *(const f32x4*)(dl + wave * TM + qt * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qt * 16 + 4 * g + r;   // a query sees this key iff both sit in the same (non-empty) window
            const float p = (qi >= klo_ && qi < khi_) ? __expf(sc[r] * 0.25f - l4[r]) : 0.f;
            pv[qt][r] = p;
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
                                                         pack_tiles(pv[2 * u], pv[2 * u + 1]), accv, 0, 0, 0);
        }
      }
      gk[kt] = acck;
      gv[kt] = accv;
    }
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
  __syncthreads();
  tile_store_rows<3 * E>(qs, LDQ, dqkv_out, tm->rows, 0, 0);
  f32x4 gx[2][4];
  zero_acc(gx);
  tile_gemm<2, 12>(wqkvt, 2 * wave, qs, LDQ, gx);   // dx = Wqkv^T dqkv (+ dz1: the residual)
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {   // over dO: its last readers were the attention passes, before the barrier above
    const int n = 16 * (2 * wave + nb) + 4 * g;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) *(u32x2*)(xs + (mb * 16 + c) * LDX + n) = pack4(gx[nb][mb] + dz[nb][mb]);
  }
  __syncthreads();
  tile_store_rows<E>(xs, LDX, dx, tm->rows, 0, 0);
}

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
__global__ void __launch_bounds__(kPlanThreads)
window_tile_plan_kernel(const int32_t* __restrict__ win_len, int64_t num_windows, int32_t* __restrict__ win_tile,
                        int32_t* __restrict__ win_base, int32_t* __restrict__ num_tiles) {
  __shared__ int cnt[kPlanThreads];
  const int64_t chunk = max((int64_t)128, (num_windows + kPlanThreads - 1) / kPlanThreads);
  const int64_t w0 = (int64_t)threadIdx.x * chunk, w1 = min(num_windows, w0 + chunk);
  int tiles = 0, fill = 0;
  for (int64_t w = w0; w < w1; ++w) {
    const int n = win_len[w];
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
  bool open = false;
  for (int64_t w = w0; w < w1; ++w) {
    const int n = win_len[w];
    if (n <= 0) {
      win_tile[w] = -1;
      win_base[w] = 0;
      continue;
    }
    if (fill + n > TM) {
      ++tile;
      fill = 0;
    }
    open = true;
    win_tile[w] = tile;
    win_base[w] = fill;
    fill += n;
  }
  (void)open;
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
  u32x4 gv, xv[XP];
  float dbacc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dbacc[j] = 0.f;
  auto fetch = [&](int s) {
    const int64_t r = t0 + (int64_t)s * 32 + grow;
    gv = u32x4{0u, 0u, 0u, 0u};
    if (r < t1) gv = *(const u32x4*)(gp + r * ldg + gpc * 8);
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int i = threadIdx.x + j * kThreads, row = i / (K / 8), pc = i % (K / 8);
      const int64_t rr = t0 + (int64_t)s * 32 + row;
      xv[j] = u32x4{0u, 0u, 0u, 0u};
      if (rr < t1) {
        xv[j] = *(const u32x4*)(xp + rr * K + pc * 8);
        if (ap) {
          const u32x4 a = *(const u32x4*)(ap + rr * K + pc * 8);
          xv[j].x = add_bf16x2(xv[j].x, a.x);
          xv[j].y = add_bf16x2(xv[j].y, a.y);
          xv[j].z = add_bf16x2(xv[j].z, a.z);
          xv[j].w = add_bf16x2(xv[j].w, a.w);
        }
      }
    }
  };
  auto stash = [&](int buf) {
    *(u32x4*)(gs + (buf * 32 + grow) * LDG + gpc * 8) = gv;
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int i = threadIdx.x + j * kThreads, row = i / (K / 8), pc = i % (K / 8);
      *(u32x4*)(xs + (buf * 32 + row) * LDK + pc * 8) = xv[j];
    }
    dbacc[0] += __uint_as_float(gv.x << 16);
    dbacc[1] += __uint_as_float(gv.x & 0xffff0000u);
    dbacc[2] += __uint_as_float(gv.y << 16);
    dbacc[3] += __uint_as_float(gv.y & 0xffff0000u);
    dbacc[4] += __uint_as_float(gv.z << 16);
    dbacc[5] += __uint_as_float(gv.z & 0xffff0000u);
    dbacc[6] += __uint_as_float(gv.w << 16);
    dbacc[7] += __uint_as_float(gv.w & 0xffff0000u);
  };
  constexpr int KB = K / 64;      // 16-column blocks of X per wave
  f32x4 acc[4][KB];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc[nb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (steps > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) fetch(s + 1);
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
    if (s + 1 < steps) stash(buf ^ 1);
    __syncthreads();
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
  hipLaunchKernelGGL(window_tile_plan_kernel, dim3(1), dim3(kPlanThreads), 0, stream, win_len, num_windows, win_tile,
                     win_base, num_tiles);
  hipLaunchKernelGGL(window_tile_fill_kernel, dim3((unsigned)ococc_cdiv(num_windows * 64, 256)), dim3(256), 0, stream,
                     win_len, win_off, tok, num_windows, win_tile, win_base, tile_rows, tile_span);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

#define OCOCC_BLOCK_DIMS_OK(d_model, heads, ffn) \
  OCOCC_REQUIRE((d_model) == E && (heads) == NH && (ffn) == FF, "the fused encoder-layer kernels are built for d_model 128, 8 heads, feed-forward 256")

extern "C" int ococc_window_attn_block_fwd_bf16(const uint16_t* x, const uint16_t* pos, const int32_t* tile_rows,
                                                const int32_t* tile_span, int64_t num_tiles, int32_t d_model,
                                                int32_t num_heads, const uint16_t* wqkv_frag, const float* bqkv,
                                                const uint16_t* wo_frag, const float* bo, const float* ln_weight,
                                                const float* ln_bias, float eps, uint16_t* y,
                                                ococc_stream_t stream) {
  OCOCC_BLOCK_DIMS_OK(d_model, num_heads, FF);
  OCOCC_REQUIRE(num_tiles >= 0, "bad sizes");
  if (num_tiles == 0) return OCOCC_OK;
  OCOCC_REQUIRE(x && tile_rows && tile_span && wqkv_frag && bqkv && wo_frag && bo && ln_weight && ln_bias && y,
                "null pointer");
  OCOCC_REQUIRE(aligned16(x) && aligned16(pos) && aligned16(y) && aligned16(wqkv_frag) && aligned16(wo_frag) &&
                    aligned16(bqkv) && aligned16(bo) && aligned16(ln_weight) && aligned16(ln_bias),
                "buffers must be 16-byte aligned");
  OCOCC_HIP(hipFuncSetAttribute((const void*)window_attn_block_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kAttnLds));
  hipLaunchKernelGGL(window_attn_block_fwd_kernel, dim3((unsigned)num_tiles), dim3(kThreads), kAttnLds,
                     (hipStream_t)stream, x, pos, tile_rows, tile_span, wqkv_frag, bqkv, wo_frag, bo, ln_weight,
                     ln_bias, eps, y);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_window_attn_block_bwd_bf16(const uint16_t* x, const uint16_t* pos, const uint16_t* dy,
                                                const int32_t* tile_rows, const int32_t* tile_span,
                                                int64_t num_tiles, int32_t d_model, int32_t num_heads,
                                                const uint16_t* wqkv_frag, const float* bqkv, const uint16_t* wo_frag,
                                                const float* bo, const float* ln_weight, float eps,
                                                const uint16_t* wo_t_frag, const uint16_t* wqkv_t_frag, uint16_t* dx,
                                                uint16_t* dqkv, uint16_t* dz, uint16_t* attn_out, float* ln_partial,
                                                ococc_stream_t stream) {
  OCOCC_BLOCK_DIMS_OK(d_model, num_heads, FF);
  OCOCC_REQUIRE(num_tiles >= 0, "bad sizes");
  if (num_tiles == 0) return OCOCC_OK;
  OCOCC_REQUIRE(x && dy && tile_rows && tile_span && wqkv_frag && bqkv && wo_frag && bo && ln_weight && wo_t_frag &&
                    wqkv_t_frag && dx && dqkv && dz && attn_out && ln_partial,
                "null pointer");
  OCOCC_REQUIRE(aligned16(x) && aligned16(pos) && aligned16(dy) && aligned16(dx) && aligned16(dqkv) && aligned16(dz) &&
                    aligned16(attn_out) && aligned16(wqkv_frag) && aligned16(wo_frag) && aligned16(wo_t_frag) &&
                    aligned16(wqkv_t_frag) && aligned16(bqkv) && aligned16(bo) && aligned16(ln_weight),
                "buffers must be 16-byte aligned");
  OCOCC_HIP(hipFuncSetAttribute((const void*)window_attn_block_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kAttnLds));
  hipLaunchKernelGGL(window_attn_block_bwd_kernel, dim3((unsigned)num_tiles), dim3(kThreads), kAttnLds,
                     (hipStream_t)stream, x, pos, dy, tile_rows, tile_span, wqkv_frag, bqkv, wo_frag, bo, ln_weight, eps,
                     wo_t_frag, wqkv_t_frag, dx, dqkv, dz, attn_out, ln_partial);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
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
  const unsigned grid = (unsigned)ococc_cdiv(num_tokens, TM);
#define OCOCC_FFN_FWD(A)                                                                                          \
  do {                                                                                                            \
    OCOCC_HIP(hipFuncSetAttribute((const void*)token_ffn_block_fwd_kernel<A>,                                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kFfnLds));                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(token_ffn_block_fwd_kernel<A>), dim3(grid), dim3(kThreads), kFfnLds,       \
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
  const unsigned grid = (unsigned)ococc_cdiv(num_tokens, TM);
#define OCOCC_FFN_BWD(A)                                                                                           \
  do {                                                                                                             \
    OCOCC_HIP(hipFuncSetAttribute((const void*)token_ffn_block_bwd_kernel<A>,                                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kFfnLds));                           \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(token_ffn_block_bwd_kernel<A>), dim3(grid), dim3(kThreads), kFfnLds,        \
                       (hipStream_t)stream, x, dy, num_tokens, w1_frag, b1, w2_frag, b2, ln_weight, eps, w2_t_frag, \
                       w1_t_frag, dx, act_out, dh, dz, ln_partial);                                                \
  } while (0)
  if (act == 0) OCOCC_FFN_BWD(0);
  else OCOCC_FFN_BWD(1);
#undef OCOCC_FFN_BWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
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
