// A9  the attention core of the temporal transformer in one launch per direction.
//
// Reference: SimpleEncoderLayer.forward -> nn.MultiheadAttention (mmdet3d/models/occ/layers.py:35-87), called by
// OccBBoxHead.transformer_forward_fixed_length / _various_length (ococc_bbox_head.py:849-995) with q = k = x + pos, v = x,
// 4 heads of 384 channels over the L <= 200 frames of a tracklet, a boolean upper-triangular attn_mask (get_future_mask,
// :1034-1043), an optional key_padding_mask, dropout 0.1 on the probabilities.  There the core is
//   scores = (q / sqrt(D)) k^T  -> masked_fill -> softmax -> dropout -> @ v
// i.e. bmm, fill, softmax, dropout, bmm (and their five backward counterparts) on [B H, L, S] tensors: at L = 32 ten
// launches of a few microseconds of work each per layer.  Here: one workgroup per (tracklet, head, 16 query rows) computes
// its rows of the scores against K staged through LDS 64 keys at a time, the masked softmax, the dropout mask (a
// counter-based hash of (seed, element): the backward pass regenerates it, nothing is stored) and the product with V; the
// probabilities (before dropout) are kept for the backward pass.  Backward, one launch: workgroups of role A own 16 query
// rows (dQ), workgroups of role B 16 keys (dK, dV); both recompute d(probabilities) = dO V^T from their side, and the
// softmax backward's row term is the flash-attention identity  sum_s P dP = dO . O.  No atomics: every output element has
// one adder, the sums run in a fixed order.  f32 throughout (the reference's precision for this block); the work is a few
// MFLOP per tracklet -- the point is the launch count and the five [B H, L, S] round trips, not the matrix pipe.
#include "common.hpp"

namespace {

constexpr int kAT = 256;     // threads per workgroup
constexpr int kAR = 16;      // query rows (role A, forward) or keys (role B) a workgroup owns
constexpr int kACH = 64;     // rows of K / V / Q / dO staged per chunk
constexpr int kAMaxS = 256;  // longest sequence (the reference's PositionalEncoding: max_len = 200)
constexpr int kASP = kAMaxS + 4;

struct AttnArgs {
  const float* q;   // token-major rows t = l * B + b, head h at columns h * D .. h * D + D - 1
  const float* k;
  const float* v;
  int64_t ldq, ldk, ldv;
  const uint8_t* attn_mask;   // [L, S], 1 = not allowed, or null
  const uint8_t* key_pad;     // [B, S], 1 = padding, or null
  int32_t B, H, L, S, D;
  float scale;                // 1 / sqrt(D)
  uint32_t drop_thr;          // an element is kept when its 24-bit hash >= drop_thr (0: no dropout)
  float drop_scale;           // 1 / (1 - p)
  uint32_t seed_lo, seed_hi;
  const uint64_t* seed_dev;   // non-null: the seed is read from device memory (a captured graph draws a new one per replay)
  float* p;                   // [B H, L, S] probabilities before dropout (forward: written; backward: read)
  float* o;                   // [L B, ldo] context rows (forward: written; backward: read)
  int64_t ldo;
  const float* d_o;           // backward
  int64_t lddo;
  float* dq;
  float* dk;
  float* dv;
  int64_t lddq, lddk, lddv;
  int32_t q_tiles;            // backward: workgroups [0, q_tiles) of grid.y are role A, the rest role B
};

struct AttnSeed {
  uint32_t lo, hi;
};
__device__ __forceinline__ AttnSeed attn_seed(const AttnArgs& a) {
  if (a.drop_thr && a.seed_dev) {
    const uint64_t s = *a.seed_dev;
    return AttnSeed{(uint32_t)s, (uint32_t)(s >> 32)};
  }
  return AttnSeed{a.seed_lo, a.seed_hi};
}
__device__ __forceinline__ bool attn_keep(const AttnArgs& a, const AttnSeed& sd, int bh, int l, int s) {
  if (!a.drop_thr) return true;
  uint32_t h = (uint32_t)((bh * a.L + l) * a.S + s) ^ sd.lo;
  h *= 0x9E3779B1u;
  h ^= h >> 16;
  h = (h + sd.hi) * 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return (h >> 8) >= a.drop_thr;
}
__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
__device__ __forceinline__ float sum16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ float max16(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64));
  v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
// rows [r0, r0 + nrows) of a token-major matrix (row l of tracklet b, head h) -> LDS tile [nrows][LD], zeros past `limit`
__device__ __forceinline__ void stage_rows(float* dst, int LD, const float* src, int64_t ld, int r0, int nrows, int limit, int B,
                                           int b, int col0, int D4, float mul) {
  for (int i = threadIdx.x; i < nrows * D4; i += kAT) {
    const int r = i / D4, c4 = i - r * D4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r0 + r < limit) v = *(const f32x4*)(src + ((int64_t)(r0 + r) * B + b) * ld + col0 + c4 * 4);
    *(f32x4*)(dst + r * LD + c4 * 4) = v * mul;
  }
}
// out[r][j] = tile[r] . chunk[j] for this thread's r = tid / 16 and j = tid % 16 + 16 c, c = 0..3 (chunk rows of LDS)
__device__ __forceinline__ void dots_16x64(const float* tile, const float* chunk, int LD, int D4, float (&acc)[4]) {
  const int r = threadIdx.x >> 4, j0 = threadIdx.x & 15;
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = 0.f;
  for (int d4 = 0; d4 < D4; ++d4) {
    const f32x4 t = *(const f32x4*)(tile + r * LD + d4 * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] += dot4(t, *(const f32x4*)(chunk + (j0 + 16 * c) * LD + d4 * 4));
  }
}
constexpr int kANJ = 6;   // float4 column groups per thread: D <= 16 * 4 * kANJ = 384
// acc[jj] += sum over the chunk's rows j < n of coef[r][j] * chunk[j][columns of this thread]
__device__ __forceinline__ void rows_times_chunk(const float* coef, int coef_ld, const float* chunk, int LD, int D4, int n,
                                                 f32x4 (&acc)[kANJ]) {
  const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15;
  for (int j = 0; j < n; ++j) {
    const float w = coef[r * coef_ld + j];
#pragma unroll
    for (int jj = 0; jj < kANJ; ++jj) {
      const int c4 = c0 + 16 * jj;
      if (c4 < D4) acc[jj] += *(const f32x4*)(chunk + j * LD + c4 * 4) * w;
    }
  }
}
__device__ __forceinline__ void store_rows(float* dst, int64_t ld, int row0, int limit, int B, int b, int col0, int D4,
                                           const f32x4 (&acc)[kANJ], float mul) {
  const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15;
  if (row0 + r >= limit) return;
#pragma unroll
  for (int jj = 0; jj < kANJ; ++jj) {
    const int c4 = c0 + 16 * jj;
    if (c4 < D4) *(f32x4*)(dst + ((int64_t)(row0 + r) * B + b) * ld + col0 + c4 * 4) = acc[jj] * mul;
  }
}

__global__ void __launch_bounds__(kAT) attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LD = a.D + 4, D4 = a.D >> 2;
  float* sq = sm;                  // [kAR][LD]   the query rows, scaled
  float* sc = sq + kAR * LD;       // [kACH][LD]  a chunk of K, then of V
  float* ss = sc + kACH * LD;      // [kAR][kASP] scores -> probabilities after dropout
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int l0 = blockIdx.y * kAR;
  const int r = threadIdx.x >> 4, j0 = threadIdx.x & 15;
  const AttnSeed seed = attn_seed(a);
  stage_rows(sq, LD, a.q, a.ldq, l0, kAR, a.L, a.B, b, h * a.D, D4, a.scale);
  for (int s0 = 0; s0 < a.S; s0 += kACH) {
    __syncthreads();
    stage_rows(sc, LD, a.k, a.ldk, s0, kACH, a.S, a.B, b, h * a.D, D4, 1.f);
    __syncthreads();
    float acc[4];
    dots_16x64(sq, sc, LD, D4, acc);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (s0 + j0 + 16 * c < a.S) ss[r * kASP + s0 + j0 + 16 * c] = acc[c];
  }
  __syncthreads();
  {   // masked softmax of row r over its keys j0, j0 + 16, ... (16 lanes per row), then dropout
    const int l = l0 + r;
    const bool live = l < a.L;
    float m = -INFINITY;
    for (int s = j0; s < a.S; s += 16) {
      float x = ss[r * kASP + s];
      if (live && ((a.attn_mask && a.attn_mask[(int64_t)l * a.S + s]) || (a.key_pad && a.key_pad[(int64_t)b * a.S + s]))) x = -INFINITY;
      ss[r * kASP + s] = x;
      m = fmaxf(m, x);
    }
    m = max16(m);
    float sum = 0.f;
    for (int s = j0; s < a.S; s += 16) {
      const float e = __expf(ss[r * kASP + s] - m);   // (a row without any allowed key: exp(-inf + inf) = NaN, as torch.softmax)
      ss[r * kASP + s] = e;
      sum += e;
    }
    const float inv = 1.f / sum16(sum);
    for (int s = j0; s < a.S; s += 16) {
      const float pr = ss[r * kASP + s] * inv;
      if (live) a.p[((int64_t)bh * a.L + l) * a.S + s] = pr;
      ss[r * kASP + s] = (live && attn_keep(a, seed, bh, l, s)) ? pr * a.drop_scale : 0.f;
    }
  }
  f32x4 o[kANJ];
#pragma unroll
  for (int jj = 0; jj < kANJ; ++jj) o[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < a.S; s0 += kACH) {
    __syncthreads();
    stage_rows(sc, LD, a.v, a.ldv, s0, kACH, a.S, a.B, b, h * a.D, D4, 1.f);
    __syncthreads();
    const int n = a.S - s0 < kACH ? a.S - s0 : kACH;
    rows_times_chunk(ss + s0, kASP, sc, LD, D4, n, o);
  }
  store_rows(a.o, a.ldo, l0, a.L, a.B, b, h * a.D, D4, o, 1.f);
}

// delta[l] = dO[l] . O[l] for rows [l0, l0 + n): one wave per row in turns
__device__ __forceinline__ void row_deltas(const AttnArgs& a, int b, int h, int l0, int n, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < n; i += kAT / 64) {
    const int l = l0 + i;
    float s = 0.f;
    if (l < a.L) {
      const float* po = a.o + ((int64_t)l * a.B + b) * a.ldo + h * a.D;
      const float* pd = a.d_o + ((int64_t)l * a.B + b) * a.lddo + h * a.D;
      for (int d = lane; d < a.D; d += 64) s += po[d] * pd[d];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) out[i] = s;
  }
}

__global__ void __launch_bounds__(kAT) attn_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LD = a.D + 4, D4 = a.D >> 2;
  float* st = sm;                   // [kAR][LD]   role A: the dO rows; role B: the V rows of the keys
  float* sc = st + kAR * LD;        // [kACH][LD]  a chunk of V / K (A) or of dO / Q (B)
  float* sd = sc + kACH * LD;       // [kAR][kASP] role A: dS [query][key]; role B: dS^T [key][query]
  float* sp = sd + kAR * kASP;      // [kAR][kACH + 4] role B: probabilities after dropout, transposed, of the chunk
  float* sdel = sp + kAR * (kACH + 4);   // [kAMaxS] row terms of the softmax backward
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int r = threadIdx.x >> 4, j0 = threadIdx.x & 15;
  const AttnSeed seed = attn_seed(a);
  f32x4 acc[kANJ], acc2[kANJ];
#pragma unroll
  for (int jj = 0; jj < kANJ; ++jj) acc[jj] = acc2[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  if ((int)blockIdx.y < a.q_tiles) {
    // ---- role A: query rows l0 .. l0 + 15 -> dQ ----
    const int l0 = blockIdx.y * kAR, l = l0 + r;
    stage_rows(st, LD, a.d_o, a.lddo, l0, kAR, a.L, a.B, b, h * a.D, D4, 1.f);
    row_deltas(a, b, h, l0, kAR, sdel);
    for (int s0 = 0; s0 < a.S; s0 += kACH) {
      __syncthreads();
      stage_rows(sc, LD, a.v, a.ldv, s0, kACH, a.S, a.B, b, h * a.D, D4, 1.f);
      __syncthreads();
      float dp[4];
      dots_16x64(st, sc, LD, D4, dp);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int s = s0 + j0 + 16 * c;
        if (s < a.S) {
          float ds = 0.f;
          if (l < a.L) {
            const float pr = a.p[((int64_t)bh * a.L + l) * a.S + s];
            const float dpr = attn_keep(a, seed, bh, l, s) ? dp[c] * a.drop_scale : 0.f;
            ds = pr * (dpr - sdel[r]);
          }
          sd[r * kASP + s] = ds;
        }
      }
    }
    for (int s0 = 0; s0 < a.S; s0 += kACH) {
      __syncthreads();
      stage_rows(sc, LD, a.k, a.ldk, s0, kACH, a.S, a.B, b, h * a.D, D4, 1.f);
      __syncthreads();
      const int n = a.S - s0 < kACH ? a.S - s0 : kACH;
      rows_times_chunk(sd + s0, kASP, sc, LD, D4, n, acc);
    }
    store_rows(a.dq, a.lddq, l0, a.L, a.B, b, h * a.D, D4, acc, a.scale);
    return;
  }
  // ---- role B: keys s0 .. s0 + 15 -> dK, dV ----
  const int s0 = ((int)blockIdx.y - a.q_tiles) * kAR, s = s0 + r;
  stage_rows(st, LD, a.v, a.ldv, s0, kAR, a.S, a.B, b, h * a.D, D4, 1.f);
  row_deltas(a, b, h, 0, a.L, sdel);
  for (int l0 = 0; l0 < a.L; l0 += kACH) {
    __syncthreads();
    stage_rows(sc, LD, a.d_o, a.lddo, l0, kACH, a.L, a.B, b, h * a.D, D4, 1.f);
    __syncthreads();
    float dp[4];
    dots_16x64(st, sc, LD, D4, dp);   // dp[c] = V[s] . dO[l0 + j0 + 16 c]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = j0 + 16 * c, l = l0 + j;
      float pd = 0.f, ds = 0.f;
      if (l < a.L && s < a.S) {
        const float pr = a.p[((int64_t)bh * a.L + l) * a.S + s];
        const bool keep = attn_keep(a, seed, bh, l, s);
        pd = keep ? pr * a.drop_scale : 0.f;
        ds = pr * ((keep ? dp[c] * a.drop_scale : 0.f) - sdel[l]);
      }
      sp[r * (kACH + 4) + j] = pd;
      if (l < a.L) sd[r * kASP + l] = ds;
    }
    __syncthreads();
    const int n = a.L - l0 < kACH ? a.L - l0 : kACH;
    rows_times_chunk(sp, kACH + 4, sc, LD, D4, n, acc);   // dV += Pd^T dO
  }
  for (int l0 = 0; l0 < a.L; l0 += kACH) {
    __syncthreads();
    stage_rows(sc, LD, a.q, a.ldq, l0, kACH, a.L, a.B, b, h * a.D, D4, 1.f);
    __syncthreads();
    const int n = a.L - l0 < kACH ? a.L - l0 : kACH;
    rows_times_chunk(sd + l0, kASP, sc, LD, D4, n, acc2);   // dK += dS^T q
  }
  store_rows(a.dv, a.lddv, s0, a.S, a.B, b, h * a.D, D4, acc, 1.f);
  store_rows(a.dk, a.lddk, s0, a.S, a.B, b, h * a.D, D4, acc2, a.scale);
}

int check_attn(const AttnArgs& a) {
  OCOCC_REQUIRE(a.B >= 1 && a.H >= 1 && a.L >= 1 && a.S >= 1, "empty problem");
  if (a.S > kAMaxS || a.L > kAMaxS || a.D > 64 * kANJ || a.D % 4 != 0)
    return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "sequences up to 256 tokens, head width a multiple of 4 up to 384");
  OCOCC_REQUIRE(a.q && a.k && a.v && a.p && a.o, "null pointer");
  OCOCC_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0 &&
                    (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.o) & 15) == 0,
                "rows must be 16-byte aligned");
  OCOCC_REQUIRE((int64_t)a.B * a.H * a.L * a.S < (1ll << 31), "too many probabilities for the 32-bit dropout counter");
  return OCOCC_OK;
}

}  // namespace

extern "C" int ococc_temporal_attention_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                                int64_t ldv, const uint8_t* attn_mask, const uint8_t* key_padding_mask,
                                                int32_t batch, int32_t heads, int32_t L, int32_t S, int32_t head_dim,
                                                float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                                float* probs, float* out, int64_t ldo, ococc_stream_t stream_) {
  OCOCC_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout probability in [0, 1)");
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.attn_mask = attn_mask; a.key_pad = key_padding_mask;
  a.B = batch; a.H = heads; a.L = L; a.S = S; a.D = head_dim;
  a.scale = scale;
  a.drop_thr = (uint32_t)(dropout_p * 16777216.f);
  a.drop_scale = 1.f / (1.f - dropout_p);
  a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  a.seed_dev = seed_dev;
  a.p = probs; a.o = out; a.ldo = ldo;
  if (int rc = check_attn(a)) return rc;
  const int lds = ((kAR + kACH) * (head_dim + 4) + kAR * kASP) * 4;
  OCOCC_HIP(hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(batch * heads, (L + kAR - 1) / kAR), dim3(kAT), lds, (hipStream_t)stream_, a);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_temporal_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                                int64_t ldv, int32_t batch, int32_t heads, int32_t L, int32_t S,
                                                int32_t head_dim, float scale, float dropout_p, uint64_t seed,
                                                const uint64_t* seed_dev, const float* probs, const float* out, int64_t ldo, const float* d_out,
                                                int64_t lddo, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                                                int64_t lddv, ococc_stream_t stream_) {
  OCOCC_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout probability in [0, 1)");
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.B = batch; a.H = heads; a.L = L; a.S = S; a.D = head_dim;
  a.scale = scale;
  a.drop_thr = (uint32_t)(dropout_p * 16777216.f);
  a.drop_scale = 1.f / (1.f - dropout_p);
  a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  a.seed_dev = seed_dev;
  a.p = const_cast<float*>(probs); a.o = const_cast<float*>(out); a.ldo = ldo;
  a.d_o = d_out; a.lddo = lddo;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  if (int rc = check_attn(a)) return rc;
  OCOCC_REQUIRE(d_out && dq && dk && dv, "null pointer");
  OCOCC_REQUIRE(lddo % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 &&
                    (((uintptr_t)d_out | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                "rows must be 16-byte aligned");
  a.q_tiles = (L + kAR - 1) / kAR;
  const int k_tiles = (S + kAR - 1) / kAR;
  const int lds = ((kAR + kACH) * (head_dim + 4) + kAR * kASP + kAR * (kACH + 4) + kAMaxS) * 4;
  OCOCC_HIP(hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(batch * heads, a.q_tiles + k_tiles), dim3(kAT), lds, (hipStream_t)stream_, a);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
