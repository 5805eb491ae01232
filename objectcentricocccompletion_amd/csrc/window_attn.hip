// B7 window attention core: softmax(Q K^T / sqrt(d) + key mask) V for padded windows of the SST
// blocks (mmdet3d/models/sst/sst_basic_block_v2.py:41-75 runs nn.MultiheadAttention on
// [num_windows, max_tokens, C] tensors built by flat2window_v2).  d_head = 16 (d_model 128,
// 8 heads), max_tokens <= 160; tokens of a window occupy its first key_len slots, so the
// key_padding_mask is a length.
//
// Forward and backward are MFMA kernels with one workgroup per (window, head group); see the
// comments on the kernels.  A lane always ends with 4 consecutive channels of one token: 8-byte
// stores.  No atomics: deterministic.
#include "common.hpp"

namespace {

constexpr int kD = 16;        // head dim
constexpr int kMaxTiles = 10; // max_tokens <= 160
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ s16x4 ld4(const uint16_t* p) { return *(const s16x4*)p; }

// ds_read_b64_tr_b16: the lanes of a 16-lane group address a 4-row x 16-column bf16 block (lane li:
// row li>>2, columns 4(li&3)..+3) and lane li receives column li of the 4 rows -- i.e. an MFMA
// operand whose k index runs over LDS rows, read from a row-major tile without transposing it.
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

__device__ __forceinline__ void st4(uint16_t* p, const f32x4 o) {
  u32x2 v;
  v.x = (uint32_t)ococc_f32_to_bf16(o[0]) | ((uint32_t)ococc_f32_to_bf16(o[1]) << 16);
  v.y = (uint32_t)ococc_f32_to_bf16(o[2]) | ((uint32_t)ococc_f32_to_bf16(o[3]) << 16);
  *(u32x2*)p = v;
}

// Forward.  One workgroup per (window, group of HG heads); K and V of the window are staged once with
// 16-byte loads into row-major LDS rows [token][K heads | V heads | pad]; wave w runs heads w, w+4.
//   S^T = K Q^T  with v_mfma_f32_16x16x16_bf16 (k = d_head = 16): keys on the MFMA rows, queries on
//   the lanes, so a lane owns one query column and the softmax over keys is a register reduction
//   plus two wave shuffles;
//   O^T = V^T P^T with v_mfma_f32_16x16x32_bf16: the probabilities are already in the B-operand
//   registers (two 16-key score tiles form one 32-deep k-step in a permuted key order) and V^T comes
//   from the row-major LDS tile through transposing reads in the same order.
template <int MT>  // compile-time bound on the number of 16-token tiles (2, 4, 7 or 10)
__global__ void __launch_bounds__(256)
window_attn_fwd_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                       const uint16_t* __restrict__ v, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                       const int32_t* __restrict__ key_len, int T, int H, int HG, float scale,
                       uint16_t* __restrict__ out, int64_t out_stride, float* __restrict__ lse,
                       const int32_t* __restrict__ tok) {
  // tok == nullptr: padded layout, token t of window w is row w*T + t of q/k/v/out.
  // tok != nullptr: flat-token layout, row tok[w*T + t] (-1 = padding slot): the kernel gathers the
  // window's tokens itself and writes each output row exactly once; no padded copy exists in HBM.
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint16_t* rows = (uint16_t*)smem_raw;
  const int w = blockIdx.x, hg0 = blockIdx.y * HG;
  auto rowof = [&](int t) -> int64_t { return tok ? (int64_t)tok[(int64_t)w * T + t] : (int64_t)w * T + t; };
  const int tiles = (T + 15) >> 4;
  const int TPE = ((tiles + 1) >> 1) * 32;  // PV consumes key tiles in pairs: rows up to TPE are zero filled
  const int LDR = 2 * HG * kD + 16;         // elements per LDS row (32 B pad)
  const int PP = 4 * HG;                    // 16-byte pieces per row (K then V)
  const int len = key_len[w];
  for (int i = threadIdx.x; i < TPE * PP; i += 256) {
    const int t = i / PP, pc = i % PP;
    u32x4 val = {0u, 0u, 0u, 0u};
    const int64_t r = t < T ? rowof(t) : -1;
    if (r >= 0) {
      const uint16_t* src = pc < 2 * HG ? k + r * k_stride + hg0 * kD + pc * 8
                                        : v + r * v_stride + hg0 * kD + (pc - 2 * HG) * 8;
      val = *(const u32x4*)src;
    }
    *(u32x4*)(rows + t * LDR + pc * 8) = val;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int q_ = c >> 2, p_ = c & 3;
  for (int hh = wave; hh < HG; hh += 4) {
    const int h = hg0 + hh;
    const uint16_t* ks = rows + hh * kD;
    const uint16_t* vs = rows + (HG + hh) * kD;
    for (int qt = 0; qt < tiles; ++qt) {
      const int qi = qt * 16 + c;  // this lane's query
      const int64_t qrow = qi < T ? rowof(qi) : -1;
      s16x4 bq = {0, 0, 0, 0};
      if (qrow >= 0) bq = ld4(q + qrow * q_stride + h * kD + 4 * g);
      f32x4 s[MT + 1];  // one spare tile so that an odd MT pairs its last tile with zeros
      s[MT] = f32x4{0.f, 0.f, 0.f, 0.f};
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < MT; ++kt) {
        s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt < tiles) {
          const s16x4 ak = ld4(ks + (kt * 16 + c) * LDR + 4 * g);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ak, bq, acc, 0, 0, 0);  // rows: keys 4g+r, col: query c
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            const float val = key < len ? acc[r] * scale : -INFINITY;
            s[kt][r] = val;
            m = fmaxf(m, val);
          }
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < MT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = (kt < tiles && m > -INFINITY) ? __expf(s[kt][r] - m) : 0.f;
          s[kt][r] = e;
          sum += e;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = sum > 0.f ? 1.f / sum : 0.f;
      // O^T[d][query] = sum_key V^T[d][key] P[key][query]; k-step u covers key tiles 2u, 2u+1 in the
      // permuted order  position 8g+j -> key (j<4 ? 32u + 4g + j : 32u + 16 + 4g + j - 4)
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < (MT + 1) / 2; ++u) {
        if (2 * u < tiles) {
          const bf16x8 pb = pack_tiles(s[2 * u] * inv, s[2 * u + 1] * inv);  // zeros when tile 2u+1 does not exist
          const uint16_t* a0 = vs + (32 * u + 4 * g + q_) * LDR + 4 * p_;
          const bf16x8 va = tr_pair(a0, a0 + 16 * LDR);
          o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o, 0, 0, 0);  // rows: d = 4g+r, col: query c
        }
      }
      if (qi < T) {
        if (qrow >= 0) st4(out + qrow * out_stride + h * kD + 4 * g, o);
        if (lse && g == 0) lse[((int64_t)w * H + h) * T + qi] = sum > 0.f ? m + __logf(sum) : 0.f;
      }
    }
  }
}

// Backward, same workgroup shape.  Q, K, V and dO of the window sit row-major in LDS; delta =
// rowsum(dO * O) is formed while staging.  Both passes recompute the probabilities from the saved
// log-sum-exp with 16x16x16 MFMAs:
//   pass 1 (lanes own queries, tiles S^T / dP^T):  dQ^T = K^T dS^T           (K^T by transposing reads)
//   pass 2 (lanes own keys,    tiles S   / dP  ):  dK^T = Q^T dS, dV^T = dO^T P  (Q^T, dO^T likewise)
// so every contraction is a 16x16x32 MFMA whose B operand is already in registers; nothing is
// transposed through LDS and there are no atomics (deterministic).
template <int MT>
__global__ void __launch_bounds__(256)
window_attn_bwd_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                       const uint16_t* __restrict__ v, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                       const uint16_t* __restrict__ out, const uint16_t* __restrict__ dout, int64_t o_stride,
                       const float* __restrict__ lse, const int32_t* __restrict__ key_len, int T, int H, int HG,
                       float scale, uint16_t* __restrict__ dq, uint16_t* __restrict__ dk,
                       uint16_t* __restrict__ dv, int64_t dq_stride, int64_t dk_stride, int64_t dv_stride,
                       const int32_t* __restrict__ tok) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int w = blockIdx.x, hg0 = blockIdx.y * HG;
  auto rowof = [&](int t) -> int64_t { return tok ? (int64_t)tok[(int64_t)w * T + t] : (int64_t)w * T + t; };
  const int tiles = (T + 15) >> 4;
  const int TPE = ((tiles + 1) >> 1) * 32;
  const int LDR = 4 * HG * kD + 16;  // Q | K | V | dO, 32 B pad
  const int PP = 8 * HG;             // 16-byte pieces per row
  uint16_t* rows = (uint16_t*)smem_raw;
  float* lse_s = (float*)(smem_raw + (size_t)TPE * LDR * 2);
  float* delta_s = lse_s + HG * TPE;
  const int len = key_len[w];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < TPE * PP; i += 256) {  // TPE*PP is a multiple of 256: uniform trip count
    const int t = i / PP, pc = i % PP;
    const int sel = pc / (2 * HG), sub = pc % (2 * HG);  // matrix, piece inside it
    u32x4 val = {0u, 0u, 0u, 0u};
    float part = 0.f;
    const int64_t r = t < T ? rowof(t) : -1;
    if (r >= 0) {
      const int col = hg0 * kD + sub * 8;
      const uint16_t* src = sel == 0 ? q + r * q_stride + col
                          : sel == 1 ? k + r * k_stride + col
                          : sel == 2 ? v + r * v_stride + col
                                     : dout + r * o_stride + col;
      val = *(const u32x4*)src;
      if (sel == 3) {
        const u32x4 ov = *(const u32x4*)(out + r * o_stride + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          part += ococc_bf16_to_f32((uint16_t)(val[j] & 0xffffu)) * ococc_bf16_to_f32((uint16_t)(ov[j] & 0xffffu));
          part += ococc_bf16_to_f32((uint16_t)(val[j] >> 16)) * ococc_bf16_to_f32((uint16_t)(ov[j] >> 16));
        }
      }
    }
    *(u32x4*)(rows + t * LDR + pc * 8) = val;
    part += __shfl_xor(part, 1, 64);  // the two 8-channel halves of a head are adjacent pieces
    if (sel == 3 && (sub & 1) == 0) {
      const int hh = sub >> 1;
      delta_s[hh * TPE + t] = part;
      lse_s[hh * TPE + t] = t < T ? lse[((int64_t)w * H + hg0 + hh) * T + t] : 0.f;
    }
  }
  __syncthreads();
  const int c = lane & 15, g = lane >> 4;
  const int q_ = c >> 2, p_ = c & 3;
  for (int hh = wave; hh < HG; hh += 4) {
    const int h = hg0 + hh;
    const uint16_t* qs = rows + hh * kD;
    const uint16_t* ks = rows + (HG + hh) * kD;
    const uint16_t* vs = rows + (2 * HG + hh) * kD;
    const uint16_t* ds = rows + (3 * HG + hh) * kD;
    const float* lq = lse_s + hh * TPE;
    const float* dl = delta_s + hh * TPE;
    // ---- pass 1: dQ of query tile qt (lane: query c) ----
    for (int qt = 0; qt < tiles; ++qt) {
      const int qi = qt * 16 + c;
      const s16x4 bq = ld4(qs + qi * LDR + 4 * g);
      const s16x4 bdo = ld4(ds + qi * LDR + 4 * g);
      const float lse_q = lq[qi], delta_q = dl[qi];
      f32x4 dsT[MT + 1];
      dsT[MT] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < MT; ++kt) {
        dsT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < tiles) {
          const s16x4 ak = ld4(ks + (kt * 16 + c) * LDR + 4 * g);
          const s16x4 av = ld4(vs + (kt * 16 + c) * LDR + 4 * g);
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          const f32x4 sc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ak, bq, z, 0, 0, 0);   // S^T[key][query]
          const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bdo, z, 0, 0, 0);  // dP^T[key][query]
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            const float p = key < len ? __expf(sc[r] * scale - lse_q) : 0.f;
            dsT[kt][r] = p * (dp[r] - delta_q) * scale;
          }
        }
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < (MT + 1) / 2; ++u) {
        if (2 * u < tiles) {
          const bf16x8 pb = pack_tiles(dsT[2 * u], dsT[2 * u + 1]);
          const uint16_t* a0 = ks + (32 * u + 4 * g + q_) * LDR + 4 * p_;
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(a0, a0 + 16 * LDR), pb, acc, 0, 0, 0);
        }
      }
      const int64_t qrow = qi < T ? rowof(qi) : -1;
      if (qrow >= 0) st4(dq + qrow * dq_stride + h * kD + 4 * g, acc);
    }
    // ---- pass 2: dK, dV of key tile kt (lane: key c) ----
    for (int kt = 0; kt < tiles; ++kt) {
      const int kj = kt * 16 + c;
      const s16x4 bk = ld4(ks + kj * LDR + 4 * g);
      const s16x4 bv = ld4(vs + kj * LDR + 4 * g);
      const bool key_ok = kj < len;
      f32x4 dsv[MT + 1], pv[MT + 1];
      dsv[MT] = f32x4{0.f, 0.f, 0.f, 0.f};
      pv[MT] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qt = 0; qt < MT; ++qt) {
        dsv[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        pv[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (qt < tiles) {
          const s16x4 aq = ld4(qs + (qt * 16 + c) * LDR + 4 * g);
          const s16x4 ado = ld4(ds + (qt * 16 + c) * LDR + 4 * g);
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          const f32x4 sc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(aq, bk, z, 0, 0, 0);   // S[query][key]
          const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ado, bv, z, 0, 0, 0);  // dP[query][key]
          const f32x4 l4 = *(const f32x4*)(lq + qt * 16 + 4 * g);
          const f32x4 d4 = *(const f32x4*)(dl + qt * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qt * 16 + 4 * g + r;
            const float p = (key_ok && qi < T) ? __expf(sc[r] * scale - l4[r]) : 0.f;
            pv[qt][r] = p;
            dsv[qt][r] = p * (dp[r] - d4[r]) * scale;
          }
        }
      }
      f32x4 acck = {0.f, 0.f, 0.f, 0.f}, accv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < (MT + 1) / 2; ++u) {
        if (2 * u < tiles) {
          const uint16_t* aq0 = qs + (32 * u + 4 * g + q_) * LDR + 4 * p_;
          const uint16_t* ad0 = ds + (32 * u + 4 * g + q_) * LDR + 4 * p_;
          acck = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(aq0, aq0 + 16 * LDR),
                                                         pack_tiles(dsv[2 * u], dsv[2 * u + 1]), acck, 0, 0, 0);
          accv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(ad0, ad0 + 16 * LDR),
                                                         pack_tiles(pv[2 * u], pv[2 * u + 1]), accv, 0, 0, 0);
        }
      }
      const int64_t krow = kj < T ? rowof(kj) : -1;
      if (krow >= 0) {
        st4(dk + krow * dk_stride + h * kD + 4 * g, acck);
        st4(dv + krow * dv_stride + h * kD + 4 * g, accv);
      }
    }
  }
}

inline int pick_head_group(int H, int T, bool bwd) {
  const int tiles = (T + 15) >> 4, TPE = ((tiles + 1) >> 1) * 32;
  const int cands[4] = {8, 4, 2, 1};
  for (int i = 0; i < 4; ++i) {
    const int hg = cands[i];
    if (H % hg) continue;
    const int64_t bytes = bwd ? (int64_t)TPE * (4 * hg * kD + 16) * 2 + 2 * hg * TPE * 4
                              : (int64_t)TPE * (2 * hg * kD + 16) * 2;
    if (bytes <= 150 * 1024) return hg;
  }
  return 1;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

static int attn_fwd_impl(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_stride, int64_t k_stride,
                         int64_t v_stride, const int32_t* key_len, int64_t num_windows, int32_t max_tokens,
                         int32_t num_heads, int32_t head_dim, float scale, uint16_t* out, int64_t out_stride,
                         float* lse, const int32_t* tok, ococc_stream_t stream) {
  OCOCC_REQUIRE(head_dim == kD, "window attention is built for head_dim 16 (d_model 128, 8 heads)");
  OCOCC_REQUIRE(max_tokens >= 1 && max_tokens <= kMaxTiles * 16, "max_tokens must be 1..160");
  OCOCC_REQUIRE(num_windows >= 0 && num_heads >= 1, "bad sizes");
  if (num_windows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(q && k && v && key_len && out, "null pointer");
  OCOCC_REQUIRE(q_stride % 4 == 0 && out_stride % 4 == 0, "row strides must be multiples of 4 elements");
  OCOCC_REQUIRE(k_stride % 8 == 0 && v_stride % 8 == 0 && aligned16(k) && aligned16(v),
                "k / v rows must be 16-byte aligned (row stride a multiple of 8 elements)");
  const int T = (int)max_tokens, HG = pick_head_group(num_heads, T, false);
  const int tiles = (T + 15) >> 4, TPE = ((tiles + 1) >> 1) * 32;
  const int lds = TPE * (2 * HG * kD + 16) * 2;
#define OCOCC_ATTN_FWD(MT)                                                                                  \
  do {                                                                                                       \
    OCOCC_HIP(hipFuncSetAttribute((const void*)window_attn_fwd_kernel<MT>,                                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));                         \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(window_attn_fwd_kernel<MT>),                                          \
                       dim3((unsigned)num_windows, (unsigned)(num_heads / HG)), dim3(256), lds,              \
                       (hipStream_t)stream, q, k, v, q_stride, k_stride, v_stride, key_len, T,               \
                       (int)num_heads, HG, scale, out, out_stride, lse, tok);                                \
  } while (0)
  if (tiles <= 2) OCOCC_ATTN_FWD(2);
  else if (tiles <= 4) OCOCC_ATTN_FWD(4);
  else if (tiles <= 7) OCOCC_ATTN_FWD(7);
  else OCOCC_ATTN_FWD(10);
#undef OCOCC_ATTN_FWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_window_attn_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v,
                                          int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                          const int32_t* key_len, int64_t num_windows, int32_t max_tokens,
                                          int32_t num_heads, int32_t head_dim, float scale, uint16_t* out,
                                          int64_t out_stride, float* lse, ococc_stream_t stream) {
  return attn_fwd_impl(q, k, v, q_stride, k_stride, v_stride, key_len, num_windows, max_tokens, num_heads, head_dim,
                       scale, out, out_stride, lse, nullptr, stream);
}

extern "C" int ococc_window_attn_fwd_gather_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v,
                                                 int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                                 const int32_t* token_index, const int32_t* key_len,
                                                 int64_t num_windows, int32_t max_tokens, int32_t num_heads,
                                                 int32_t head_dim, float scale, uint16_t* out, int64_t out_stride,
                                                 float* lse, ococc_stream_t stream) {
  OCOCC_REQUIRE(token_index || num_windows == 0, "null token_index");
  return attn_fwd_impl(q, k, v, q_stride, k_stride, v_stride, key_len, num_windows, max_tokens, num_heads, head_dim,
                       scale, out, out_stride, lse, token_index, stream);
}

static int attn_bwd_impl(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t q_stride, int64_t k_stride,
                         int64_t v_stride, const uint16_t* out, const uint16_t* dout, int64_t o_stride,
                         const float* lse, const int32_t* key_len, int64_t num_windows, int32_t max_tokens,
                         int32_t num_heads, int32_t head_dim, float scale, uint16_t* dq, uint16_t* dk, uint16_t* dv,
                         int64_t dq_stride, int64_t dk_stride, int64_t dv_stride, const int32_t* tok,
                         ococc_stream_t stream) {
  OCOCC_REQUIRE(head_dim == kD, "window attention is built for head_dim 16");
  OCOCC_REQUIRE(max_tokens >= 1 && max_tokens <= kMaxTiles * 16, "max_tokens must be 1..160");
  if (num_windows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(q && k && v && out && dout && lse && key_len && dq && dk && dv, "null pointer");
  OCOCC_REQUIRE(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 && o_stride % 8 == 0 &&
                    aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out) && aligned16(dout),
                "q / k / v / out / dout rows must be 16-byte aligned (row stride a multiple of 8 elements)");
  OCOCC_REQUIRE(dq_stride % 4 == 0 && dk_stride % 4 == 0 && dv_stride % 4 == 0, "gradient row strides must be multiples of 4");
  const int T = (int)max_tokens, HG = pick_head_group(num_heads, T, true);
  const int tiles = (T + 15) >> 4, TPE = ((tiles + 1) >> 1) * 32;
  const int lds = TPE * (4 * HG * kD + 16) * 2 + 2 * HG * TPE * 4;
#define OCOCC_ATTN_BWD(MT)                                                                                  \
  do {                                                                                                       \
    OCOCC_HIP(hipFuncSetAttribute((const void*)window_attn_bwd_kernel<MT>,                                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds));                         \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(window_attn_bwd_kernel<MT>),                                          \
                       dim3((unsigned)num_windows, (unsigned)(num_heads / HG)), dim3(256), lds,              \
                       (hipStream_t)stream, q, k, v, q_stride, k_stride, v_stride, out, dout, o_stride, lse, \
                       key_len, T, (int)num_heads, HG, scale, dq, dk, dv, dq_stride, dk_stride, dv_stride,   \
                       tok);                                                                                 \
  } while (0)
  if (tiles <= 2) OCOCC_ATTN_BWD(2);
  else if (tiles <= 4) OCOCC_ATTN_BWD(4);
  else if (tiles <= 7) OCOCC_ATTN_BWD(7);
  else OCOCC_ATTN_BWD(10);
#undef OCOCC_ATTN_BWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_window_attn_bwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v,
                                          int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                          const uint16_t* out, const uint16_t* dout, int64_t o_stride,
                                          const float* lse, const int32_t* key_len, int64_t num_windows,
                                          int32_t max_tokens, int32_t num_heads, int32_t head_dim, float scale,
                                          uint16_t* dq, uint16_t* dk, uint16_t* dv, int64_t dq_stride,
                                          int64_t dk_stride, int64_t dv_stride, ococc_stream_t stream) {
  return attn_bwd_impl(q, k, v, q_stride, k_stride, v_stride, out, dout, o_stride, lse, key_len, num_windows,
                       max_tokens, num_heads, head_dim, scale, dq, dk, dv, dq_stride, dk_stride, dv_stride, nullptr,
                       stream);
}

extern "C" int ococc_window_attn_bwd_gather_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v,
                                                 int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                                 const uint16_t* out, const uint16_t* dout, int64_t o_stride,
                                                 const float* lse, const int32_t* token_index,
                                                 const int32_t* key_len, int64_t num_windows, int32_t max_tokens,
                                                 int32_t num_heads, int32_t head_dim, float scale, uint16_t* dq,
                                                 uint16_t* dk, uint16_t* dv, int64_t dq_stride, int64_t dk_stride,
                                                 int64_t dv_stride, ococc_stream_t stream) {
  OCOCC_REQUIRE(token_index || num_windows == 0, "null token_index");
  return attn_bwd_impl(q, k, v, q_stride, k_stride, v_stride, out, dout, o_stride, lse, key_len, num_windows,
                       max_tokens, num_heads, head_dim, scale, dq, dk, dv, dq_stride, dk_stride, dv_stride,
                       token_index, stream);
}
