// B7 window attention core: softmax(Q K^T / sqrt(d) + key mask) V for padded windows of the SST
// blocks (mmdet3d/models/sst/sst_basic_block_v2.py:41-75 runs nn.MultiheadAttention on
// [num_windows, max_tokens, C] tensors built by flat2window_v2).  d_head = 16 (d_model 128,
// 8 heads), max_tokens <= 160; tokens of a window occupy its first key_len slots, so the
// key_padding_mask is a length.
//
// Forward, one workgroup per (window, head), gfx950 MFMA:
//   S^T = K Q^T  with v_mfma_f32_16x16x16_bf16 (k = d_head = 16, no padding waste): keys on the
//   MFMA rows, queries on the lanes, so a lane owns one query column and the softmax over keys is
//   a register reduction + two wave shuffles;
//   O^T = V^T P^T with v_mfma_f32_16x16x32_bf16: the probabilities are already in the B-operand
//   registers (two 16-key score tiles form one 32-deep k-step with a permuted key order; V^T is
//   read from LDS in the same permuted order) -- no LDS round trip for P.
//   A lane ends with 4 consecutive channels of one query: 8-byte stores.
// Backward: recompute P from Q, K and the saved log-sum-exp; dQ by a thread per query, dK/dV by
// a thread per key (two O(T^2 d) passes in LDS, fp32 VALU, deterministic, no atomics).
#include "common.hpp"

namespace {

constexpr int kD = 16;        // head dim
constexpr int kMaxTiles = 10; // max_tokens <= 160
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ s16x4 ld4(const uint16_t* p) { return *(const s16x4*)p; }

__global__ void __launch_bounds__(256)
window_attn_fwd_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                       const uint16_t* __restrict__ v, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                       const int32_t* __restrict__ key_len, int T, int H, float scale,
                       uint16_t* __restrict__ out, int64_t out_stride, float* __restrict__ lse) {
  // LDS: K rows [TP][16] and V^T [16][TP + 8]
  __shared__ __attribute__((aligned(16))) uint16_t ks[kMaxTiles * 16 * kD];
  __shared__ __attribute__((aligned(16))) uint16_t vt[kD * (kMaxTiles * 16 + 8)];
  const int w = blockIdx.x, h = blockIdx.y;
  const int tiles = (T + 15) >> 4, TP = tiles * 16, LDV = kMaxTiles * 16 + 8;
  const int len = key_len[w];
  const int64_t row0 = (int64_t)w * T;
  const int TPE = ((tiles + 1) >> 1) * 32;  // PV consumes key tiles in pairs: zero the odd tail too
  for (int i = threadIdx.x; i < TPE * kD; i += 256) {
    const int t = i / kD, d = i % kD;
    uint16_t kv = 0, vv = 0;
    if (t < T) {
      kv = k[(row0 + t) * k_stride + h * kD + d];
      vv = v[(row0 + t) * v_stride + h * kD + d];
    }
    if (t < TP) ks[t * kD + d] = kv;
    vt[d * LDV + t] = vv;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  for (int qt = wave; qt < tiles; qt += 4) {
    const int qi = qt * 16 + c;  // this lane's query
    s16x4 bq = {0, 0, 0, 0};
    if (qi < T) bq = ld4(q + (row0 + qi) * q_stride + h * kD + 4 * g);
    f32x4 s[kMaxTiles];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < kMaxTiles; ++kt) {
      s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (kt < tiles) {
        const s16x4 ak = ld4(ks + (kt * 16 + c) * kD + 4 * g);
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
    for (int kt = 0; kt < kMaxTiles; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (kt < tiles && m > -INFINITY) ? __expf(s[kt][r] - m) : 0.f;
        s[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    // O^T[d][query] = sum_key V^T[d][key] P[key][query]; k-step u covers key tiles 2u, 2u+1 with the
    // permuted order  position 8g+j -> key (j<4 ? 32u + 4g + j : 32u + 16 + 4g + j - 4)
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kMaxTiles / 2; ++u) {
      if (2 * u < tiles) {
        bf16x8 pb, va;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pb[j] = (__bf16)(s[2 * u][j] * inv);
          pb[4 + j] = (__bf16)(s[2 * u + 1][j] * inv);  // zeros when tile 2u+1 does not exist
        }
        const bf16x4 v0 = *(const bf16x4*)(vt + c * LDV + 32 * u + 4 * g);
        const bf16x4 v1 = *(const bf16x4*)(vt + c * LDV + 32 * u + 16 + 4 * g);
        va = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o, 0, 0, 0);  // rows: d = 4g+r, col: query c
      }
    }
    if (qi < T) {
      u32x2 p;
      p.x = (uint32_t)ococc_f32_to_bf16(o[0]) | ((uint32_t)ococc_f32_to_bf16(o[1]) << 16);
      p.y = (uint32_t)ococc_f32_to_bf16(o[2]) | ((uint32_t)ococc_f32_to_bf16(o[3]) << 16);
      *(u32x2*)(out + (row0 + qi) * out_stride + h * kD + 4 * g) = p;
      if (lse && g == 0) lse[((int64_t)w * H + h) * T + qi] = sum > 0.f ? m + __logf(sum) : 0.f;
    }
  }
}

__global__ void __launch_bounds__(256)
window_attn_bwd_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                       const uint16_t* __restrict__ v, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                       const uint16_t* __restrict__ out, const uint16_t* __restrict__ dout, int64_t o_stride,
                       const float* __restrict__ lse, const int32_t* __restrict__ key_len, int T, int H,
                       float scale, uint16_t* __restrict__ dq, uint16_t* __restrict__ dk,
                       uint16_t* __restrict__ dv, int64_t dq_stride, int64_t dk_stride, int64_t dv_stride) {
  constexpr int TM = kMaxTiles * 16;
  __shared__ float qs[TM * kD], ks[TM * kD], vs[TM * kD], dos[TM * kD];
  __shared__ float delta[TM], lses[TM];
  const int w = blockIdx.x, h = blockIdx.y;
  const int len = key_len[w];
  const int64_t row0 = (int64_t)w * T;
  for (int i = threadIdx.x; i < T * kD; i += 256) {
    const int t = i / kD, d = i % kD;
    qs[i] = ococc_bf16_to_f32(q[(row0 + t) * q_stride + h * kD + d]);
    ks[i] = ococc_bf16_to_f32(k[(row0 + t) * k_stride + h * kD + d]);
    vs[i] = ococc_bf16_to_f32(v[(row0 + t) * v_stride + h * kD + d]);
    dos[i] = ococc_bf16_to_f32(dout[(row0 + t) * o_stride + h * kD + d]);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256) {
    float s = 0.f;
    for (int d = 0; d < kD; ++d) s += dos[t * kD + d] * ococc_bf16_to_f32(out[(row0 + t) * o_stride + h * kD + d]);
    delta[t] = s;
    lses[t] = lse[((int64_t)w * H + h) * T + t];
  }
  __syncthreads();
  // pass 1: thread per query i -> dQ_i
  for (int i = threadIdx.x; i < T; i += 256) {
    float acc[kD];
#pragma unroll
    for (int d = 0; d < kD; ++d) acc[d] = 0.f;
    for (int j = 0; j < len; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < kD; ++d) {
        s += qs[i * kD + d] * ks[j * kD + d];
        dp += dos[i * kD + d] * vs[j * kD + d];
      }
      const float p = __expf(s * scale - lses[i]);
      const float ds = p * (dp - delta[i]) * scale;
#pragma unroll
      for (int d = 0; d < kD; ++d) acc[d] += ds * ks[j * kD + d];
    }
#pragma unroll
    for (int d = 0; d < kD; ++d) dq[(row0 + i) * dq_stride + h * kD + d] = ococc_f32_to_bf16(acc[d]);
  }
  // pass 2: thread per key j -> dK_j, dV_j
  for (int j = threadIdx.x; j < T; j += 256) {
    float ak[kD], av[kD];
#pragma unroll
    for (int d = 0; d < kD; ++d) ak[d] = av[d] = 0.f;
    if (j < len) {
      for (int i = 0; i < T; ++i) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < kD; ++d) {
          s += qs[i * kD + d] * ks[j * kD + d];
          dp += dos[i * kD + d] * vs[j * kD + d];
        }
        const float p = __expf(s * scale - lses[i]);
        const float ds = p * (dp - delta[i]) * scale;
#pragma unroll
        for (int d = 0; d < kD; ++d) {
          ak[d] += ds * qs[i * kD + d];
          av[d] += p * dos[i * kD + d];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < kD; ++d) {
      dk[(row0 + j) * dk_stride + h * kD + d] = ococc_f32_to_bf16(ak[d]);
      dv[(row0 + j) * dv_stride + h * kD + d] = ococc_f32_to_bf16(av[d]);
    }
  }
}

}  // namespace

extern "C" int ococc_window_attn_fwd_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v,
                                          int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                          const int32_t* key_len, int64_t num_windows, int32_t max_tokens,
                                          int32_t num_heads, int32_t head_dim, float scale, uint16_t* out,
                                          int64_t out_stride, float* lse, ococc_stream_t stream) {
  OCOCC_REQUIRE(head_dim == kD, "window attention is built for head_dim 16 (d_model 128, 8 heads)");
  OCOCC_REQUIRE(max_tokens >= 1 && max_tokens <= kMaxTiles * 16, "max_tokens must be 1..160");
  OCOCC_REQUIRE(num_windows >= 0 && num_heads >= 1, "bad sizes");
  if (num_windows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(q && k && v && key_len && out, "null pointer");
  OCOCC_REQUIRE(q_stride % 4 == 0 && out_stride % 4 == 0, "row strides must be multiples of 4 elements");
  hipLaunchKernelGGL(window_attn_fwd_kernel, dim3((unsigned)num_windows, (unsigned)num_heads), dim3(256), 0,
                     (hipStream_t)stream, q, k, v, q_stride, k_stride, v_stride, key_len, (int)max_tokens,
                     (int)num_heads, scale, out, out_stride, lse);
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
  OCOCC_REQUIRE(head_dim == kD, "window attention is built for head_dim 16");
  OCOCC_REQUIRE(max_tokens >= 1 && max_tokens <= kMaxTiles * 16, "max_tokens must be 1..160");
  if (num_windows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(q && k && v && out && dout && lse && key_len && dq && dk && dv, "null pointer");
  hipLaunchKernelGGL(window_attn_bwd_kernel, dim3((unsigned)num_windows, (unsigned)num_heads), dim3(256), 0,
                     (hipStream_t)stream, q, k, v, q_stride, k_stride, v_stride, out, dout, o_stride, lse,
                     key_len, (int)max_tokens, (int)num_heads, scale, dq, dk, dv, dq_stride, dk_stride, dv_stride);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
