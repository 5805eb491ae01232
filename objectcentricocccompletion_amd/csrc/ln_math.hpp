// Row arithmetic of the LayerNorm (+ GELU) backward, shared by the stand-alone kernels (layernorm_act.hip) and the
// epilogue of the tile convolution (sparse_conv_tile.hip) so that both produce the same bits.
//
// These kernels are VALU-bound, not HBM-bound: ~45 scalar f32 instructions per element x 16 lanes per clock and SIMD is
// 20 us for the 126 k x 128 activation of configs[1], against 12 us for its 96 MB at 8 TB/s.  Hence (a) packed f32
// arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two channels per instruction), (b) v_rcp_f32 instead of
// the IEEE division sequence __frcp_rn expands to (10 instructions), (c) the sign select of the normal CDF as a
// bit-field insert.  Reference arithmetic: nn.LayerNorm + nn.GELU() (exact erf), sparse_block.py:216-289.
#pragma once
#include "common.hpp"

typedef float ln_f32x2 __attribute__((ext_vector_type(2)));

// d GELU(z) / dz = Phi(z) + z phi(z) for two channels.  Phi through Abramowitz & Stegun 7.1.26
//   erf(x) = 1 - (a1 t + ... + a5 t^5) exp(-x^2),  t = 1 / (1 + p x),  |error| <= 1.5e-7  (x >= 0)
// with x = |z| / sqrt 2, so that exp(-x^2) = exp(-z^2 / 2) is also the Gaussian density's exponential.
__device__ __forceinline__ ln_f32x2 ln_gelu_grad2(ln_f32x2 z) {
  const ln_f32x2 zz = z * z;
  ln_f32x2 e, t, az;
  e.x = __builtin_amdgcn_exp2f(zz.x * -0.72134752044448170368f);  // exp(-z^2/2) = 2^(-z^2 log2(e) / 2)
  e.y = __builtin_amdgcn_exp2f(zz.y * -0.72134752044448170368f);
  az.x = __builtin_fabsf(z.x);
  az.y = __builtin_fabsf(z.y);
  const ln_f32x2 d = az * (0.3275911f * 0.70710678118654752440f) + 1.f;
  t.x = __builtin_amdgcn_rcpf(d.x);
  t.y = __builtin_amdgcn_rcpf(d.y);
  ln_f32x2 poly = t * 1.061405429f + -1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t + -0.284496736f;
  poly = poly * t + 0.254829592f;
  const ln_f32x2 q = 0.5f - (poly * t) * (e * 0.5f);   // Phi(|z|) - 1/2  (>= 0)
  ln_f32x2 cdf;
  cdf.x = 0.5f + __builtin_copysignf(q.x, z.x);
  cdf.y = 0.5f + __builtin_copysignf(q.y, z.y);
  return cdf + z * (e * 0.39894228040143267794f);
}

// GELU(z) = z Phi(z) for two channels through the Phi of ln_gelu_grad2 (exp + rcp): rounds 1-2's forward GELU, kept as the
// reference form of ln_gelu2 below (which every forward kernel uses since round 3)
__device__ __forceinline__ ln_f32x2 ln_gelu2_exp(ln_f32x2 z) {
  const ln_f32x2 zz = z * z;
  ln_f32x2 e, t, az;
  e.x = __builtin_amdgcn_exp2f(zz.x * -0.72134752044448170368f);
  e.y = __builtin_amdgcn_exp2f(zz.y * -0.72134752044448170368f);
  az.x = __builtin_fabsf(z.x);
  az.y = __builtin_fabsf(z.y);
  const ln_f32x2 d = az * (0.3275911f * 0.70710678118654752440f) + 1.f;
  t.x = __builtin_amdgcn_rcpf(d.x);
  t.y = __builtin_amdgcn_rcpf(d.y);
  ln_f32x2 poly = t * 1.061405429f + -1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t + -0.284496736f;
  poly = poly * t + 0.254829592f;
  const ln_f32x2 q = 0.5f - (poly * t) * (e * 0.5f);
  ln_f32x2 cdf;
  cdf.x = 0.5f + __builtin_copysignf(q.x, z.x);
  cdf.y = 0.5f + __builtin_copysignf(q.y, z.y);
  return z * cdf;
}

// GELU with ONE transcendental per element -- the forward GELU of every kernel here (VALU-bound epilogues next to the
// matrix pipe: mlp_layer.hip, window_block.hip; the LayerNorm kernels; the convolution epilogues):
//   GELU(z) = max(z, 0) - |z| u(|z|),   u(a) = (1 - erf(a / sqrt 2)) / 2 = 1 / (2 P(a)^16)
// Abramowitz & Stegun 7.1.28, erf(x) = 1 - (1 + a1 x + ... + a6 x^6)^-16, |error| <= 3e-7 (x >= 0); the coefficients
// below are a_i / sqrt(2)^i.  No exponential, no sign select: 12 packed operations, 2 reciprocals and 4 single ones per
// pair against 14 + 4 + 6 above.  |GELU error| <= 1.5e-7 |z|.
__device__ __forceinline__ ln_f32x2 ln_gelu2(ln_f32x2 z) {
  ln_f32x2 a, r, relu;
  a.x = __builtin_fabsf(z.x);
  a.y = __builtin_fabsf(z.y);
  ln_f32x2 p = a * 5.38297500e-6f + 4.88906356e-5f;
  p = p * a + 3.80035750e-5f;
  p = p * a + 3.27762632e-3f;
  p = p * a + 2.11410061e-2f;
  p = p * a + 4.98673470e-2f;
  p = p * a + 1.f;
  r.x = __builtin_amdgcn_rcpf(p.x);
  r.y = __builtin_amdgcn_rcpf(p.y);
  r = r * r;
  r = r * r;
  r = r * r;
  r = r * r;
  relu.x = __builtin_fmaxf(z.x, 0.f);
  relu.y = __builtin_fmaxf(z.y, 0.f);
  return relu - (a * r) * 0.5f;
}
// One value at a time (the convolution epilogues, the generic LayerNorm kernel): the exp + rcp form -- its two short
// dependency chains finish sooner than the single long one above where there is no second channel to interleave with
// (the 32 -> 64 tile convolution with LN + GELU epilogue: 23.4 us against 25.0 us).  The two forms agree to 1e-6.
__device__ __forceinline__ float ln_gelu1(float z) { return ln_gelu2_exp(ln_f32x2{z, z}).x; }

// Dropout behind the activation (build_mlp's Sequential(Linear, norm, act, Dropout), sst_ops.py:333-360), folded into
// the LN kernels: the keep mask is a counter-based hash of (seed, element index) -- one 32-bit hash per channel pair,
// 16 bits each against the threshold -- so the backward kernel regenerates it instead of reading a stored mask
// (torch's dropout writes a byte mask and reads it back: 5.6 ms per step on the decoder's [1 M, 1024] activations).
// thr = round(p * 65536) (0: no dropout), scale = 65536 / (65536 - thr).
struct LnDropout {
  uint32_t thr;
  float scale;
  uint32_t seed_lo, seed_hi;
};
__device__ __forceinline__ ln_f32x2 ln_dropout_mask2(const LnDropout& d, int64_t row, int pair, int pairs_per_row) {
  uint32_t h = ((uint32_t)row * (uint32_t)pairs_per_row + (uint32_t)pair) ^ d.seed_lo;
  h *= 0x9E3779B1u;
  h ^= h >> 16;
  h = (h + d.seed_hi + (uint32_t)(row >> 24)) * 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return ln_f32x2{(h & 0xffffu) >= d.thr ? d.scale : 0.f, (h >> 16) >= d.thr ? d.scale : 0.f};
}

__device__ __forceinline__ void ln_unpack8(const u32x4 v, ln_f32x2 (&f)[4]) {
  f[0] = ln_f32x2{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
  f[1] = ln_f32x2{__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
  f[2] = ln_f32x2{__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u)};
  f[3] = ln_f32x2{__uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)};
}
__device__ __forceinline__ uint32_t ln_pack2(ln_f32x2 v) {
  return (uint32_t)ococc_f32_to_bf16(v.x) | ((uint32_t)ococc_f32_to_bf16(v.y) << 16);
}

// One 8-channel piece of a row.  x: the block's conv output, dv: the gradient of the block's output (both as
// floats of bf16 values), g / b: gamma / beta of the lane's channels.  Adds the piece's terms to dg / db, leaves
// xhat and dz * gamma for the second half, returns the lane's share of the two row sums.
template <bool GELU>
__device__ __forceinline__ void ln_bwd_piece8(ln_f32x2 (&x)[4], const ln_f32x2 (&dv)[4], float mean, float rstd,
                                              const ln_f32x2 (&g)[4], const ln_f32x2 (&b)[4], ln_f32x2 (&dg)[4],
                                              ln_f32x2 (&db)[4], ln_f32x2 (&dzg)[4], float& s1, float& s2) {
  ln_f32x2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    x[p] = (x[p] - mean) * rstd;  // xhat
    ln_f32x2 dz = dv[p];
    if (GELU) dz = dz * ln_gelu_grad2(x[p] * g[p] + b[p]);
    dg[p] += dz * x[p];
    db[p] += dz;
    dzg[p] = dz * g[p];
    a1 += dzg[p];
    a2 += dzg[p] * x[p];
  }
  s1 = a1.x + a1.y;
  s2 = a2.x + a2.y;
}
// second half: s1, s2 = the row's two sums divided by the channel count -> the 8 gradients, packed to bf16
__device__ __forceinline__ u32x4 ln_bwd_finish8(const ln_f32x2 (&xh)[4], const ln_f32x2 (&dzg)[4], float rstd,
                                                float s1, float s2) {
  u32x4 q;
  q.x = ln_pack2(((dzg[0] - s1) - xh[0] * s2) * rstd);
  q.y = ln_pack2(((dzg[1] - s1) - xh[1] * s2) * rstd);
  q.z = ln_pack2(((dzg[2] - s1) - xh[2] * s2) * rstd);
  q.w = ln_pack2(((dzg[3] - s1) - xh[3] * s2) * rstd);
  return q;
}
