// B5 fused LayerNorm (+ GELU) over feature rows, forward and backward.
// Reference: the norm/act layers make_sparse_convmodule appends after a sparse
// conv (mmdet3d/ops/sparse_block.py:216-289) and every Linear->LN->GELU stage
// of build_mlp (mmdet3d/ops/sst/sst_ops.py:333-360): separate LayerNorm and GELU
// kernels, each reading and writing the [n, c] activation.
// Here: one pass per direction.  HBM-bound: forward moves 2*n*c*s bytes,
// backward 3*n*c*s.  A row is spread over LPR lanes (power of two), lane li
// owns channels li, li+LPR, ... so that a wave instruction touches whole rows
// contiguously; row statistics are reduced with wave shuffles, never LDS.
// dgamma/dbeta: per-lane running sums over the rows a workgroup owns, combined
// through LDS, written to per-workgroup slabs and added in a fixed order.
#include "common.hpp"
#include "ln_math.hpp"
#include "param_reduce.hpp"

namespace {

constexpr int kBwdMaxBlocks = 1024;  // one dgamma/dbeta slab per block; 4 blocks per CU = the 16 waves the registers allow
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<uint16_t>(const uint16_t* p) { return ococc_bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<uint16_t>(uint16_t* p, float v) { *p = ococc_f32_to_bf16(v); }

__device__ __forceinline__ float group_sum(float v, int lpr) {
  for (int d = lpr >> 1; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// GELU derivative without libm's erff: Abramowitz & Stegun 7.1.26,
//   erf(x) = 1 - (a1 t + ... + a5 t^5) exp(-x^2),  t = 1 / (1 + p x),  |error| <= 1.5e-7  (x >= 0),
// whose exponential exp(-z^2/2) is the one the Gaussian density of the GELU derivative needs anyway.
// Phi(z) = (1 + erf(z / sqrt 2)) / 2; e = exp(-z^2/2) is returned for the caller.
__device__ __forceinline__ float norm_cdf(float z, float* e_out) {
  const float x = fabsf(z) * kInvSqrt2;
  const float e = __expf(-x * x);
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * x);  // (v_rcp_f32; __frcp_rn expands to the 10-instruction IEEE division)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float half_tail = 0.5f * poly * e;  // (1 - erf(x)) / 2
  *e_out = e;
  return z >= 0.f ? 1.f - half_tail : half_tail;
}
// forward: the same Phi, packed where the kernel walks channel pairs (ln_math.hpp)
__device__ __forceinline__ float gelu(float z) { return ln_gelu1(z); }
__device__ __forceinline__ float gelu_grad(float z) {
  float e;
  const float cdf = norm_cdf(z, &e);
  return cdf + z * kInvSqrt2Pi * e;
}

// VPL = channels per lane held in registers (c <= VPL * lpr)
template <typename T, int VPL>
__global__ void __launch_bounds__(256)
ln_act_fwd_kernel(const T* __restrict__ x, int64_t n, int c, int lpr,
                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                  int act, T* __restrict__ y, float* __restrict__ mean_rstd) {
  const int rows_per_block = 256 / lpr;
  const int li = threadIdx.x % lpr;
  const int rloc = threadIdx.x / lpr;
  const float inv_c = 1.f / (float)c;
  for (int64_t r = (int64_t)blockIdx.x * rows_per_block + rloc; r < n;
       r += (int64_t)gridDim.x * rows_per_block) {
    float v[VPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int ch = li + j * lpr;
      v[j] = ch < c ? ld<T>(x + r * c + ch) : 0.f;
      s += v[j];
    }
    const float mean = group_sum(s, lpr) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int ch = li + j * lpr;
      const float d = ch < c ? v[j] - mean : 0.f;
      sq += d * d;
    }
    const float rstd = rsqrtf(group_sum(sq, lpr) * inv_c + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int ch = li + j * lpr;
      if (ch < c) {
        float z = (v[j] - mean) * rstd * gamma[ch] + beta[ch];
        if (act == 1) z = gelu(z);
        st<T>(y + r * c + ch, z);
      }
    }
    if (mean_rstd && li == 0) {
      mean_rstd[r * 2] = mean;
      mean_rstd[r * 2 + 1] = rstd;
    }
  }
}

template <typename T, int VPL>
__global__ void __launch_bounds__(256)
ln_act_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t n, int c, int lpr,
                  const float* __restrict__ gamma, const float* __restrict__ beta,
                  const float* __restrict__ mean_rstd, int act, T* __restrict__ dx,
                  float* __restrict__ partials) {
  const int rows_per_block = 256 / lpr;
  const int li = threadIdx.x % lpr;
  const int rloc = threadIdx.x / lpr;
  const float inv_c = 1.f / (float)c;
  float dg[VPL], db[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) dg[j] = db[j] = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * rows_per_block + rloc; r < n;
       r += (int64_t)gridDim.x * rows_per_block) {
    const float mean = mean_rstd[r * 2], rstd = mean_rstd[r * 2 + 1];
    float xh[VPL], dzg[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int ch = li + j * lpr;
      xh[j] = 0.f;
      dzg[j] = 0.f;
      if (ch < c) {
        const float g = gamma[ch];
        xh[j] = (ld<T>(x + r * c + ch) - mean) * rstd;
        float dz = ld<T>(dy + r * c + ch);
        if (act == 1) dz *= gelu_grad(xh[j] * g + beta[ch]);
        dg[j] += dz * xh[j];
        db[j] += dz;
        dzg[j] = dz * g;
        s1 += dzg[j];
        s2 += dzg[j] * xh[j];
      }
    }
    s1 = group_sum(s1, lpr) * inv_c;
    s2 = group_sum(s2, lpr) * inv_c;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int ch = li + j * lpr;
      if (ch < c) st<T>(dx + r * c + ch, rstd * (dzg[j] - s1 - xh[j] * s2));
    }
  }
  // combine the row groups of this block through LDS (fixed order), one slab per block
  extern __shared__ __attribute__((aligned(16))) float red[];  // [rows_per_block][2*c]
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int ch = li + j * lpr;
    if (ch < c) {
      red[rloc * 2 * c + ch] = dg[j];
      red[rloc * 2 * c + c + ch] = db[j];
    }
  }
  __syncthreads();
  float* slab = partials + (int64_t)blockIdx.x * 2 * c;
  for (int i = threadIdx.x; i < 2 * c; i += 256) {
    float s = 0.f;
    for (int g = 0; g < rows_per_block; ++g) s += red[g * 2 * c + i];
    slab[i] = s;
  }
}

// ---- few, wide rows (the RoI-level MLPs and the temporal transformer of configs[2]: 128 .. 1024 rows of 1024 .. 2048 f32
// channels): ONE ROW PER WORKGROUP, channel tid + 256 j in thread tid.  The generic kernels above put 256 / 64 = 4 rows in
// a workgroup and 32 channels in a lane: 32 workgroups for 128 rows, a chain of 32 dependent loads each -- 16 us forward and
// 25 us backward for 1 MB.  Sums cross the four waves through LDS; the backward's parameter partial row IS the row's own
// dz * xhat | dz (one partial row per input row).
constexpr int kRowKernelMaxRows = 1024, kRowKernelMaxC = 2048;   // (<= kBwdMaxBlocks partial rows: the backward workspace holds that many)
inline bool row_kernel_ok(int64_t n, int c) { return n <= kRowKernelMaxRows && c > 512 && c <= kRowKernelMaxC; }

// (a, b) summed over the 256 threads of the workgroup; every thread gets both sums.  Two barriers.
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a += __shfl_xor(a, d, 64);
    b += __shfl_xor(b, d, 64);
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();   // (red may still be read from the previous use)
  if ((threadIdx.x & 63) == 0) {
    red[2 * wave] = a;
    red[2 * wave + 1] = b;
  }
  __syncthreads();
  a = (red[0] + red[2]) + (red[4] + red[6]);
  b = (red[1] + red[3]) + (red[5] + red[7]);
}

template <typename T>
__global__ void __launch_bounds__(256)
ln_act_fwd_row_kernel(const T* __restrict__ x, int c, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                      int act, T* __restrict__ y, float* __restrict__ mean_rstd) {
  __shared__ float red[8];
  const int64_t r = blockIdx.x;
  const float inv_c = 1.f / (float)c;
  float v[8];
  float s = 0.f, zero = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = threadIdx.x + 256 * j;
    v[j] = ch < c ? ld<T>(x + r * c + ch) : 0.f;
    s += v[j];
  }
  block_sum2(s, zero, red);
  const float mean = s * inv_c;
  float sq = 0.f;
  zero = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float d = threadIdx.x + 256 * j < c ? v[j] - mean : 0.f;
    sq += d * d;
  }
  block_sum2(sq, zero, red);
  const float rstd = rsqrtf(sq * inv_c + eps);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = threadIdx.x + 256 * j;
    if (ch < c) {
      float z = (v[j] - mean) * rstd * gamma[ch] + beta[ch];
      if (act == 1) z = gelu(z);
      st<T>(y + r * c + ch, z);
    }
  }
  if (mean_rstd && threadIdx.x == 0) {
    mean_rstd[r * 2] = mean;
    mean_rstd[r * 2 + 1] = rstd;
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
ln_act_bwd_row_kernel(const T* __restrict__ x, const T* __restrict__ dy, int c, const float* __restrict__ gamma,
                      const float* __restrict__ beta, const float* __restrict__ mean_rstd, int act, T* __restrict__ dx,
                      float* __restrict__ partials) {
  __shared__ float red[8];
  const int64_t r = blockIdx.x;
  const float inv_c = 1.f / (float)c;
  const float mean = mean_rstd[r * 2], rstd = mean_rstd[r * 2 + 1];
  float xh[8], dzg[8];
  float s1 = 0.f, s2 = 0.f;
  float* slab = partials + r * 2 * c;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = threadIdx.x + 256 * j;
    xh[j] = dzg[j] = 0.f;
    if (ch < c) {
      const float g = gamma[ch];
      xh[j] = (ld<T>(x + r * c + ch) - mean) * rstd;
      float dz = ld<T>(dy + r * c + ch);
      if (act == 1) dz *= gelu_grad(xh[j] * g + beta[ch]);
      slab[ch] = dz * xh[j];
      slab[c + ch] = dz;
      dzg[j] = dz * g;
      s1 += dzg[j];
      s2 += dzg[j] * xh[j];
    }
  }
  block_sum2(s1, s2, red);
  s1 *= inv_c;
  s2 *= inv_c;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = threadIdx.x + 256 * j;
    if (ch < c) st<T>(dx + r * c + ch, rstd * (dzg[j] - s1 - xh[j] * s2));
  }
}

// ---- vectorised bf16 path: c = 8 * LPR with LPR a power of two <= 64.  Lane li owns the 8
// contiguous channels 8*li .. 8*li+7 (one 16-byte load/store), so a wave instruction moves
// 1 KiB of whole rows; gamma/beta live in registers for the whole kernel.
__device__ __forceinline__ void unpack8(const u32x4 v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
  v.x = (uint32_t)ococc_f32_to_bf16(f[0]) | ((uint32_t)ococc_f32_to_bf16(f[1]) << 16);
  v.y = (uint32_t)ococc_f32_to_bf16(f[2]) | ((uint32_t)ococc_f32_to_bf16(f[3]) << 16);
  v.z = (uint32_t)ococc_f32_to_bf16(f[4]) | ((uint32_t)ococc_f32_to_bf16(f[5]) << 16);
  v.w = (uint32_t)ococc_f32_to_bf16(f[6]) | ((uint32_t)ococc_f32_to_bf16(f[7]) << 16);
  return v;
}

// 8 consecutive channels of a row as four f32 pairs: bf16 rows move 16 bytes per lane, f32 rows 32 (the SIR layers of
// the RoI encoder run their Linear -> LN -> GELU stages in f32: 120 launches per step of configs[2])
struct Ln8 {
  u32x4 a, b;
};
template <typename T> __device__ __forceinline__ Ln8 ln_load8(const T* p);
template <> __device__ __forceinline__ Ln8 ln_load8<uint16_t>(const uint16_t* p) { return Ln8{*(const u32x4*)p, u32x4{0u, 0u, 0u, 0u}}; }
template <> __device__ __forceinline__ Ln8 ln_load8<float>(const float* p) { return Ln8{*(const u32x4*)p, *(const u32x4*)(p + 4)}; }
template <typename T> __device__ __forceinline__ void ln_unpack8t(const Ln8& v, ln_f32x2 (&f)[4]);
template <> __device__ __forceinline__ void ln_unpack8t<uint16_t>(const Ln8& v, ln_f32x2 (&f)[4]) { ln_unpack8(v.a, f); }
template <> __device__ __forceinline__ void ln_unpack8t<float>(const Ln8& v, ln_f32x2 (&f)[4]) {
  f[0] = ln_f32x2{__uint_as_float(v.a.x), __uint_as_float(v.a.y)};
  f[1] = ln_f32x2{__uint_as_float(v.a.z), __uint_as_float(v.a.w)};
  f[2] = ln_f32x2{__uint_as_float(v.b.x), __uint_as_float(v.b.y)};
  f[3] = ln_f32x2{__uint_as_float(v.b.z), __uint_as_float(v.b.w)};
}
template <typename T> __device__ __forceinline__ void ln_store8(T* p, const ln_f32x2 (&f)[4]);
template <> __device__ __forceinline__ void ln_store8<uint16_t>(uint16_t* p, const ln_f32x2 (&f)[4]) {
  *(u32x4*)p = u32x4{ln_pack2(f[0]), ln_pack2(f[1]), ln_pack2(f[2]), ln_pack2(f[3])};
}
template <> __device__ __forceinline__ void ln_store8<float>(float* p, const ln_f32x2 (&f)[4]) {
  *(f32x4*)p = f32x4{f[0].x, f[0].y, f[1].x, f[1].y};
  *(f32x4*)(p + 4) = f32x4{f[2].x, f[2].y, f[3].x, f[3].y};
}

template <int LPR, typename T = uint16_t>
__global__ void __launch_bounds__(256)
ln_act_fwd_vec_kernel(const T* __restrict__ x, int64_t n, const float* __restrict__ gamma,
                      const float* __restrict__ beta, float eps, int act,
                      T* __restrict__ y, float* __restrict__ mean_rstd, LnDropout drop) {
  constexpr int C = LPR * 8, RPB = 256 / LPR;
  const int li = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
  ln_f32x2 g[4], b[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    g[p] = ln_f32x2{gamma[li * 8 + 2 * p], gamma[li * 8 + 2 * p + 1]};
    b[p] = ln_f32x2{beta[li * 8 + 2 * p], beta[li * 8 + 2 * p + 1]};
  }
  // two rows per trip (both loads in flight), packed f32 arithmetic, GELU through ln_math.hpp
  const int64_t stride = (int64_t)gridDim.x * RPB;
  for (int64_t r0 = (int64_t)blockIdx.x * RPB + rloc; r0 < n; r0 += 2 * stride) {
    const int64_t r1 = r0 + stride;
    const bool two = r1 < n;
    const int64_t rr[2] = {r0, two ? r1 : r0};
    Ln8 xin[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) xin[u] = ln_load8<T>(x + rr[u] * C + li * 8);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      ln_f32x2 v[4];
      ln_unpack8t<T>(xin[u], v);
      const ln_f32x2 sv = (v[0] + v[1]) + (v[2] + v[3]);
      const float mean = group_sum(sv.x + sv.y, LPR) * (1.f / C);
      ln_f32x2 sq = {0.f, 0.f};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        v[p] = v[p] - mean;
        sq += v[p] * v[p];
      }
      const float rstd = rsqrtf(group_sum(sq.x + sq.y, LPR) * (1.f / C) + eps);
      ln_f32x2 zz[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        ln_f32x2 z = (v[p] * rstd) * g[p] + b[p];
        if (act == 1) z = ln_gelu2(z);
        if (drop.thr) z = z * ln_dropout_mask2(drop, rr[u], li * 4 + p, C / 2);
        zz[p] = z;
      }
      ln_store8<T>(y + rr[u] * C + li * 8, zz);
      if (mean_rstd && li == 0) {
        mean_rstd[rr[u] * 2] = mean;
        mean_rstd[rr[u] * 2 + 1] = rstd;
      }
    }
  }
}

template <int LPR, bool GELU, typename T>
__device__ __forceinline__ void ln_act_bwd_vec_body(const T* __restrict__ x, const T* __restrict__ dy, int64_t n,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean_rstd, T* __restrict__ dx,
                                                    float* __restrict__ partials, const LnDropout& drop) {
  constexpr int C = LPR * 8, RPB = 256 / LPR;
  const int li = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
  ln_f32x2 g[4], b[4], dg[4], db[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    g[p] = ln_f32x2{gamma[li * 8 + 2 * p], gamma[li * 8 + 2 * p + 1]};
    b[p] = ln_f32x2{beta[li * 8 + 2 * p], beta[li * 8 + 2 * p + 1]};
    dg[p] = db[p] = ln_f32x2{0.f, 0.f};
  }
  // two rows per trip: both rows' loads are issued before either is consumed
  const int64_t stride = (int64_t)gridDim.x * RPB;
  for (int64_t r0 = (int64_t)blockIdx.x * RPB + rloc; r0 < n; r0 += 2 * stride) {
    const int64_t r1 = r0 + stride;
    const bool two = r1 < n;
    const int64_t rr[2] = {r0, two ? r1 : r0};
    Ln8 xin[2], din[2];
    float mean[2], rstd[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      xin[u] = ln_load8<T>(x + rr[u] * C + li * 8);
      din[u] = ln_load8<T>(dy + rr[u] * C + li * 8);
      mean[u] = mean_rstd[rr[u] * 2];
      rstd[u] = mean_rstd[rr[u] * 2 + 1];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      ln_f32x2 xv[4], dv[4], dzg[4];
      ln_unpack8t<T>(xin[u], xv);
      ln_unpack8t<T>(din[u], dv);
      if (drop.thr) {
#pragma unroll
        for (int p = 0; p < 4; ++p) dv[p] = dv[p] * ln_dropout_mask2(drop, rr[u], li * 4 + p, C / 2);
      }
      float s1, s2;
      ln_bwd_piece8<GELU>(xv, dv, mean[u], rstd[u], g, b, dg, db, dzg, s1, s2);
      s1 = group_sum(s1, LPR) * (1.f / C);
      s2 = group_sum(s2, LPR) * (1.f / C);
      if (sizeof(T) == 2) {
        *(u32x4*)(dx + rr[u] * C + li * 8) = ln_bwd_finish8(xv, dzg, rstd[u], s1, s2);
      } else {
        ln_f32x2 o[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p] = ((dzg[p] - s1) - xv[p] * s2) * rstd[u];
        ln_store8<T>(dx + rr[u] * C + li * 8, o);
      }
    }
  }
  extern __shared__ __attribute__((aligned(16))) float red[];  // [RPB][2*C]
  float* mine = red + rloc * 2 * C;
  *(f32x4*)(mine + li * 8) = f32x4{dg[0].x, dg[0].y, dg[1].x, dg[1].y};
  *(f32x4*)(mine + li * 8 + 4) = f32x4{dg[2].x, dg[2].y, dg[3].x, dg[3].y};
  *(f32x4*)(mine + C + li * 8) = f32x4{db[0].x, db[0].y, db[1].x, db[1].y};
  *(f32x4*)(mine + C + li * 8 + 4) = f32x4{db[2].x, db[2].y, db[3].x, db[3].y};
  __syncthreads();
  float* slab = partials + (int64_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float s = 0.f;
#pragma unroll 4
    for (int g = 0; g < RPB; ++g) s += red[g * 2 * C + i];
    slab[i] = s;
  }
}
template <int LPR, typename T = uint16_t>
__global__ void __launch_bounds__(256)
ln_act_bwd_vec_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t n,
                      const float* __restrict__ gamma, const float* __restrict__ beta,
                      const float* __restrict__ mean_rstd, int act, T* __restrict__ dx,
                      float* __restrict__ partials, LnDropout drop) {
  if (act == 1) ln_act_bwd_vec_body<LPR, true, T>(x, dy, n, gamma, beta, mean_rstd, dx, partials, drop);
  else ln_act_bwd_vec_body<LPR, false, T>(x, dy, n, gamma, beta, mean_rstd, dx, partials, drop);
}

// Wide rows (C = 512 * VEC, e.g. the 1024-wide layers of the occupancy decoder): one wave per row, VEC
// 16-byte pieces per lane interleaved by 64 pieces so that every load of a wave is one contiguous 1 KB.
template <int VEC>
__global__ void __launch_bounds__(256)
ln_act_fwd_wide_kernel(const uint16_t* __restrict__ x, int64_t n, const float* __restrict__ gamma,
                       const float* __restrict__ beta, float eps, int act,
                       uint16_t* __restrict__ y, float* __restrict__ mean_rstd, LnDropout drop) {
  constexpr int C = 512 * VEC;
  const int li = threadIdx.x & 63, rloc = threadIdx.x >> 6;
  ln_f32x2 g[VEC][4], b[VEC][4];
#pragma unroll
  for (int u = 0; u < VEC; ++u) {
    const int ch = (u * 64 + li) * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      g[u][p] = ln_f32x2{gamma[ch + 2 * p], gamma[ch + 2 * p + 1]};
      b[u][p] = ln_f32x2{beta[ch + 2 * p], beta[ch + 2 * p + 1]};
    }
  }
  for (int64_t r = (int64_t)blockIdx.x * 4 + rloc; r < n; r += (int64_t)gridDim.x * 4) {
    ln_f32x2 v[VEC][4];
    ln_f32x2 sv = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      ln_unpack8(*(const u32x4*)(x + r * C + (u * 64 + li) * 8), v[u]);
      sv += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    }
    const float mean = group_sum(sv.x + sv.y, 64) * (1.f / C);
    ln_f32x2 sq = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < VEC; ++u)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        v[u][p] = v[u][p] - mean;
        sq += v[u][p] * v[u][p];
      }
    const float rstd = rsqrtf(group_sum(sq.x + sq.y, 64) * (1.f / C) + eps);
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      u32x4 q;
      uint32_t* qq = (uint32_t*)&q;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        ln_f32x2 z = (v[u][p] * rstd) * g[u][p] + b[u][p];
        if (act == 1) z = ln_gelu2(z);
        if (drop.thr) z = z * ln_dropout_mask2(drop, r, (u * 64 + li) * 4 + p, C / 2);
        qq[p] = ln_pack2(z);
      }
      *(u32x4*)(y + r * C + (u * 64 + li) * 8) = q;
    }
    if (mean_rstd && li == 0) {
      mean_rstd[r * 2] = mean;
      mean_rstd[r * 2 + 1] = rstd;
    }
  }
}

template <int VEC, bool GELU>
__device__ __forceinline__ void ln_act_bwd_wide_body(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, int64_t n,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean_rstd, uint16_t* __restrict__ dx,
                                                     float* __restrict__ partials, const LnDropout& drop) {
  constexpr int C = 512 * VEC;
  const int li = threadIdx.x & 63, rloc = threadIdx.x >> 6;
  // gamma / beta of the lane's channels stay in registers; packed f32 arithmetic (ln_math.hpp)
  ln_f32x2 g[VEC][4], b[VEC][4], dg[VEC][4], db[VEC][4];
#pragma unroll
  for (int u = 0; u < VEC; ++u) {
    const int ch = (u * 64 + li) * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      g[u][p] = ln_f32x2{gamma[ch + 2 * p], gamma[ch + 2 * p + 1]};
      b[u][p] = ln_f32x2{beta[ch + 2 * p], beta[ch + 2 * p + 1]};
      dg[u][p] = db[u][p] = ln_f32x2{0.f, 0.f};
    }
  }
  for (int64_t r = (int64_t)blockIdx.x * 4 + rloc; r < n; r += (int64_t)gridDim.x * 4) {
    u32x4 xin[VEC], din[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      xin[u] = *(const u32x4*)(x + r * C + (u * 64 + li) * 8);
      din[u] = *(const u32x4*)(dy + r * C + (u * 64 + li) * 8);
    }
    const float mean = mean_rstd[r * 2], rstd = mean_rstd[r * 2 + 1];
    ln_f32x2 xv[VEC][4], dzg[VEC][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      ln_f32x2 dv[4];
      ln_unpack8(xin[u], xv[u]);
      ln_unpack8(din[u], dv);
      if (drop.thr) {
#pragma unroll
        for (int p = 0; p < 4; ++p) dv[p] = dv[p] * ln_dropout_mask2(drop, r, (u * 64 + li) * 4 + p, C / 2);
      }
      float t1, t2;
      ln_bwd_piece8<GELU>(xv[u], dv, mean, rstd, g[u], b[u], dg[u], db[u], dzg[u], t1, t2);
      s1 += t1;
      s2 += t2;
    }
    s1 = group_sum(s1, 64) * (1.f / C);
    s2 = group_sum(s2, 64) * (1.f / C);
#pragma unroll
    for (int u = 0; u < VEC; ++u)
      *(u32x4*)(dx + r * C + (u * 64 + li) * 8) = ln_bwd_finish8(xv[u], dzg[u], rstd, s1, s2);
  }
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][2*C]
  float* mine = red + rloc * 2 * C;
#pragma unroll
  for (int u = 0; u < VEC; ++u) {
    const int ch = (u * 64 + li) * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      mine[ch + 2 * p] = dg[u][p].x;
      mine[ch + 2 * p + 1] = dg[u][p].y;
      mine[C + ch + 2 * p] = db[u][p].x;
      mine[C + ch + 2 * p + 1] = db[u][p].y;
    }
  }
  __syncthreads();
  float* slab = partials + (int64_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) slab[i] = (red[i] + red[2 * C + i]) + (red[4 * C + i] + red[6 * C + i]);
}
template <int VEC>
__global__ void __launch_bounds__(256)
ln_act_bwd_wide_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, int64_t n,
                       const float* __restrict__ gamma, const float* __restrict__ beta,
                       const float* __restrict__ mean_rstd, int act, uint16_t* __restrict__ dx,
                       float* __restrict__ partials, LnDropout drop) {
  if (act == 1) ln_act_bwd_wide_body<VEC, true>(x, dy, n, gamma, beta, mean_rstd, dx, partials, drop);
  else ln_act_bwd_wide_body<VEC, false>(x, dy, n, gamma, beta, mean_rstd, dx, partials, drop);
}

__global__ void __launch_bounds__(256)
ln_param_reduce_kernel(const float* __restrict__ partials, int nblocks, int c,
                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  ln_param_reduce_block(partials, nblocks, c, dgamma, dbeta, blockIdx.x);
}

__global__ void __launch_bounds__(256) ln_param_reduce_multi_kernel(LnReducePack p, int count) {
  ln_param_reduce_multi_body(p, count, (int)blockIdx.x);
}

inline int pick_lpr(int c) {
  int lpr = 1;
  while (lpr * 4 < c && lpr < 64) lpr <<= 1;  // ~4 channels per lane
  return lpr;
}
inline int bwd_blocks(int64_t n, int lpr) {
  int64_t b = ococc_cdiv(n, 256 / lpr);
  if (b > kBwdMaxBlocks) b = kBwdMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ void __launch_bounds__(256)
cast_f32_bf16_kernel(const float* __restrict__ s, uint16_t* __restrict__ d, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x)
    d[i] = ococc_f32_to_bf16(s[i]);
}
__global__ void __launch_bounds__(256)
cast_bf16_f32_kernel(const uint16_t* __restrict__ s, float* __restrict__ d, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x)
    d[i] = ococc_bf16_to_f32(s[i]);
}

inline bool vec_ok(int c) { return c % 8 == 0 && c >= 16 && c <= 512 && ((c / 8) & (c / 8 - 1)) == 0; }
inline int vec_blocks(int64_t n, int c, int cap) {
  int64_t b = ococc_cdiv(n, 256 / (c / 8));
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

#define OCOCC_LN_VEC_SWITCH(LPRV, CALL) \
  switch (LPRV) {                        \
    case 2: CALL(2); break;              \
    case 4: CALL(4); break;              \
    case 8: CALL(8); break;              \
    case 16: CALL(16); break;            \
    case 32: CALL(32); break;            \
    default: CALL(64); break;            \
  }

template <typename T>
int launch_fwd(const T* x, int64_t n, int c, const float* gamma, const float* beta, float eps,
               int act, T* y, float* mean_rstd, hipStream_t stream, LnDropout drop = LnDropout{0u, 1.f, 0u, 0u}) {
  if (vec_ok(c) && (sizeof(T) == 2 || drop.thr == 0)) {
    const int grid = vec_blocks(n, c, 4096);
#define CALL(L) hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_fwd_vec_kernel<L, T>), dim3(grid), dim3(256), 0, stream, x, n, gamma, beta, eps, act, y, mean_rstd, drop)
    OCOCC_LN_VEC_SWITCH(c / 8, CALL)
#undef CALL
    OCOCC_CHECK_LAUNCH();
    return OCOCC_OK;
  }
  if (sizeof(T) == 2 && (c == 1024 || c == 1536 || c == 2048)) {
    const int grid = (int)(ococc_cdiv(n, 4) < 8192 ? ococc_cdiv(n, 4) : 8192);
#define CALLW(V) hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_fwd_wide_kernel<V>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)x, n, gamma, beta, eps, act, (uint16_t*)y, mean_rstd, drop)
    if (c == 1024) CALLW(2); else if (c == 1536) CALLW(3); else CALLW(4);
#undef CALLW
    OCOCC_CHECK_LAUNCH();
    return OCOCC_OK;
  }
  if (drop.thr) return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "fused dropout: bf16 rows of 16..512 (x8) or 1024/1536/2048 channels");
  if (row_kernel_ok(n, c)) {
    if (n > 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_fwd_row_kernel<T>), dim3((unsigned)n), dim3(256), 0, stream, x, c, gamma, beta, eps,
                         act, y, mean_rstd);
    OCOCC_CHECK_LAUNCH();
    return OCOCC_OK;
  }
  const int lpr = pick_lpr(c);
  const int vpl = (int)ococc_cdiv(c, lpr);
  const int grid = ococc_grid_1d(ococc_cdiv(n, 256 / lpr) * 256, 256);
  if (vpl <= 4)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_fwd_kernel<T, 4>), dim3(grid), dim3(256), 0, stream, x, n, c, lpr, gamma, beta, eps, act, y, mean_rstd);
  else if (vpl <= 8)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_fwd_kernel<T, 8>), dim3(grid), dim3(256), 0, stream, x, n, c, lpr, gamma, beta, eps, act, y, mean_rstd);
  else if (vpl <= 32)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_fwd_kernel<T, 32>), dim3(grid), dim3(256), 0, stream, x, n, c, lpr, gamma, beta, eps, act, y, mean_rstd);
  else
    return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "c must be <= 2048");
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

// rows of the partials slab = blocks of the backward kernel (same choice as launch_bwd below)
inline int bwd_partial_rows(int64_t n, int c, bool two_byte) {
  if (vec_ok(c)) return vec_blocks(n, c, kBwdMaxBlocks);
  if (two_byte && (c == 1024 || c == 1536 || c == 2048)) {
    int64_t gb = ococc_cdiv(n, 4);
    return (int)(gb > kBwdMaxBlocks ? kBwdMaxBlocks : (gb < 1 ? 1 : gb));
  }
  if (row_kernel_ok(n, c)) return (int)(n < 1 ? 1 : n);   // one partial row per input row
  return bwd_blocks(n, pick_lpr(c));
}

template <typename T>
int launch_bwd(const T* x, const T* dy, int64_t n, int c, const float* gamma, const float* beta,
               const float* mean_rstd, int act, T* dx, float* dgamma, float* dbeta, float* partials,
               hipStream_t stream, LnDropout drop = LnDropout{0u, 1.f, 0u, 0u}) {
  if (vec_ok(c) && (sizeof(T) == 2 || drop.thr == 0)) {
    const int grid = bwd_partial_rows(n, c, sizeof(T) == 2);
#define CALL(L) hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_bwd_vec_kernel<L, T>), dim3(grid), dim3(256), (256 / L) * 2 * (L * 8) * 4, stream, x, dy, n, gamma, beta, mean_rstd, act, dx, partials, drop)
    OCOCC_LN_VEC_SWITCH(c / 8, CALL)
#undef CALL
    OCOCC_CHECK_LAUNCH();
    if (!dgamma && !dbeta) return OCOCC_OK;  // partials only: ococc_layernorm_param_reduce_multi finishes them
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * c + 7) / 8), dim3(256), 0, stream, partials,
                       grid, c, dgamma, dbeta);
    OCOCC_CHECK_LAUNCH();
    return OCOCC_OK;
  }
  if (sizeof(T) == 2 && (c == 1024 || c == 1536 || c == 2048)) {
    const int grid = bwd_partial_rows(n, c, true);
#define CALLW(V) hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_bwd_wide_kernel<V>), dim3(grid), dim3(256), 4 * 2 * c * 4, stream, (const uint16_t*)x, (const uint16_t*)dy, n, gamma, beta, mean_rstd, act, (uint16_t*)dx, partials, drop)
    if (c == 1024) CALLW(2); else if (c == 1536) CALLW(3); else CALLW(4);
#undef CALLW
    OCOCC_CHECK_LAUNCH();
    if (!dgamma && !dbeta) return OCOCC_OK;
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * c + 7) / 8), dim3(256), 0, stream, partials, grid, c,
                       dgamma, dbeta);
    OCOCC_CHECK_LAUNCH();
    return OCOCC_OK;
  }
  if (drop.thr) return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "fused dropout: bf16 rows of 16..512 (x8) or 1024/1536/2048 channels");
  if (row_kernel_ok(n, c)) {
    const int rows = bwd_partial_rows(n, c, sizeof(T) == 2);   // (= n)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_bwd_row_kernel<T>), dim3((unsigned)n), dim3(256), 0, stream, x, dy, c, gamma, beta,
                       mean_rstd, act, dx, partials);
    OCOCC_CHECK_LAUNCH();
    if (!dgamma && !dbeta) return OCOCC_OK;
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * c + 7) / 8), dim3(256), 0, stream, partials, rows, c, dgamma, dbeta);
    OCOCC_CHECK_LAUNCH();
    return OCOCC_OK;
  }
  const int lpr = pick_lpr(c);
  const int vpl = (int)ococc_cdiv(c, lpr);
  const int grid = bwd_blocks(n, lpr);
  if (vpl <= 4)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_bwd_kernel<T, 4>), dim3(grid), dim3(256), (256 / lpr) * 2 * c * 4, stream, x, dy, n, c, lpr, gamma, beta, mean_rstd, act, dx, partials);
  else if (vpl <= 8)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_bwd_kernel<T, 8>), dim3(grid), dim3(256), (256 / lpr) * 2 * c * 4, stream, x, dy, n, c, lpr, gamma, beta, mean_rstd, act, dx, partials);
  else if (vpl <= 32)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ln_act_bwd_kernel<T, 32>), dim3(grid), dim3(256), (256 / lpr) * 2 * c * 4, stream, x, dy, n, c, lpr, gamma, beta, mean_rstd, act, dx, partials);
  else
    return ococc_fail(OCOCC_EUNSUPPORTED, __func__, "c must be <= 2048");
  OCOCC_CHECK_LAUNCH();
  if (!dgamma && !dbeta) return OCOCC_OK;
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * c + 7) / 8), dim3(256), 0,
                     stream, partials, grid, c, dgamma, dbeta);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

}  // namespace

extern "C" int ococc_layernorm_act_fwd(const void* x, int64_t n, int32_t c, const float* gamma,
                                       const float* beta, float eps, int32_t act, void* y,
                                       float* mean_rstd, int32_t dtype, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n >= 0 && c >= 1, "bad sizes");
  OCOCC_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (gelu)");
  OCOCC_REQUIRE(dtype == OCOCC_F32 || dtype == OCOCC_BF16, "dtype must be f32/bf16");
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(x && y && gamma && beta, "null pointer");
  if (dtype == OCOCC_F32)
    return launch_fwd<float>((const float*)x, n, c, gamma, beta, eps, act, (float*)y, mean_rstd, stream);
  return launch_fwd<uint16_t>((const uint16_t*)x, n, c, gamma, beta, eps, act, (uint16_t*)y, mean_rstd, stream);
}

extern "C" int64_t ococc_layernorm_act_bwd_workspace_bytes(int64_t n, int32_t c) {
  if (n < 0 || c < 1) return -1;
  return (int64_t)kBwdMaxBlocks * 2 * c * (int64_t)sizeof(float);  // one slab per block
}

extern "C" int ococc_layernorm_act_bwd(const void* x, const void* dy, int64_t n, int32_t c,
                                       const float* gamma, const float* beta,
                                       const float* mean_rstd, int32_t act, void* dx,
                                       float* dgamma, float* dbeta, int32_t dtype, void* workspace,
                                       int64_t workspace_bytes, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n >= 0 && c >= 1, "bad sizes");
  OCOCC_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (gelu)");
  OCOCC_REQUIRE(dtype == OCOCC_F32 || dtype == OCOCC_BF16, "dtype must be f32/bf16");
  if (n == 0) {
    if (dgamma) OCOCC_HIP(hipMemsetAsync(dgamma, 0, (size_t)c * sizeof(float), stream));
    if (dbeta) OCOCC_HIP(hipMemsetAsync(dbeta, 0, (size_t)c * sizeof(float), stream));
    return OCOCC_OK;
  }
  OCOCC_REQUIRE(x && dy && dx && gamma && beta && mean_rstd, "null pointer");
  OCOCC_REQUIRE(workspace && workspace_bytes >= ococc_layernorm_act_bwd_workspace_bytes(n, c),
                "workspace too small");
  if (dtype == OCOCC_F32)
    return launch_bwd<float>((const float*)x, (const float*)dy, n, c, gamma, beta, mean_rstd, act,
                             (float*)dx, dgamma, dbeta, (float*)workspace, stream);
  return launch_bwd<uint16_t>((const uint16_t*)x, (const uint16_t*)dy, n, c, gamma, beta, mean_rstd,
                              act, (uint16_t*)dx, dgamma, dbeta, (float*)workspace, stream);
}

namespace {
inline LnDropout make_dropout(uint32_t keep_threshold, uint64_t seed) {
  return LnDropout{keep_threshold, 65536.f / (65536.f - (float)keep_threshold), (uint32_t)seed, (uint32_t)(seed >> 32)};
}
}  // namespace

extern "C" int ococc_layernorm_act_dropout_fwd_bf16(const uint16_t* x, int64_t n, int32_t c, const float* gamma,
                                                    const float* beta, float eps, int32_t act,
                                                    uint32_t drop_threshold, uint64_t seed, uint16_t* y,
                                                    float* mean_rstd, ococc_stream_t stream_) {
  OCOCC_REQUIRE(n >= 0 && c >= 1, "bad sizes");
  OCOCC_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (gelu)");
  OCOCC_REQUIRE(drop_threshold < 65536u, "drop_threshold = round(p * 65536) must be below 65536");
  if (n == 0) return OCOCC_OK;
  OCOCC_REQUIRE(x && y && gamma && beta, "null pointer");
  return launch_fwd<uint16_t>(x, n, c, gamma, beta, eps, act, y, mean_rstd, (hipStream_t)stream_,
                              make_dropout(drop_threshold, seed));
}

extern "C" int ococc_layernorm_act_dropout_bwd_bf16(const uint16_t* x, const uint16_t* dy, int64_t n, int32_t c,
                                                    const float* gamma, const float* beta, const float* mean_rstd,
                                                    int32_t act, uint32_t drop_threshold, uint64_t seed, uint16_t* dx,
                                                    float* dgamma, float* dbeta, void* workspace,
                                                    int64_t workspace_bytes, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n >= 0 && c >= 1, "bad sizes");
  OCOCC_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (gelu)");
  OCOCC_REQUIRE(drop_threshold < 65536u, "drop_threshold = round(p * 65536) must be below 65536");
  if (n == 0) {
    if (dgamma) OCOCC_HIP(hipMemsetAsync(dgamma, 0, (size_t)c * sizeof(float), stream));
    if (dbeta) OCOCC_HIP(hipMemsetAsync(dbeta, 0, (size_t)c * sizeof(float), stream));
    return OCOCC_OK;
  }
  OCOCC_REQUIRE(x && dy && dx && gamma && beta && mean_rstd, "null pointer");
  OCOCC_REQUIRE(workspace && workspace_bytes >= ococc_layernorm_act_bwd_workspace_bytes(n, c), "workspace too small");
  return launch_bwd<uint16_t>(x, dy, n, c, gamma, beta, mean_rstd, act, dx, dgamma, dbeta, (float*)workspace, stream,
                              make_dropout(drop_threshold, seed));
}

extern "C" int32_t ococc_layernorm_act_bwd_partial_rows(int64_t n, int32_t c, int32_t dtype) {
  if (n < 1 || c < 1 || (dtype != OCOCC_F32 && dtype != OCOCC_BF16)) return 0;
  return bwd_partial_rows(n, c, dtype == OCOCC_BF16);
}

extern "C" int ococc_layernorm_param_reduce_multi(int32_t count, const void* const* partials,
                                                  const int32_t* rows, const int32_t* c, void* const* dgamma,
                                                  void* const* dbeta, ococc_stream_t stream_) {
  OCOCC_REQUIRE(count >= 0 && count <= kLnMultiMax, "count must be in [0, 16]");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(partials && rows && c && dgamma && dbeta, "null pointer");
  LnReducePack p;
  int blocks = 0;
  for (int j = 0; j < count; ++j) {
    OCOCC_REQUIRE(partials[j] && rows[j] >= 1 && c[j] >= 1, "bad layer");
    p.partials[j] = (const float*)partials[j];
    p.dgamma[j] = (float*)dgamma[j];
    p.dbeta[j] = (float*)dbeta[j];
    p.rows[j] = rows[j];
    p.c[j] = c[j];
    p.first[j] = blocks;
    blocks += (2 * c[j] + 7) / 8;
  }
  p.first[count] = blocks;
  hipLaunchKernelGGL(ln_param_reduce_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, p, count);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_cast_f32_to_bf16(const float* src, uint16_t* dst, int64_t count,
                                      ococc_stream_t stream_) {
  OCOCC_REQUIRE(count >= 0, "negative count");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(src && dst, "null pointer");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ococc_grid_1d(count, 256)), dim3(256), 0,
                     (hipStream_t)stream_, src, dst, count);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_cast_bf16_to_f32(const uint16_t* src, float* dst, int64_t count,
                                      ococc_stream_t stream_) {
  OCOCC_REQUIRE(count >= 0, "negative count");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(src && dst, "null pointer");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ococc_grid_1d(count, 256)), dim3(256), 0,
                     (hipStream_t)stream_, src, dst, count);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
