// Sparse max pooling over a rulebook (mmdet3d/ops/spconv: pool.py:20-84, ops.py:162-184, include/spconv/maxpool.h,
// src/maxpool.cc:9-55 -- the reference's own semantics, which are NOT those of a dense max pool: the output starts at
// ZERO and takes an input value only where that is larger, so an output site whose inputs are all negative holds 0;
// the backward pass hands an output's gradient to EVERY input equal to it, pair by pair).
//
// Both directions are output-stationary over the offset-major gather tables the convolutions use
// (ococc_rulebook_pairs_to_table): forward out[o] = max(0, x[table[k][o]] for k), backward
// dx[i] = sum over k in ascending order of dy[o] where o = table_bwd[k][i] and out[o] == x[i] -- no atomics, the
// summation order of the reference's CPU functor (offsets ascending), bit for bit in f32.
#include "common.hpp"

namespace {

template <typename T> __device__ __forceinline__ float pool_load(const T* p);
template <> __device__ __forceinline__ float pool_load<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float pool_load<uint16_t>(const uint16_t* p) { return ococc_bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void pool_store(T* p, float v);
template <> __device__ __forceinline__ void pool_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void pool_store<uint16_t>(uint16_t* p, float v) { *p = ococc_f32_to_bf16(v); }

template <typename T>
__global__ void __launch_bounds__(256)
maxpool_fwd_kernel(const T* __restrict__ x, const int32_t* __restrict__ table, int kvol, int64_t n_out, int c,
                   T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out * c) return;
  const int64_t o = i / c;
  const int ch = (int)(i - o * c);
  float v = 0.f;   // (the reference's torch::zeros output, pool_ops.h:34)
  for (int k = 0; k < kvol; ++k) {
    const int32_t r = table[(int64_t)k * n_out + o];
    if (r >= 0) {
      const float u = pool_load(x + (int64_t)r * c + ch);
      if (v < u) v = u;
    }
  }
  pool_store(out + i, v);
}

template <typename T>
__global__ void __launch_bounds__(256)
maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ out, const T* __restrict__ dout,
                   const int32_t* __restrict__ table_bwd, int kvol, int64_t n_in, int c, T* __restrict__ din) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_in * c) return;
  const int64_t r = i / c;
  const int ch = (int)(i - r * c);
  const float mine = pool_load(x + i);
  float g = 0.f;
  for (int k = 0; k < kvol; ++k) {
    const int32_t o = table_bwd[(int64_t)k * n_in + r];
    if (o >= 0 && pool_load(out + (int64_t)o * c + ch) == mine) g += pool_load(dout + (int64_t)o * c + ch);
  }
  pool_store(din + i, g);
}

}  // namespace

extern "C" int ococc_indice_maxpool(const void* features, int32_t dtype, int64_t n_in, int32_t channels,
                                    const int32_t* table, int32_t kvol, int64_t n_out, void* out, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n_in >= 0 && n_out >= 0 && channels >= 1 && kvol >= 1, "bad sizes");
  OCOCC_REQUIRE(dtype == OCOCC_F32 || dtype == OCOCC_BF16, "dtype must be f32/bf16");
  if (n_out == 0) return OCOCC_OK;
  OCOCC_REQUIRE(table && out && (features || n_in == 0), "null pointer");
  const unsigned grid = (unsigned)ococc_cdiv(n_out * channels, 256);
  if (dtype == OCOCC_F32)
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)features, table, (int)kvol,
                       n_out, (int)channels, (float*)out);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<uint16_t>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)features, table,
                       (int)kvol, n_out, (int)channels, (uint16_t*)out);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_indice_maxpool_backward(const void* features, const void* out_features, const void* out_bp,
                                             int32_t dtype, int64_t n_in, int32_t channels, const int32_t* table_bwd,
                                             int32_t kvol, int64_t n_out, void* input_bp, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(n_in >= 0 && n_out >= 0 && channels >= 1 && kvol >= 1, "bad sizes");
  OCOCC_REQUIRE(dtype == OCOCC_F32 || dtype == OCOCC_BF16, "dtype must be f32/bf16");
  if (n_in == 0) return OCOCC_OK;
  OCOCC_REQUIRE(features && table_bwd && input_bp && ((out_features && out_bp) || n_out == 0), "null pointer");
  const unsigned grid = (unsigned)ococc_cdiv(n_in * channels, 256);
  if (dtype == OCOCC_F32)
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)features,
                       (const float*)out_features, (const float*)out_bp, table_bwd, (int)kvol, n_in, (int)channels,
                       (float*)input_bp);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<uint16_t>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)features,
                       (const uint16_t*)out_features, (const uint16_t*)out_bp, table_bwd, (int)kvol, n_in, (int)channels,
                       (uint16_t*)input_bp);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
