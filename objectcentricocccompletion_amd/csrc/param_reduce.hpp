// Column sums of the LayerNorm backward's per-workgroup partial rows -> d gamma / d beta.  Shared by the LN-only launch
// (layernorm_act.hip) and the joint end-of-backward launch that also finishes the weight-gradient slabs
// (sparse_conv.hip: ococc_backward_param_reduce_multi).
#pragma once
#include "common.hpp"

namespace {

// 8 outputs per block, 32 lanes each: lane l adds partials l, l+32, ... (fixed order),
// then a fixed-shape butterfly combines the 32 lanes -> deterministic.
__device__ __forceinline__ void ln_param_reduce_block(const float* __restrict__ partials, int nblocks, int c,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      int block) {
  const int i = block * 8 + (threadIdx.x >> 5);
  const int l = threadIdx.x & 31;
  // 32 running sums per lane (fixed combination order): ONE round of independent loads for up to 1 024 partial rows.
  // (One sum made a lane's strided loads a chain of dependent L2 round trips, 6.8 us per call whatever the width; eight
  // sums left four dependent rounds, and inside the joint end-of-backward launch -- 2 048 workgroups reading slabs beside
  // these few -- every round took ~3 us: the launch lasted 15.5 us for 10.6 us of slab sums.)
  constexpr int U = 32;
  float a[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = 0.f;
  if (i < 2 * c) {
    const float* src = partials + i;
    int b = l;
    for (; b + 32 * (U - 1) < nblocks; b += 32 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] += src[(int64_t)(b + 32 * u) * 2 * c];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (b + 32 * u < nblocks) a[u] += src[(int64_t)(b + 32 * u) * 2 * c];
  }
#pragma unroll
  for (int w = U / 2; w >= 8; w >>= 1)   // (32 -> 8 sums: row j of the former eight-sum form is a[j] + a[j + 8] + ...)
#pragma unroll
    for (int u = 0; u < w; ++u) a[u] += a[u + w];
  float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) s += __shfl_xor(s, d, 32);
  if (l == 0 && i < 2 * c) {
    if (i < c) {
      if (dgamma) dgamma[i] = s;
    } else {
      if (dbeta) dbeta[i - c] = s;
    }
  }
}

// The reductions of several layers in one launch: layer j owns blocks [first[j], first[j+1]).
constexpr int kLnMultiMax = 16;
struct LnReducePack {
  const float* partials[kLnMultiMax];
  float* dgamma[kLnMultiMax];
  float* dbeta[kLnMultiMax];
  int rows[kLnMultiMax];
  int c[kLnMultiMax];
  int first[kLnMultiMax + 1];
};
__device__ __forceinline__ void ln_param_reduce_multi_body(const LnReducePack& p, int count, int block) {
  int j = 0;
  while (j + 1 < count && block >= p.first[j + 1]) ++j;
  ln_param_reduce_block(p.partials[j], p.rows[j], p.c[j], p.dgamma[j], p.dbeta[j], block - p.first[j]);
}

}  // namespace
