// Index maps of the bf16 operand layouts of the convolution weights, shared by the preparation kernels
// (sparse_conv.hip) and the optimizer step that refreshes the operands in place (adamw.hip).
#pragma once
#include <stdint.h>

// Destination index of element (k, r, c) of the prepared matrix wn[k][ncols][kd] whose row-major position is i.  Bit 2
// of the mode (value 4) asks for the MFMA-FRAGMENT-major order of the tile / pull kernels, whose waves load their weight
// fragments straight from global memory: fragment (column block r/16, k-step c/32) is 64 lanes x 16 bytes, lane =
// 16 * ((c % 32) / 8) + r % 16, so that one load instruction reads 1 KB of consecutive bytes.
__host__ __device__ __forceinline__ int64_t ococc_prep_dest(int mode, int64_t i, int k, int r, int c, int ncols, int kd) {
  if (!(mode & 4)) return i;
  const int nb = ncols / 16, ksteps = kd / 32;
  const int lane = 16 * ((c % 32) / 8) + (r % 16);
  return ((((int64_t)k * nb + r / 16) * ksteps + c / 32) * 64 + lane) * 8 + (c % 8);
}

// The same map from the SOURCE side: where element `src` of W[kvol][cin][cout] (row-major) lands in the operand of
// `mode` (mode & 3: 0 forward wn[k][co][ci]; 1 sub-manifold input gradient, offsets mirrored; 2 generic input gradient).
__host__ __device__ __forceinline__ int64_t ococc_operand_index(int mode, int kvol, int cin, int cout, int64_t src) {
  const int64_t kc = (int64_t)cin * cout;
  const int ks = (int)(src / kc);
  const int rem = (int)(src - ks * kc);
  const int ci = rem / cout, co = rem - ci * cout;
  const int base = mode & 3;
  if (base == 0) return ococc_prep_dest(mode, (int64_t)ks * kc + (int64_t)co * cin + ci, ks, co, ci, cout, cin);
  const int k = base == 1 ? kvol - 1 - ks : ks;
  return ococc_prep_dest(mode, (int64_t)k * kc + rem, k, ci, co, cin, cout);
}
