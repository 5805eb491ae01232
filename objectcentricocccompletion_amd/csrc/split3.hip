// f32 matrix products on the bf16 matrix cores at f32-level accuracy: the operand split.
//
// The f32 GEMMs of configs[2] -- the temporal transformer and the RoI-level MLPs (mmdet3d/models/occ/layers.py:35-87,
// ococc_bbox_head.py:116-193,849-908: nn.Linear / nn.MultiheadAttention in f32) -- run on the f32 matrix instructions
// (v_mfma_f32_16x16x4_f32: 132 TFLOP/s chip-wide, a twentieth of the bf16 rate); at 64 tracklets per GPU they are 17 of
// the 55 ms of kernels in a step.  Every f32 value x is hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x|,
// so   x w  =  hi_x hi_w + hi_x lo_w + lo_x hi_w  +  O(2^-16 |x w|)
// and a product of two f32 matrices is ONE bf16 GEMM with f32 accumulation over a contraction three times as long:
//   [x_hi | x_hi | x_lo] [w_hi | w_lo | w_hi]^T.    (measured: 4.5e-6 relative to f64 where the f32 GEMM has 7e-7 and plain
//   bf16 operands 2.3e-3; 2.3-2.8 x faster than the f32 library GEMM from 1 024 rows on -- tools/probe/gemm_x3_probe.py)
// This file is the split: one pass over an f32 matrix writes the three-part bf16 operand in either or both of the two
// forms a GEMM wants -- the parts side by side along the contraction ([rows, 3 cols]: the matrix is the row-major operand
// whose rows are contracted) or stacked ([3 rows, cols]: its columns are).  objectcentricocccompletion_amd/gemm.py holds
// the products (library bf16 GEMMs, f32 output) and the per-parameter cache of the weights' operands.
#include "common.hpp"

namespace {

// part p of pattern `pat` is hi (0) or lo (1):  pattern 0 = (hi, hi, lo), pattern 1 = (hi, lo, hi)
__device__ __forceinline__ bool part_is_lo(int pat, int p) { return pat == 0 ? p == 2 : p == 1; }

__global__ void __launch_bounds__(256)
split3_kernel(const float* __restrict__ src, int64_t rows, int64_t cols, int64_t ld, uint16_t* __restrict__ cat, int cat_pat,
              uint16_t* __restrict__ stack, int stack_pat) {
  const int64_t pieces_per_row = cols >> 3;   // 8 elements per thread
  const int64_t total = rows * pieces_per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / pieces_per_row, c = (i - r * pieces_per_row) << 3;
    const f32x4 a = *(const f32x4*)(src + r * ld + c), b = *(const f32x4*)(src + r * ld + c + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = v[j] - ococc_bf16_to_f32(ococc_f32_to_bf16(v[j]));   // exact in f32
    u32x4 hi, lo;
    hi.x = ococc_pack_bf16x2(v[0], v[1]); hi.y = ococc_pack_bf16x2(v[2], v[3]);
    hi.z = ococc_pack_bf16x2(v[4], v[5]); hi.w = ococc_pack_bf16x2(v[6], v[7]);
    lo.x = ococc_pack_bf16x2(l[0], l[1]); lo.y = ococc_pack_bf16x2(l[2], l[3]);
    lo.z = ococc_pack_bf16x2(l[4], l[5]); lo.w = ococc_pack_bf16x2(l[6], l[7]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      if (cat) *(u32x4*)(cat + r * 3 * cols + p * cols + c) = part_is_lo(cat_pat, p) ? lo : hi;
      if (stack) *(u32x4*)(stack + ((int64_t)p * rows + r) * cols + c) = part_is_lo(stack_pat, p) ? lo : hi;
    }
  }
}

}  // namespace

extern "C" int ococc_split3_bf16(const float* src, int64_t rows, int64_t cols, int64_t ld_src, uint16_t* cat_cols,
                                 int32_t cat_pattern, uint16_t* stack_rows, int32_t stack_pattern, ococc_stream_t stream_) {
  OCOCC_REQUIRE(rows >= 0 && cols >= 0 && ld_src >= cols, "bad sizes");
  OCOCC_REQUIRE(cols % 8 == 0 && ld_src % 4 == 0, "columns in multiples of 8, rows 16-byte aligned");
  OCOCC_REQUIRE((cat_pattern == 0 || cat_pattern == 1) && (stack_pattern == 0 || stack_pattern == 1),
                "pattern 0 = (hi, hi, lo), 1 = (hi, lo, hi)");
  OCOCC_REQUIRE(cat_cols || stack_rows, "no output");
  if (rows == 0 || cols == 0) return OCOCC_OK;
  OCOCC_REQUIRE(src && ((uintptr_t)src & 15) == 0 && ((uintptr_t)cat_cols & 15) == 0 && ((uintptr_t)stack_rows & 15) == 0,
                "null or misaligned pointer");
  const int64_t pieces = rows * (cols >> 3);
  hipLaunchKernelGGL(split3_kernel, dim3(ococc_grid_1d(pieces, 256, 4096)), dim3(256), 0, (hipStream_t)stream_, src, rows, cols,
                     ld_src, cat_cols, (int)cat_pattern, stack_rows, (int)stack_pattern);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
