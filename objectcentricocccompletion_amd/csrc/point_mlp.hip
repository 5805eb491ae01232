// A6, fused: one per-point layer of the SIR encoders,  y = act(LN(W x)),  with the input row assembled on the fly and
// the segment maximum of y taken in the same launch.  Replaces, per layer of SIRLayer.forward
// (mmdet3d/models/voxel_encoders/voxel_encoder.py:764-832) and of build_mlp's rel_mlp (mmdet3d/ops/sst/sst_ops.py:333-360):
//   torch.cat / the element-wise products that build the layer input   (voxel_encoder.py:779-805, 818-820)
//   nn.Linear(bias=False) -> LayerNorm(eps 1e-3) -> GELU               (voxel_encoders/utils.py:174-189)
//   scatter_v2(mode='max') = torch_scatter.scatter_max                 (sst_ops.py:150-181)
//   voxel feature back to points: voxel_feats[unq_inv]                 (voxel_encoder.py:756-758, 818)
// The input of a row is  x = [ a (*) mul (*) colscale | b * bscale | v[inv] ]  (any part may be absent):
//   rel_mlp layer 0:  a = f_cluster, colscale = 1 / rel_dist_scaler
//   vfe layer 0:      a = point features, mul = rel_mlp output, colscale = (1 / xyz_normalizer, 1, 1, ...);  b = f_cluster
//                     (with_cluster_center: b * 1 / 10)
//   vfe layer 1:      a = point features of layer 0, v = their segment maxima, inv = the point's segment
// Everything is f32, as the reference computes these layers (force_fp32, voxel_encoder.py:764): the GEMM runs on the
// f32 matrix instruction v_mfma_f32_16x16x4_f32 (157 TFLOP/s peak; the 12 layers of configs[2] are 0.16 TFLOP).
// A workgroup owns 64 consecutive rows: x is built once in LDS ([64][K + 2] floats: the +2 keeps the 16 rows of an MFMA
// operand read on different banks), out^T = W x^T with the weight as the A operand (16 output channels x 4 k, fragment
// order, streamed from L2 eight k-steps ahead) and the rows as the B operand, so a lane ends with 4 consecutive
// channels of one row.  The 4 waves split the output channels; LayerNorm sums cross the waves through LDS.
// Rows arrive sorted by segment (the pooling order, csrc/point_pool.hip): the segment maximum is a run-length walk
// over the tile per channel and one integer atomic per (run, channel) -- exact and order independent.
// Backward (point_mlp_bwd_kernel) recomputes the layer from the same inputs, routes the gradient of the segment maxima
// to the arg-max rows (smallest row index among equals, ococc_segment_argmax), and returns dz (for dW = dz^T x), the
// assembled x (same purpose), the gradients of a, mul and b, and adds the gradient of v with float atomics at run ends.
#include "point_mlp_tile.hpp"
#include "sir_fused.hpp"

namespace {

template <int NBW, int MB>
__global__ void __launch_bounds__(kT, 2)
point_mlp_fwd_kernel(PointMlpIn in, const float* __restrict__ wf, int n, const float* __restrict__ ln_w,
                     const float* __restrict__ ln_b, float eps, int act, float* __restrict__ y, float* __restrict__ vmax) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  point_mlp_fwd_tile<NBW, MB>(in, wf, n, ln_w, ln_b, eps, act, y, vmax, Tile{(int64_t)blockIdx.x, (int64_t)blockIdx.x * 16 * MB, smem_f, (int)threadIdx.x});
}

template <int NBW, int KBW, int MB>
__global__ void __launch_bounds__(kT, 2)
point_mlp_bwd_kernel(PointMlpIn in, const float* __restrict__ wf, const float* __restrict__ wtf, int n,
                     const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps, int act,
                     const float* __restrict__ dy, const float* __restrict__ dvmax, const int32_t* __restrict__ arg,
                     float* __restrict__ dz_out, float* __restrict__ xcat, float* __restrict__ da, float* __restrict__ dmul,
                     float* __restrict__ db, float* __restrict__ dv, float* __restrict__ ln_partial) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  point_mlp_bwd_tile<NBW, KBW, MB>(in, wf, wtf, n, ln_w, ln_b, eps, act, dy, n, dvmax, n, nullptr, arg, dz_out, xcat, da, dmul, db, dv,
                                   ln_partial, Tile{(int64_t)blockIdx.x, (int64_t)blockIdx.x * 16 * MB, smem_f, (int)threadIdx.x});
}

// W [n][k] f32 (element strides) -> f32 MFMA A-operand fragments [ceil(n/16)][pad_k(k)/4][64]: lane 16 g + r of block
// (nb, ks) holds W[16 nb + r][4 ks + g]; zero where the matrix ends
__global__ void __launch_bounds__(256)
point_mlp_pack_kernel(const float* __restrict__ w, int n, int k, int64_t rs, int64_t cs, float* __restrict__ dst) {
  const int ksteps = pad_k(k) >> 2, nblocks = (n + 15) >> 4, total = nblocks * ksteps * 64;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, blk = i >> 6, ks = blk % ksteps, nb = blk / ksteps;
    const int row = 16 * nb + (lane & 15), col = 4 * ks + (lane >> 4);
    dst[i] = (row < n && col < k) ? w[row * rs + col * cs] : 0.f;
  }
}

// the same for up to kMaxPack matrices in one launch (a SIR layer packs its four or five weights, both orientations, once
// per optimizer step: 120 launches of 4 us per step otherwise)
constexpr int kMaxPack = 32;
struct PackMulti {
  const float* w[kMaxPack];
  float* dst[kMaxPack];
  int64_t rs[kMaxPack], cs[kMaxPack];
  int32_t n[kMaxPack], k[kMaxPack], first_block[kMaxPack + 1];
  int32_t count;
};
__global__ void __launch_bounds__(256) point_mlp_pack_multi_kernel(PackMulti pk) {
  int t = 0;
  while (t + 1 < pk.count && (int)blockIdx.x >= pk.first_block[t + 1]) ++t;
  const int n = pk.n[t], k = pk.k[t];
  const int ksteps = pad_k(k) >> 2, nblocks = (n + 15) >> 4, total = nblocks * ksteps * 64;
  const int i = ((int)blockIdx.x - pk.first_block[t]) * 256 + threadIdx.x;
  if (i >= total) return;
  const int lane = i & 63, blk = i >> 6, ks = blk % ksteps, nb = blk / ksteps;
  const int row = 16 * nb + (lane & 15), col = 4 * ks + (lane >> 4);
  pk.dst[t][i] = (row < n && col < k) ? pk.w[t][row * pk.rs[t] + col * pk.cs[t]] : 0.f;
}

// arg[seg][ch] = smallest row whose y equals the segment maximum (the rule of the reference's own DynamicScatter,
// scatter_points_cuda.cu:136-160; torch_scatter's tie rule is unpinned): rows are sorted by segment, so the
// first hit of a run is its smallest row; runs of one segment in different tiles meet in an integer atomicMin.
__global__ void __launch_bounds__(256)
segment_argmax_kernel(const float* __restrict__ y, const float* __restrict__ vmax, const int32_t* __restrict__ inv,
                      int64_t rows, int n, int32_t* __restrict__ arg) {
  // one thread per (row, 4 channels): a row that attains its segment's maximum in a channel proposes itself; the smallest
  // proposal stands.  (Few rows per segment attain it: the atomics are rare and spread over all segments.)
  const int q = (n + 3) >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * q) return;
  const int64_t row = i / q;
  const int c0 = (int)(i - row * q) * 4;
  const int seg = inv[row];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ch = c0 + c;
    if (ch < n && y[row * n + ch] == vmax[(int64_t)seg * n + ch]) atomicMin(arg + (int64_t)seg * n + ch, (int32_t)row);
  }
}

__global__ void __launch_bounds__(256)
point_mlp_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ xc, int64_t rows, int n, int k,
                       int64_t rows_per_slice, float* __restrict__ partial) {
  __shared__ float lds[kWgradLdsFloats];
  point_mlp_wgrad_tile(dz, xc, rows, n, k, rows_per_slice, partial, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, lds, (int)threadIdx.x);
}

// the products of SEVERAL layers over the same rows in one launch (the five layers of a SIR layer's backward: 60 of the
// 1.5 k launches of a 4-tracklet step were these, 13 us each for a dozen workgroups of work)
constexpr int kWgradMulti = 24;
struct PointWgradPack {
  const float* dz[kWgradMulti];
  const float* xc[kWgradMulti];
  float* partial[kWgradMulti];
  int32_t n[kWgradMulti], k[kWgradMulti], first[kWgradMulti + 1];
  int32_t count;
};
__global__ void __launch_bounds__(256)
point_mlp_wgrad_multi_kernel(PointWgradPack pk, int64_t rows, int64_t rows_per_slice, int slices) {
  __shared__ float lds[kWgradLdsFloats];
  int j = 0;
  while (j + 1 < pk.count && (int)blockIdx.x >= pk.first[j + 1]) ++j;
  const int local = (int)blockIdx.x - pk.first[j];
  const int tiles_n = (pk.n[j] + 63) / 64;
  const int slice = local % slices, t = local / slices;
  point_mlp_wgrad_tile(pk.dz[j], pk.xc[j], rows, pk.n[j], pk.k[j], rows_per_slice, pk.partial[j], slice, t % tiles_n, t / tiles_n, lds, (int)threadIdx.x);
}

__global__ void __launch_bounds__(256) fill_kernel(float* p, int64_t count, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) p[i] = v;
}


// Rows per tile.  A tile's work is one long dependent chain (assemble, GEMM, LayerNorm, ... ~40 us forward, ~85 us
// backward at 64 rows), and configs[2]'s own batch (4 tracklets, 8-10 k points) is 128-160 tiles of 64 rows on 256
// CUs.  32-row tiles -- twice the workgroups, half the chain, the weight fragments streamed once more per row from
// L2 -- are faster over the whole range the fused layer is used in (whole configs[2] step, MI355X: 17.4 -> 16.0 ms at 4
// tracklets = 8 k points, 28.9 -> 27.5 at 16, 46.7 -> 44.8 at 32, 84.6 -> 82.6 at 64 = 131 k points); beyond that
// (not measured) the 64-row tile's halved weight traffic is kept.  g_force_tile: tests pin any of the three forms.
int g_force_tile = 0;
inline int tile_mb(int64_t rows) {
  if (g_force_tile == 16) return 1;
  if (g_force_tile == 32) return 2;
  if (g_force_tile == 64) return 4;
  if (rows <= 4096) return 1;   // (one or two tracklets: 16-row tiles, ~1 ms of a 15 ms step in two of two A/B pairs; a wash at 8 k rows)
  return ococc_cdiv(rows, TR) <= 2048 ? 2 : 4;
}

constexpr int kMaxDevices = 64;
inline int current_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d % kMaxDevices;
}

}  // namespace

int point_mlp_tile_rows(int64_t rows) { return 16 * tile_mb(rows); }   // (csrc/sir_fused.hip: the same tiles as the per-block launches)

extern "C" int64_t ococc_point_mlp_fragment_floats(int32_t n, int32_t k) {
  return n <= 0 || k <= 0 ? -1 : (int64_t)((n + 15) >> 4) * (pad_k(k) >> 2) * 64;
}

extern "C" int ococc_point_mlp_pack_f32(const float* w, int32_t n, int32_t k, int64_t row_stride, int64_t col_stride,
                                        float* frag, ococc_stream_t stream) {
  OCOCC_REQUIRE(w && frag && n > 0 && k > 0, "bad arguments");
  const int64_t total = ococc_point_mlp_fragment_floats(n, k);
  hipLaunchKernelGGL(point_mlp_pack_kernel, dim3(ococc_grid_1d(total, 256, 256)), dim3(256), 0, (hipStream_t)stream, w, (int)n,
                     (int)k, row_stride, col_stride, frag);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_point_mlp_pack_multi_f32(int32_t count, const void* const* w, const int32_t* n, const int32_t* k,
                                              const int64_t* row_stride, const int64_t* col_stride, void* const* frag,
                                              ococc_stream_t stream) {
  OCOCC_REQUIRE(count >= 0 && count <= kMaxPack, "at most 32 matrices per call");
  if (count == 0) return OCOCC_OK;
  OCOCC_REQUIRE(w && n && k && row_stride && col_stride && frag, "null pointer table");
  PackMulti pk;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    OCOCC_REQUIRE(w[i] && frag[i] && n[i] > 0 && k[i] > 0, "bad matrix");
    pk.w[i] = (const float*)w[i];
    pk.dst[i] = (float*)frag[i];
    pk.rs[i] = row_stride[i];
    pk.cs[i] = col_stride[i];
    pk.n[i] = n[i];
    pk.k[i] = k[i];
    pk.first_block[i] = blocks;
    blocks += (int)ococc_cdiv(ococc_point_mlp_fragment_floats(n[i], k[i]), 256);
  }
  pk.first_block[count] = blocks;
  pk.count = count;
  hipLaunchKernelGGL(point_mlp_pack_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pk);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

static int check_in(const PointMlpIn& in, int n) {
  const int k = in.ka + in.kb + in.kv;
  OCOCC_REQUIRE(in.rows >= 0 && in.ka >= 0 && in.kb >= 0 && in.kv >= 0 && k >= 1 && k <= kMaxK && n >= 1 && n <= kMaxN,
                "1 <= input columns <= 256, 1 <= output channels <= 144");
  OCOCC_REQUIRE((in.ka == 0 || in.a) && (in.kb == 0 || in.b) && (in.kv == 0 || (in.v && in.inv)), "null input part");
  OCOCC_REQUIRE(in.lda >= in.ka && in.ldb >= in.kb && (!in.mul || in.ldm >= in.ka), "row strides shorter than the columns");
  return OCOCC_OK;
}

extern "C" int ococc_point_mlp_fwd_f32(const float* a, int32_t ka, int32_t lda, const float* mul, int32_t ldm,
                                       const float* colscale, const float* b, int32_t kb, int32_t ldb, float bscale,
                                       const float* v, int32_t kv, const int32_t* inv, int64_t rows, const float* w_frag,
                                       int32_t n, const float* ln_weight, const float* ln_bias, float eps, int32_t act,
                                       float* y, float* seg_max, int64_t num_segments, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PointMlpIn in{a, mul, colscale, b, v, inv, ka, lda, ldm, kb, ldb, kv, bscale, rows};
  if (int rc = check_in(in, n)) return rc;
  OCOCC_REQUIRE(w_frag && y && (act >= 0 && act <= 2) && (!ln_weight == !ln_bias), "bad arguments");
  OCOCC_REQUIRE(!seg_max || (inv && num_segments >= 0), "segment maxima need inv and the segment count");
  if (seg_max && num_segments > 0)
    hipLaunchKernelGGL(fill_kernel, dim3(ococc_grid_1d(num_segments * n, 256, 1024)), dim3(256), 0, stream, seg_max,
                       num_segments * n, -INFINITY);
  if (rows == 0) return OCOCC_OK;
  const int k = ka + kb + kv, kp = pad_k(k), np = pad_k(n);
  const int mb = tile_mb(rows);
  const int lds = lds_floats(kp, np, 16 * mb) * 4;
  const unsigned grid = (unsigned)ococc_cdiv(rows, 16 * mb);
  const int dev = current_device_slot();
#define OCOCC_PM_FWD(NBW, MB)                                                                                             \
  do {                                                                                                                \
    static int lds_set[kMaxDevices] = {};  /* (the attribute sticks: raised once per instantiation and device) */     \
    if (lds > lds_set[dev]) {                                                                                         \
      OCOCC_HIP(hipFuncSetAttribute((const void*)point_mlp_fwd_kernel<NBW, MB>,                                        \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                \
      lds_set[dev] = lds;                                                                                             \
    }                                                                                                                 \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(point_mlp_fwd_kernel<NBW, MB>), dim3(grid), dim3(kT), lds, stream, in, w_frag, \
                       (int)n, ln_weight, ln_bias, eps, (int)act, y, seg_max);                                        \
  } while (0)
  if (mb == 1) {
    switch (nbw_of(n)) {
      case 1: OCOCC_PM_FWD(1, 1); break;
      case 2: OCOCC_PM_FWD(2, 1); break;
      default: OCOCC_PM_FWD(3, 1); break;
    }
  } else if (mb == 2) {
    switch (nbw_of(n)) {
      case 1: OCOCC_PM_FWD(1, 2); break;
      case 2: OCOCC_PM_FWD(2, 2); break;
      default: OCOCC_PM_FWD(3, 2); break;
    }
  } else {
    switch (nbw_of(n)) {
      case 1: OCOCC_PM_FWD(1, 4); break;
      case 2: OCOCC_PM_FWD(2, 4); break;
      default: OCOCC_PM_FWD(3, 4); break;
    }
  }
#undef OCOCC_PM_FWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_point_mlp_bwd_f32(const float* a, int32_t ka, int32_t lda, const float* mul, int32_t ldm,
                                       const float* colscale, const float* b, int32_t kb, int32_t ldb, float bscale,
                                       const float* v, int32_t kv, const int32_t* inv, int64_t rows, const float* w_frag,
                                       const float* wt_frag, int32_t n, const float* ln_weight, const float* ln_bias,
                                       float eps, int32_t act, const float* dy, const float* d_seg_max,
                                       const int32_t* seg_arg, float* dz, float* x_cat, float* da, float* dmul, float* db,
                                       float* dv, float* ln_partial, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  PointMlpIn in{a, mul, colscale, b, v, inv, ka, lda, ldm, kb, ldb, kv, bscale, rows};
  if (int rc = check_in(in, n)) return rc;
  OCOCC_REQUIRE(w_frag && wt_frag && dz && (act >= 0 && act <= 2) && (!ln_weight == !ln_bias), "bad arguments");
  OCOCC_REQUIRE((!d_seg_max) == (!seg_arg) && (!d_seg_max || inv), "the gradient of the segment maxima comes with seg_arg and inv");
  OCOCC_REQUIRE(!ln_weight || ln_partial, "LayerNorm partial rows missing");
  if (rows == 0) return OCOCC_OK;
  const int k = ka + kb + kv, kp = pad_k(k), np = pad_k(n);
  const int mb = tile_mb(rows);
  const int lds = lds_floats(kp, np, 16 * mb) * 4;
  const unsigned grid = (unsigned)ococc_cdiv(rows, 16 * mb);
  const int kbw = (((k + 15) >> 4) + 3) / 4;
  const int dev = current_device_slot();
#define OCOCC_PM_BWD(NBW, KBW, MB)                                                                                            \
  do {                                                                                                                    \
    static int lds_set[kMaxDevices] = {};                                                                                 \
    if (lds > lds_set[dev]) {                                                                                             \
      OCOCC_HIP(hipFuncSetAttribute((const void*)point_mlp_bwd_kernel<NBW, KBW, MB>,                                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                    \
      lds_set[dev] = lds;                                                                                                 \
    }                                                                                                                     \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(point_mlp_bwd_kernel<NBW, KBW, MB>), dim3(grid), dim3(kT), lds, stream, in, w_frag, \
                       wt_frag, (int)n, ln_weight, ln_bias, eps, (int)act, dy, d_seg_max, seg_arg, dz, x_cat, da, dmul,   \
                       db, dv, ln_partial);                                                                               \
  } while (0)
#define OCOCC_PM_BWD_N(KBW, MB)                 \
  switch (nbw_of(n)) {                          \
    case 1: OCOCC_PM_BWD(1, KBW, MB); break;    \
    case 2: OCOCC_PM_BWD(2, KBW, MB); break;    \
    default: OCOCC_PM_BWD(3, KBW, MB); break;   \
  }
#define OCOCC_PM_BWD_K(MB)                      \
  switch (kbw) {                                \
    case 1: OCOCC_PM_BWD_N(1, MB); break;       \
    case 2: OCOCC_PM_BWD_N(2, MB); break;       \
    case 3: OCOCC_PM_BWD_N(3, MB); break;       \
    default: OCOCC_PM_BWD_N(4, MB); break;      \
  }
  if (mb == 1) {
    OCOCC_PM_BWD_K(1)
  } else if (mb == 2) {
    OCOCC_PM_BWD_K(2)
  } else {
    OCOCC_PM_BWD_K(4)
  }
#undef OCOCC_PM_BWD_K
#undef OCOCC_PM_BWD_N
#undef OCOCC_PM_BWD
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int32_t ococc_point_mlp_wgrad_slices(int64_t rows) {
  if (rows <= 0) return 0;
  int64_t s = rows / 256;
  return (int32_t)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

extern "C" int ococc_point_mlp_wgrad_f32(const float* dz, const float* x_cat, int64_t rows, int32_t n, int32_t k,
                                         float* partial, ococc_stream_t stream_) {
  OCOCC_REQUIRE(rows >= 0 && n >= 1 && k >= 1, "bad sizes");
  if (rows == 0) return OCOCC_OK;
  OCOCC_REQUIRE(dz && x_cat && partial, "null pointer");
  const int slices = ococc_point_mlp_wgrad_slices(rows);
  const int64_t per = ococc_align_up(ococc_cdiv(rows, slices), 32);   // (the last slice may come out short or empty: zeros)
  hipLaunchKernelGGL(point_mlp_wgrad_kernel, dim3(slices, (n + 63) / 64, (k + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream_, dz, x_cat, rows, (int)n, (int)k, per, partial);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int ococc_point_mlp_wgrad_multi_f32(int32_t count, const float* const* dz, const float* const* x_cat, int64_t rows,
                                               const int32_t* n, const int32_t* k, float* const* partial,
                                               ococc_stream_t stream_) {
  OCOCC_REQUIRE(count >= 1 && count <= kWgradMulti && rows >= 0, "1..24 layers per call");
  OCOCC_REQUIRE(dz && x_cat && n && k && partial, "null pointer table");
  if (rows == 0) return OCOCC_OK;
  const int slices = ococc_point_mlp_wgrad_slices(rows);
  const int64_t per = ococc_align_up(ococc_cdiv(rows, slices), 32);
  PointWgradPack pk;
  int blocks = 0;
  for (int j = 0; j < count; ++j) {
    OCOCC_REQUIRE(dz[j] && x_cat[j] && partial[j] && n[j] >= 1 && k[j] >= 1, "bad layer");
    pk.dz[j] = dz[j]; pk.xc[j] = x_cat[j]; pk.partial[j] = partial[j]; pk.n[j] = n[j]; pk.k[j] = k[j];
    pk.first[j] = blocks;
    blocks += slices * ((n[j] + 63) / 64) * ((k[j] + 63) / 64);
  }
  pk.first[count] = blocks;
  pk.count = count;
  hipLaunchKernelGGL(point_mlp_wgrad_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, pk, rows, per, slices);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}

extern "C" int64_t ococc_point_mlp_tiles(int64_t rows) { return rows < 0 ? -1 : ococc_cdiv(rows, 16 * tile_mb(rows)); }

// tests: rows per tile pinned to 32 or 64 (0: by input size)
extern "C" int ococc_point_mlp_force_tile(int32_t tile_rows) {
  OCOCC_REQUIRE(tile_rows == 0 || tile_rows == 16 || tile_rows == 32 || tile_rows == 64, "0 (automatic), 16, 32 or 64");
  g_force_tile = tile_rows;
  return OCOCC_OK;
}

extern "C" int ococc_point_mlp_segment_argmax(const float* y, const float* seg_max, const int32_t* inv, int64_t rows,
                                              int32_t n, int64_t num_segments, int32_t* seg_arg, ococc_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  OCOCC_REQUIRE(rows >= 0 && n >= 1 && num_segments >= 0, "bad sizes");
  if (num_segments == 0) return OCOCC_OK;
  OCOCC_REQUIRE(y && seg_max && inv && seg_arg, "null pointer");
  OCOCC_HIP(hipMemsetAsync(seg_arg, 0x7f, (size_t)num_segments * n * 4, stream));
  if (rows == 0) return OCOCC_OK;
  hipLaunchKernelGGL(segment_argmax_kernel, dim3((unsigned)ococc_cdiv(rows * ((n + 3) / 4), 256)), dim3(256), 0, stream, y,
                     seg_max, inv, rows, (int)n, seg_arg);
  OCOCC_CHECK_LAUNCH();
  return OCOCC_OK;
}
